import os, sys, time, torch
sys.path[:0]=[os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/dif-pan_amd', os.environ.get('GRAFT_REPO_ROOT','/root/repo')]
from ddif.layout import engine_cfg
from ddif.synth import synth_state_dict, synth_tiles
from oracle import ddif_oracle as O
n=int(sys.argv[1]); torch.set_num_threads(n)
cfg=engine_cfg(8,1); sd=synth_state_dict(cfg); cond=synth_tiles(1)["cond"]
tabs=O.schedule_tables(O.cosine_betas(1000))
def run(k):
    t=time.perf_counter()
    with torch.no_grad(): O.ddpm_sample(sd,cfg,cond,tabs,max_steps=k)
    return time.perf_counter()-t
run(1); print(n,'threads:', run(3)/3,'s/step', flush=True)

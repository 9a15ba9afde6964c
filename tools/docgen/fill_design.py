import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from design_parts import STATUS, MEASURE, SHARES, R6
tag = sys.argv[1]          # gpurun_out prefix, e.g. r06_z
ser = tag[4:]
G = 'gpurun_out/' + tag + '_'
r = json.load(open(G + 'bench_T1000_B64.json'))
rf = r['roofline']; ws = rf['whole_step']
bf = json.load(open(G + 'bench_wv3_bf16.json')); gf = json.load(open(G + 'bench_gf2_dpm50.json')); cv = json.load(open(G + 'bench_cave128_t2000.json'))
tr = json.load(open(G + 'bench_wv3_train_b32.json')); t4 = json.load(open(G + 'bench_wv3_train_b4_share.json'))
busy = json.load(open(G + 'mfma_busy.json'))['whole_run_mfma_busy']
sh = gf['projected_strong_scaling']['ms_per_job_by_tiles_per_gpu']; pj = gf['projected_strong_scaling']['projected_speedup_by_gpus']
def whole_round():
    # profiles/r06/whole_round_lib_ab.txt: lines 'other rep N ms/step X' (round 5's library) and 'tree rep N ms/step Y', the LAST block of the file
    a, b = [], []
    for ln in open('profiles/r06/whole_round_lib_ab.txt'):
        mm = re.match(r'(other|tree) rep \d+ ms/step ([\d.]+)', ln)
        if mm: (a if mm.group(1) == 'other' else b).append(float(mm.group(2)))
    n = 3
    a, b = a[-n:], b[-n:]
    ma, mb = sum(a) / len(a), sum(b) / len(b)
    return '%.3f → %.3f ms = −%.1f %%' % (ma, mb, 100 * (1 - mb / ma))
WHOLE = whole_round()
tests = open(G + 'tests.log').read()
m = re.search(r'(\d+) passed.* in ([\d.]+)s', tests)
ngpu, tgpu = (m.group(1), str(int(float(m.group(2))))) if m else ('?', '?')
classes = '\n'.join('| %s | %d | %.3f | %.3f | %.3f |' % (c['class'], round(c['launches_per_step']), c['ms_per_step'], c['floor_ms_per_step'], c['frac_of_floor']) for c in ws['classes'])
classes += '\n| whole step | 132 | %.3f | %.3f | **%.3f** |' % (ws['ms_per_denoising_step'], rf['step_floor_ms'], rf['step_frac'])
def rows(path):
    d = {}
    for ln in open(path):
        mm = re.match(r'(\S+)\s+([\d.e+-]+)\s+\((\S+)\)(?:\s+PSNR diff ([\d.e+-]+) dB)?', ln)
        if mm: d[mm.group(1)] = (mm.group(2), mm.group(4))
    return d
pa, pb = rows(G + 'parity_report_f16x2.txt'), rows(G + 'parity_report_exact_fp32.txt')
groups = [('forward × 7 (16² … 64², three data sets)', [k for k in pa if k.startswith('fwd_')]),
          ('`ddpm_wv3_64_T1000`, `_b`, `_c` (configs[1], three tiles, 1000 steps)', ['ddpm_wv3_64_T1000', 'ddpm_wv3_64_T1000_b', 'ddpm_wv3_64_T1000_c']),
          ('`ddpm_cave_128_T2000` (configs[3], full 2000-step chain)', ['ddpm_cave_128_T2000']),
          ('`dpm_gf2_64_T1000_s50_o2` (configs[2], 50 NFE)', ['dpm_gf2_64_T1000_s50_o2']),
          ('other DDPM / DDIM goldens × 5', ['ddpm_wv3_16_T10', 'ddpm_wv3_16_T1000', 'ddpm_gf2_32_T50', 'ddim_wv3_32_T500_25', 'ddim_gf2_16_T1000_25']),
          ('other DPM-Solver++ goldens × 4 (order 2 / 3)', ['dpm_gf2_32_T1000_s10_o2', 'dpm_gf2_32_T1000_s50_o2', 'dpm_wv3_16_T500_s12_o3', 'dpm_wv3_16_T500_s6_o3'])]
def mx(d, ks, i):
    v = [float(d[k][i]) for k in ks if k in d and d[k][i]]
    return ('%.1e' % max(v)) if v else '—'
par = '| golden(s) | max \\|hip − reference\\|, f16x2 (default) | exact fp32 | max ΔPSNR dB (f16x2 / exact) |\n|---|---|---|---|\n'
par += '\n'.join('| %s | %s | %s | %s / %s |' % (n, mx(pa, ks, 0), mx(pb, ks, 0), mx(pa, ks, 1), mx(pb, ks, 1)) for n, ks in groups)
sub = {
 'MS': '%.3f' % (r['ms_per_step'] / 1000), 'MSJOB': '%.0f' % r['ms_per_step'], 'MPS': '%.4f' % r['value'], 'BUILD': r['build_id'], 'XCPU': '%.0f' % r['vs_cpu_baseline'], 'XCPU1': '%.0f' % r['vs_cpu_baseline_b1'],
 'SF': '%.3f' % rf['step_frac'], 'NLAUNCH': str(r['config']['launches_per_denoising_step']), 'PMAX': '%.1e' % r['parity']['max_abs'], 'PPS': '%.1e' % r['parity']['psnr_diff_db'],
 'EX': '%.2f' % r['exact_fp32']['ms_per_denoising_step'], 'ACH': '%.1f' % rf['achieved'], 'FR': '%.3f' % rf['frac'], 'ALU': '%.1f' % rf['avg_launch_us'],
 'TRF': '%.1f' % ((rf['traffic'] or 0) / 1e6), 'TRX': '%.2f' % ((rf['traffic'] or 0) / rf['algorithmic_bytes_per_launch']), 'BUSY': '%.3f' % busy, 'CLASSES': classes,
 'CPU': '%.2e' % r['cpu_baseline']['value'], 'BF': '%.4f' % bf['value'], 'BFMS': '%.3f' % (bf['ms_per_step'] / 1000), 'BFD': '%.1e' % bf['drift']['max_abs'],
 'GF': '%.3f' % gf['value'], 'GFX': '%.0f' % gf.get('vs_cpu_baseline', 0), 'CVX': '%.0f' % cv.get('vs_cpu_baseline', 0), 'GFMS': '%.1f' % gf['ms_per_step'], 'CV': '%.4f' % cv['value'], 'CVMS': '%.2f' % (cv['ms_per_step'] / 1000), 'TRV': '%.0f' % tr['value'], 'TR': '%.2f' % tr['ms_per_step'],
 'T4': '%.1f' % t4['ms_per_step'], 'S64': '%.1f' % sh['64'], 'S32': '%.1f' % sh['32'], 'S16': '%.1f' % sh['16'], 'S8': '%.1f' % sh['8'], 'P2': '%.2f' % pj['2'], 'P4': '%.2f' % pj['4'], 'P8': '%.2f' % pj['8'],
 'LRMS': '%.2f' % [c for c in ws['classes'] if c['class'].startswith('low-resolution')][0]['ms_per_step'], 'LRFR': '%.2f' % [c for c in ws['classes'] if c['class'].startswith('low-resolution')][0]['frac_of_floor'],
 'WHOLE': WHOLE,
 'NGPU': ngpu, 'TGPU': tgpu, 'NCPU': '134', 'PR': ser + '_', 'PARITY': par,
}
s = open('DESIGN.md').read()
for k, v in (('STATUS', STATUS), ('MEASURE', MEASURE), ('SHARES', SHARES), ('R6', R6)):
    s = s.replace('@@' + k + '@@', v)
for k, v in sub.items():
    s = s.replace('@@' + k + '@@', v)
left = re.findall(r'@@\w+@@', s)
open('DESIGN.md', 'w').write(s)
lines = s.split('\n')
print('lines', len(lines), 'longest', max(len(l) for l in lines), 'over160', sum(1 for l in lines if len(l) > 160), 'placeholders left', left)

STATUS = r'''Headline (BASELINE configs[1], B = 64 tiles of 64×64×8, T = 1000, one MI355X): **@@MS@@ ms per denoising step = @@MPS@@ MP/s** in the final set
`profiles/r06/z_*` (build `@@BUILD@@`; two runs of the set on two boxes: 3.390 and 3.389 ms; the build before the last two neutral changes measured **3.256 ms = 0.0805 MP/s** on a faster box, `y_bench_T1000_B64_build_6caf75ed.json`), **132 launches** per
step (146 in round 5), @@XCPU@@ × the 16-thread CPU port (@@XCPU1@@ × its B = 1 rate), `roofline.step_frac` @@SF@@ (0.183 in round 5). Driver-measured round 5: 3.541 ms.
**The whole round on one box** (round 5's final library against this one, interleaved three times, `profiles/r06/whole_round_lib_ab.txt`): **@@WHOLE@@**; the boxes of the pool differ by ± 5 % (the tree before the row staging measured 3.27–3.59 ms on six of them, round 5's 3.54–3.82): only same-box pairs are comparable, and every change below is one (single A/Bs: −2.4, −1.9, −2.0, −1.7, −1.2, −0.7, −0.5, −0.5, −0.4 %).

| VERDICT r5 item | status | evidence |
|---|---|---|
| 1. headline ≤ 3.20 ms, ≤ 126 launches | **not met: @@MS@@ ms in the final set, 132 launches** | (a) `x_conv` + FiLM folded into the producer's epilogue where one wave holds all couts (`n_ct == 1`: stem, two `block2`, first Downsample) as a register GEMM on the accumulator layout (§3 EPI_XF): −4 launches, −0.4 % — the fold only saves the re-read of x, the 67 MB FiLM tensor and both outputs remain (`xf_fold_ab.txt`); the eight other sites have 32-cout tiles under 64–128 couts and would need a cross-wave exchange for ≤ 10 µs each: not built. (b) the 8² decoder attention half is ONE launch (`linattn8_fused`, §3): 12 → 4 launches, −1.7 % (`la8_ab.txt`); the 192-channel block of the 16² level joined `linattn_fused` (−2 launches, −0.5 %). (c) `linattn_fused` at 64²: table published (`mbench_la_ablations_stamps.txt`: the softmax + attn_out section was 13.5 k of 31.5 k ticks per work item, all cross-lane reductions); q contraction issued transposed → reductions in registers, one online-softmax exchange per wave pair: **73.8 → 56.9 µs** (≤ 50 not met), −1.9 % on the step; four-wave workgroups where eight leave CUs idle: −1.2 %. The "105 µs first instance" is the 96-channel block (6 chunks + 3 softmax blocks instead of 4 + 2; now 87 µs). The "72 µs first `ffn.0`" is not a property of the op: which of the four instances is slow changes from process to process and none was in the probe run (`arena_pad_probe.txt`: 47.0–48.4 µs all four, any padding) — physical placement of the allocation, not reproducible. (d) the final conv already ran f16x2 (tiling 37); its 42 µs are the sampler epilogue's loads: not changed. Beyond the list: `attn_block` on four workgroups per sample (22.5 → 15.8 µs, −2.0 %) |
| 2. small batch (8 / 16-tile shares ≤ 95 / 105 ms) | **not met: @@S8@@ / @@S16@@ ms** (120.7 / 128.2 in round 5); projected 8-GPU speed-up @@P8@@ × (1.67) | the half-tile GroupNorm-partial granularity IS built (`ConvArgs::st_halves`): the 16×16 tilings write their partials per 8×16 half in the small tiling's order, the conv outputs of both tilings are bit-identical, so the plan picks 8×16 tiles when a launch would not fill the CUs and a tile of a batch stays **bit-equal** to the tile run alone (tested with mixed tilings): −4.2 % at 8 tiles. The fused kernels above carry the rest (−8 launches, four-wave / split workgroups). `bench.py --config gf2_dpm50` now measures the 32 / 16 / 8-tile shares itself (`projected_strong_scaling`); config 5's per-rank share (4 tiles): **@@T4@@ ms** per iteration (`z_bench_wv3_train_b4_share.json`) |
| 3. parity record in the artefacts | **done** | `tools/parity_report.py` in `gpu_full.sh`, both modes committed (§4: every margin ≤ 1.6e-5); `bench.py` line carries `parity` (golden T = 1000 re-run: max @@PMAX@@, ΔPSNR @@PPS@@ dB) and `exact_fp32` / `ms_per_step_exact_fp32` (@@EX@@ ms per step, 184 launches, child process with `DDIF_F16=0 DDIF_X3=0`) |
| 4. config-exact goldens | **done** | `dpm_gf2_64_T1000_s50_o2`, `ddpm_cave_128_T2000` (full chain), `ddpm_wv3_64_T1000_b/_c`; the B = 64 test deals three distinct goldens out as b % 3; `-m gpu` = @@TGPU@@ s (585 s in round 5) |
| 5. low-resolution class ≤ 1.05 ms | **not met: 1.28 → @@LRMS@@ ms (1.07 on the faster box)** | the 3×3 convs of the class stage by ROWS where the tile spans the image width (`kernels_lr.h` ROWS, §3): the general staging spent ≈ 25 instructions of pixel decomposition, clamps and 64-bit products on each of its 12–13 items before the first load went out and then loaded, normalised and masked the 36 % of the halo tile that is zero padding (`lr_stamps.txt`); now the item is a halo row, the thread's column is fixed, padding rows are skipped by a uniform branch: 2 452 → 1 818 instructions in front of the first MFMA, **−1.95 µs on each of the 48 launches, −2.4 % on the step**, bit-identical (`lr_rows_ab.txt`). Before that: the fused 8² attention half and −10 launches (1.28 → 1.15). Its K loop now prefetches the next tap's A fragments and issues the products product-major over the accumulator blocks (bit-identical; class −1.6 %, training −0.8 %, `lr_kloop_lib_ab.txt`). Two compiler artefacts found on the way and removed, measured neutral (`late_waits_lib_ab.txt`). Measured and dropped: the staged tile kept across the two cout tiles a workgroup would then walk in the 512-item launches (+1.1 %: two co-resident workgroups hide each other's chains better, `lr_keep_tile_ab.txt`). Not measured: pilot launch, chained `res.conv1 → res.conv2` at 16², K-partials through DPP |
| 6. training ≤ 21 ms | **not met: @@TR@@ ms** | only the low-resolution kernel's changes carry over to train-mode plans (bf16x3 instantiations: row staging −0.7 %, K loop −0.8 %; `lr_rows_ab.txt`, `lr_kloop_lib_ab.txt`); 2 270 launches of ≈ 11 µs, no single kernel above 11 % (`z_train_kernel_stats.csv`); the batched weight-gradient reduction was measured slower in round 5 (`DESIGN_HISTORY.md` §7) |
| 7. ready the 8-GPU run | **done** | world-4 gloo tests (scene shard 4 × 1 tile ≡ 1 × 4 bit for bit; DDP step, four seeds → identical replicas); `bench.py` at N > 1 prints `rccl` (ranks, per-rank ms min / max / all, all-gather µs, bus GB/s), training line the all-reduce |
| 8. prune and index | **done** | per-op tape wrappers → `tests/ddif_testops.py`, their 31 declarations → `include/ddif_testops.h` (`include/ddif.h` = 48 product entry points; `test_cabi_symbols` checks the product binds none); this file ≤ 300 lines, the record → `DESIGN_HISTORY.md`; `profiles/rNN/` with an index each; conv instantiations in four translation units (build 6 → 2.5 min) |
| ADVICE r5 (1 medium, 4 low) | fixed | `PlanHandle` keeps its math mode across a range-fallback rebuild, re-arms profiling, skips the flag read inside a stream capture, `DDIF_RANGE_CHECK=0` opt-out, blocking documented (INTEGRATION §4); NaN comment corrected; parameter cache validated per use; stitch collectives only with > 1 rank or an RCCL group; `bench.py` needs RANK and WORLD_SIZE; stale notes / dead scripts removed |'''

MEASURE = r'''**Headline line** (`profiles/r06/z_bench_T1000_B64.json`; driver contract + `roofline` + `cpu_baseline` + `parity` + `exact_fp32`): value @@MPS@@ MP/s,
@@MSJOB@@ ms per T = 1000 job of 64 tiles, `dtype: "f32-class (f16x2 split)"`, inputs resident in HBM, `vs_baseline` null (BASELINE.md has no published number).

* `roofline` (dominant kernel class = the 49 3×3 convs of the 64² / 32² levels, HIP events around every launch of 40 whole steps inside the timed region):
  `achieved` = algorithmic fp32 flops / duration = **@@ACH@@ TF**; `peak` = 2516.8 / 3 = 838.9 TF (three 16-bit products per fp32 product), `frac` **@@FR@@**;
  average launch @@ALU@@ µs (rocprofv3: `z_kernel_stats_T20_B64.csv`), algorithmic 65.5 MB per launch, `traffic` (PMC FETCH_SIZE × 2 + WRITE_SIZE, separate passes)
  **@@TRF@@ MB = @@TRX@@ × algorithmic**; whole-run MFMA-busy @@BUSY@@ (`z_mfma_busy.json`).
* One scale round over round: `step_floor_ms` = Σ over classes of max(issued 16-bit MFMA flops / 2516.8 TF, algorithmic bytes / 8 TB/s) = 0.641 ms (a property of
  the launch program); `step_frac` = floor / measured = **@@SF@@** (0.171 round 4, 0.183 round 5).

| class | launches / step | ms / step | floor ms | fraction of floor |
|---|---|---|---|---|
@@CLASSES@@

* `cpu_baseline`: the oracle (kind "port") on 16 host threads of the GPU box (cgroup quota of a 256-CPU host), first 210 of 1000 steps at B = 1 and 133 at B = 8
  scaled to T: @@CPU@@ MP/s (B = 8, the better) → **@@XCPU@@ ×**; against the B = 1 rate of BASELINE configs[0]: @@XCPU1@@ ×.
* `exact_fp32`: the same workload with every conv on `v_mfma_f32_32x32x2_f32` (`DDIF_F16=0 DDIF_X3=0`, 184 launches): **@@EX@@ ms per step** — the f16x2 headline is
  quoted between that and the bf16 throughput variant below; `parity`: the golden `ddpm_wv3_64_T1000` re-run in the benchmarked mode: max @@PMAX@@, ΔPSNR @@PPS@@ dB.

| configuration (`bench.py --config`) | result | note |
|---|---|---|
| `wv3_bf16` (configs[1] "bf16": conv operands rounded once, one product; never a parity configuration) | @@BF@@ MP/s, @@BFMS@@ ms per step | drift vs the fp32-class path after T = 1000: max @@BFD@@ |
| `gf2_dpm50` (configs[2]: one 512² scene = 64 tiles, DPM-Solver++ 2M, 50 NFE) | @@GF@@ MP/s, @@GFMS@@ ms per scene | @@GFX@@ × the oracle's DPM-Solver++ on 16 host threads (`cpu_baseline` in the line); per-rank shares: §6 |
| `cave128_t2000` (configs[3]: 8 patches of 128×128×31, T = 2000) | @@CV@@ MP/s, @@CVMS@@ s per job | @@CVX@@ × the CPU port; full chain vs the reference golden: 1.3e-6 (§4) |
| `wv3_train_b32` (configs[4]: one training iteration, batch 32 per GPU) | @@TRV@@ tiles/s, **@@TR@@ ms** per iteration | per-rank share of the stated global batch (4 tiles): @@T4@@ ms |

PCIe-inclusive rate: the boundary takes device tensors; uploading a job's cond + x_T (64 tiles × 28 channels × 64² × 4 B = 29 MB) costs < 1 ms of a 3.3 s job.'''

SHARES = r'''| tiles on the GPU (= ranks of an N-GPU run of the 64-tile scene) | 64 (N = 1) | 32 (2) | 16 (4) | 8 (8) |
|---|---|---|---|---|
| ms per scene job, one MI355X (`z_bench_gf2_dpm50.json projected_strong_scaling`) | @@S64@@ | @@S32@@ | @@S16@@ | @@S8@@ |
| projected speed-up (compute share only; the 4 MB all-gather + stitch excluded) | 1 | @@P2@@ | @@P4@@ | **@@P8@@** |

Round 5: 201.5 / 151.9 / 128.2 / 120.7 ms → 1.67 ×. An evaluation at 8 tiles is 132 dependent launches of 8–20 µs each: latency chains that do not shrink with
the work. CPU coverage of N > 1: `tests/test_distributed_cpu.py` (gloo, world 2 and 4, emulated kernels); RCCL itself at world size 1 on the lease
(`tests/test_rccl_world1.py`: all-gather, flat-bucket all-reduce, `torch.distributed.run --nproc-per-node 1 bench.py`).'''

R6 = r'''Kept (each a same-box interleaved A/B; files under `profiles/r06/`):

| change | mechanism | effect |
|---|---|---|
| transposed-q softmax in `linattn_fused` (`lafuse_softmax_lib_ab.txt`) | `D = XW` puts a column's rows into registers: max / sum are in-lane trees + one `ds_bpermute`; a wave pair exchanges (max, sum) once (online softmax) | 73.8 → 56.9 µs @64², step −1.9 % |
| `attn_block` on 4 workgroups per sample, qkv on f16x2 (`attn_split_ab.txt`, `attn_f16_ab.txt`) | the block is bound by the matrix pipe of its CU (`attn_block_stamps.txt`); k, v recomputed by each, no exchange; half the products for the GroupNorm-bounded qkv conv | 22.5 → 15.8 µs, step −2.0 % and −0.5 % |
| `linattn8_fused` (`la8_ab.txt`) | half a sample per workgroup, waves split the output channels, weights straight from L2 | 3 launches → 1 per block, step −1.7 % |
| four-wave `linattn_fused` workgroups (`la_nw_ab.txt`) | a wave's work does not depend on the workgroup size; with idle CUs, give every wave its own SIMD | step −1.2 %, bit-identical |
| 192-channel 16² block on `linattn_fused` (`la6_ab.txt`) | a sixth q block in registers (42 spilled) still beats three launches | step −0.5 % |
| batched table loads in the `linattn*` prologues (`la_tables_lib_ab.txt`) | the load → store loops were one dependent round trip per iteration on cold caches | step −0.7 % |
| EPI_XF (`xf_fold_ab.txt`) | register GEMM on the accumulator layout, K order permuted to match | −4 launches, step −0.4 % |
| 8×16 tiles at small batch + half-tile partials (`tile16_ab.txt`) | the partial array is independent of the tiling, so batch bit-equality survives | 8-tile share −4.2 % |

Measured and not kept: `attn_block` on eight waves (23.3 vs 22.3 µs: the matrix pipe has no idle slots to fill); four-wave `linattn_fused` workgroups where the eight-wave grid already fills the CUs (+4.3 % on the step, `la_nw4_everywhere_ab.txt`); the pair-form depthwise stage for the
multi-block `linattn_fused` variants (fewer LDS reads, 20–40 spilled registers: 45.5 → 48.6 µs at 32²; kept only where it does not spill: 73.8 → 69.6 µs);
an arena-placement explanation of the one slow `ffn.0` (not reproducible: above). Learned about the part: a kernel whose waves reduce ACROSS lanes per value is
vector-issue-bound long before memory matters — moving the reduction axis into registers by transposing the MFMA was worth more than any of round 5's
boundary work; `v_mfma` accumulators can feed the next contraction directly when the K order is chosen to match their layout (EPI_XF); and at ≤ 128
workgroups a kernel should shrink its workgroups before anything else. Toolchain: hipcc hoists loop-invariant per-lane address adds out of a sample loop into
spilled registers (an opaque `asm volatile("" : "+v"(x))` per iteration stops it: `kernels_attn.h`, `kernels_lafuse8.h`); a child process started from under
`rocprofv3 --pmc` hangs the call (`bench.py` skips its bracket run when profiled).

Still open, in order: the low-resolution class (@@LRMS@@ ms at @@LRFR@@ of its floor: 65 dependent chains of ≈ 13 µs; `DESIGN_HISTORY.md` §7 has five rounds of
what does not help; stamped at the end of this round, `lr_stamps.txt`: with WARM caches an item still takes 8.8 µs in the kernel, 31 % of it before its loads are even issued -- 1 000 instructions of hoisted address geometry -- and 24 % staging at one wave per SIMD with 36 % of the halo tile being zero padding: row-wise staging geometry, the partial loads first and no work on padding are worth an estimated 1–1.5 µs per launch × 65), the eight `film.x_conv` launches under wide cout tiles, `linattn8_fused`'s table prologue (3.4 µs of 20), training (§0 #6), and any
N > 1 measurement.'''

// Implicit-GEMM convolution (3x3 / 1x1, NHWC fp32) for gfx950.
//
// Replaces what the reference runs as torch `nn.Conv2d` + the elementwise ops around it: models/sr3_dwt.py:288-300
// (Block: GroupNorm -> Swish -> conv3x3), :303-327 (ResnetBlock: + time bias, + residual), :266-282 (Upsample /
// Downsample), :376-396 (CondInjection x_conv + FiLM), :528-533 (FFN convs), :338-339 (attention 1x1s).
//
// One workgroup = 4 wavefronts computes a TH x TW pixel tile x (32*NB*WN) output channels of one tile (sample):
//   M = pixels, N = output channels, K = taps x input channels.
//   * the input halo tile is staged ONCE per CK-channel chunk into LDS, with the producer-side GroupNorm(1 group)
//     + SiLU prologue applied while staging (zero padding is applied after the activation, as in the reference);
//     two input tensors can be concatenated along channels without materialising the concat (skip connections,
//     self-conditioning cat[x, x]); nearest x2 upsampling and stride 2 are index maps of the staging pass;
//   * A fragments come from LDS as ds_read_b128 (row stride CK+4 floats); the weight chunk (pre-packed on the host in
//     exactly the per-lane B-fragment order) is staged through LDS as well, one conflict-free 1 KiB ds_read_b128 per
//     wave and (tap, k8) step -- so the MFMA loop issues NO global load: vmcnt retires in order, and a B fragment
//     fetched from L2 inside the loop would make its first use wait for the older HBM tile prefetch too;
//   * the contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain: no precision change vs.
//     an FMA loop, but 64 FLOP/clk/SIMD from one wave);
//   * epilogue: + bias, + per-sample time bias, FiLM (1+scale)*y+shift, SiLU, + residual, store, and the
//     per-(tile, cout-tile) {sum, sum^2} partials the NEXT GroupNorm needs (fp64, deterministic order, no atomics).
//   * PERSISTENT + SOFTWARE-PIPELINED: a launch is 2 workgroups per CU; each workgroup walks a contiguous range of
//     (cout-tile, pixel-tile) work items x channel chunks ("stages").  While the MFMAs of stage s run out of LDS buffer
//     s&1, the global loads of stage s+2 are in flight into registers and the loads of stage s+1 (issued one stage
//     earlier) get the prologue applied and are written to buffer (s+1)&1 (one barrier per stage).
//   * the MFMA is issued as D^T = W^T . X^T (rows = couts, columns = pixels): a lane then owns 4 CONSECUTIVE couts of one
//     pixel per accumulator quad, so the epilogue moves float4s (4 stores / 4 residual loads per 32x32 block instead
//     of 16 scalar ones).  Same products, same k order: bitwise the same result as the untransposed form.
//   * vmcnt discipline (gfx9 counters retire IN ORDER and hipcc merges wait states conservatively at control-flow
//     joins): the steady-state loop is straight-line per stage kind -- `stage<LAST>` is instantiated separately for
//     "last chunk of an item" (epilogue) and "inner chunk", the prefetch is unconditional, the odd tail stage is peeled,
//     epilogue operands are loaded BEFORE the prefetch and live only inside stage<true>.  The previous form (one body
//     with `if (last)` / `if (flat + 2 < nflat)` around the loads) compiled to `s_waitcnt vmcnt(0)` before every
//     epilogue store and before the prefetch, i.e. no overlap of HBM latency with the MFMAs at all.
#pragma once
#include "ddif_dev.h"
#include "sampler_dev.h"

namespace ddif {

struct ConvArgs {
    const float* in0;
    const float* in1;
    int c0, c1;              // channels of the two concatenated sources (c1 = 0: single source)
    int B, Hin, Win;         // source spatial size
    int Hout, Wout, Cout;
    const float* w;          // packed weights, see pack_conv_weights()
    int n_chunks;            // ceil((c0 + c1) / CK)
    const float* bias;       // [Cout]; NEVER null (a zero vector when the conv has no bias: a conditional load would
                             // put a select on the loaded value in front of the MFMAs and drag its vmcnt wait there)
    const float* tbias;      // time bias rows, never null (zeros, strides 0, without one); row of sample b = tbias + step * tb_rowstride + b * tbias_stride
    int tbias_stride;
    const int* step_ptr;     // device step counter of the running sampler (null: step = 0)
    int tb_rowstride;
    const double* st0;       // GroupNorm partials of in0 / in1 (prologue), [B][np][2]
    int np0;
    const double* st1;
    int np1;
    const float* gamma;      // [c0 + c1]
    const float* beta;
    const float* res;        // residual, same shape as out (kernel template EPI_RES must match)
    const float* film;       // [B, Hout, Wout, 2*Cout]: scale | shift (kernel template EPI_FILM must match)
    float* out;
    double* st_out;          // partials of out [B][tiles_x*tiles_y*gridDim.y][2] or null
    int tiles_x, tiles_y;
    int n_ct;                // number of cout tiles (work item = cout tile x pixel tile; partial index uses it)
    long long w_bstride;     // floats between the packed weights of consecutive samples (0: shared weights)
    const float* cs_mx;      // PRO_COLSM: column-softmax max / sum of source 0, [B][Win][c0]
    const float* cs_sm;
    const float* dw_w;       // PRO_GN_DW: depthwise 3x3 weights [9][c0 + c1] applied to the normalised input
    float* out_xn;           // PRO_GN_DW: the normalised input itself, [B,H,W,c0+c1] (consumed by attn_res) or null
    float* cso_mx;           // kernels_lr.h EPI_COLST: column-softmax statistics of the OUTPUT (max / sum of exp over H), [B][Wout][Cout]
    float* cso_sm;
    long long* dbg;          // microbenchmark instrumentation (ABL & 16) only
    int b0;                  // batch window: the launch covers samples [b0, b0 + B) of the tensors (sub-batches of a forked region run concurrently)
    // EPI_SAMP (the network's final conv inside a DDPM / DDIM loop): the sampler update runs in the epilogue -- x0 = this conv's output,
    // out = update(x0, x_t, lms, noise) goes to s_out; the step counter of the NEXT step is written by workgroup 0 (double-buffered counters)
    const float* s_img;      // x_t, NHWC like the output
    const float* s_lms;
    float* s_out;            // x_{t-1}
    const SamplerRun* s_run;
    int* s_step_next;        // receives *step_ptr + 1
    int s_kind;              // 0 = DDPM p_sample, 1 = DDIM
    // f16x2 on a RAW input (no GroupNorm prologue to bound it): a value whose scaled half would overflow (|x| * 2^4 >= 65520 -> inf -> NaN in the output)
    // sets this sticky per-plan flag; the host reads it once per sampler call (ddif_plan_range_status) and rebuilds the plan with these convs on
    // bf16x3 (full fp32 range).  Null: not checked.  (A plain store of 1 by any number of threads: no atomics needed.)
    int* range_flag;
    int xcd;                 // wg_work_range: XCD-contiguous work partition (ddif_dev.h)
    // EPI_XF (round 6): the NEXT block's CondInjection (x_conv 1x1 + FiLM, models/sr3_dwt.py:376-396) in this conv's epilogue -- a second output
    //   y = (W_x out + b_x) * (1 + scale) + shift      with its own GroupNorm partials,
    // computed from the accumulator registers (see the epilogue); `out` itself (the skip feature) is still written.
    const float* xf_w;       // x_conv weights in per-lane A-operand order of the register GEMM: [32-cout block][lane][16] (ddif_net.cpp pack_xf)
    const float* xf_b;       // x_conv bias [xf_cout]
    const float* xf_film;    // [B, Hout, Wout, 2 * xf_cout]: scale | shift of the next block
    float* xf_out;           // [B, Hout, Wout, xf_cout]
    double* xf_st;           // partials of xf_out, [B][tiles_x * tiles_y][2]
    int xf_cout;             // 32 * NBX
    // Round 6: the eight-wave 16 x 16-pixel tilings write their GroupNorm partials per 8 x 16 HALF tile (waves 0-3 / 4-7), at the indices and with the summation
    // order of the four-wave 8 x 16 tiling of the same conv.  The conv outputs of the two tilings are bit-identical (same products, same K order per pixel), so with
    // this the CONSUMER sees the same partial array either way, and the host may pick the tiling by how many workgroups a launch has (8-16 tiles per GPU: the
    // 16 x 16 grid leaves CUs idle) without giving up that a tile of a batch is bit-equal to the tile run alone (tests/test_gpu_batch64.py).
    int st_halves;           // 1: two partials per item (TH = 16, NW = 8 instantiations only)
    int tiles_y8;            // ceil(Hout / 8): tile rows of the 8 x 16 tiling
};

template <int F>
struct StageKind {
    static constexpr bool LAST = (F & 1) != 0;  // last channel chunk of a work item: epilogue
    static constexpr bool TAIL = (F & 2) != 0;  // very last stage of the workgroup: nothing to prefetch or stage
};

// ABL (microbenchmark ablations only, tools/mbench.cpp): 1 = no MFMA, 2 = no input loads, 4 = no stores, 8 = no weight loads,
// 16 = s_memtime stamps of wave 0 (5 per stage) into a.dbg, 128 = no LDS fragment reads (bf16x3 loop), 256 = no staging of the tile
// EPI (epilogue variant, compile-time so that every epilogue-operand load is unconditional straight-line code -- a load
// under `if (a.res)` becomes a phi with undef, hipcc copies the loaded registers right after the load and the copy's
// vmcnt wait lands in front of the prefetch):  1 = FiLM (1+scale)*y+shift,  2 = scalar output path (Cout % 4 != 0),
// 4 = residual add,  8 = SiLU on the output (a runtime flag gets if-converted: exp + rcp computed for every conv),
// 16 = per-SAMPLE time-bias rows (tbias_stride != 0: forward() / p_losses with one t per sample; in the samplers every
// sample shares the step's row and bias + time bias sit in LDS for the whole launch).
// 64 = sampler update in the epilogue (ConvArgs::s_*; diffusion_ddpm_pan.py:418-442 p_sample / :594-621 ddim_sample on the final conv's output).
// 128 / 256 = EPI_XF1 / EPI_XF2 (round 6): the next block's x_conv (32 -> 32 / 64 channels) + FiLM as a second output of this conv.  Needs the whole cout range of a
// pixel in ONE wave (NB = WN = 1, Cout = 32).  The accumulator layout of D^T = W^T X^T -- lane (j, h) holds couts 8g + 4h + i of pixel j -- IS a B operand of
// v_mfma_f32_32x32x2_f32 for a contraction over those couts when step s = 4g + i contracts cout 8g + 4h + i (k = h): the K order of a GEMM is free, the x_conv
// weights are packed in that order, and the 32 x 32 product costs 16 exact-fp32 MFMAs per 32 pixels with no data movement at all.
enum { EPI_FILM = 1, EPI_SOUT = 2, EPI_RES = 4, EPI_SILU = 8, EPI_TBS = 16, EPI_COLST = 32, EPI_SAMP = 64, EPI_XF1 = 128, EPI_XF2 = 256 };
// VEC (input staging): 0 = scalar loads, any channel counts;  1 = float4 loads, every CK-channel chunk lies in ONE source
// (c1 == 0 or c0 % CK == 0): the source base is wave-uniform (SGPR) and a load costs one VALU add;  2 = float4 loads with
// a per-thread source select (the stem's cat[x, x] with 8 + 8 channels).
//
// Why the instruction count of everything around the MFMAs matters more than overlap: measured on MI355X
// (tools/probes/mfma_coexec.cpp) v_mfma_f32_32x32x2_f32 does NOT co-execute with the issuing wave's own VALU work (each
// v_fma between two MFMAs adds its ~5 cycles to the 64-cycle interval) and a second wave's VALU stream runs at half
// speed beside it: the f32 matrix op shares the vector datapath.  So time ~ MFMA cycles + VALU/SALU issue cycles, and
// the levers are (a) fewer non-MFMA instructions per stage (32-bit offsets off uniform bases, no per-item branches),
// (b) more MFMAs per staged byte (8-wave workgroups: a 16x16-pixel tile shares one weight chunk, 64-cout tiles share
// one input chunk).
// MATH: 0 = exact fp32 on v_mfma_f32_32x32x2_f32;  1 = "bf16x3": both operands split three ways into bf16 (hi, mid, lo --
// ddif_dev.h), six cross products per 16-channel slab on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  fp32-class
// result (not bitwise the fmaf chain), 198 instead of 512 matrix cycles per slab, and the matrix core runs beside the
// VALU instead of on it.  LDS holds three bf16 planes per operand (96 B per pixel and 16-channel chunk + 16 B pad).
// 3 = "f16x2" (ddif_dev.h): operands pre-scaled by fixed powers of two and split two ways into IEEE halves (hi, lo), THREE cross
// products per slab on v_mfma_f32_32x32x16_f16 -- the accuracy of the exact-fp32 MFMA (measured, tools/probes/f16x2.cpp) at half
// the matrix instructions of bf16x3, two LDS planes per operand instead of three (64 B per pixel and chunk + 16 B pad, 2/3 of
// the weight bytes) and a 3-op-per-value split instead of 5.5.  The accumulator is scaled back in the epilogue's fma.
// 4 = "bf16x1", the THROUGHPUT variant (BASELINE configs[1] "bf16"; never the default, never a parity configuration): both operands rounded once to
// bf16, ONE product per slab on v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 tensors / GroupNorm statistics / epilogue as everywhere else.
// One LDS plane per operand (32 B per pixel and chunk + 16 B pad).  Selected only by ddif_set_math_mode(DDIF_MATH_BF16) / DDIF_MATH=bf16.
// 5 = f16x2 with RESIDENT weights (round 5; 3x3 convs whose whole input is ONE CK = 32 channel stage and whose couts are one tile: 32 -> 32 at the
// 64 x 64 level): the cout tile's weights (36 KB) are copied into LDS once per workgroup and stay there for all of its work items, a stage is a whole
// item (18 MFMA steps, one barrier, no weight re-staging: half the staged bytes, half the barriers, twice the tile bytes in flight per prefetch).
// The weights are the MATH = 3 pack for 16-channel chunks read slab-major (step f = slab * 9 + tap), i.e. the products are accumulated in exactly
// the order of MATH = 3 with CK = 16: bit-identical results.
template <int KS, int STRIDE, int UPS, int TH, int TW, int CK, int WM, int WN, int MB, int NB, int PRO, int VEC, int EPI = 0, int ABL = 0, int MATH = 0>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvArgs a) {
    constexpr int NW = WM * WN, NTHR = 64 * NW;
    constexpr int PAD = KS / 2;
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr bool X3 = MATH >= 1;   // split-operand paths (bf16x3, f16x2): 16-bit planes in LDS
    constexpr bool F16 = MATH == 3 || MATH == 5;
    constexpr bool WR = MATH == 5;   // weights resident in LDS for the whole launch (host: n_chunks == 1, n_ct == 1, shared weights)
    constexpr bool B1 = MATH == 4;
    constexpr int NPL = F16 ? 2 : (B1 ? 1 : 3); // operand planes
    constexpr bool WSB = MATH == 2;  // ONE weight buffer (an extra barrier per stage): 69 KB of LDS -> two workgroups per CU, whose VALU
                                     // staging and bf16 MFMAs then overlap (different pipes)
    constexpr int NWB = (WSB || WR) ? 1 : 2;
    constexpr int PS = CK / 2;                            // X3: floats per bf16 plane of one staged pixel (CK x 2 B)
    constexpr int LDA = X3 ? NPL * PS + 4 : CK + 4;       // floats per staged pixel (X3: NPL planes + 16 B pad)
    constexpr int LDH = CK + 4;                           // row of the fp32 scratch tile of the depthwise prologue
    constexpr int TAPS = KS * KS;
    constexpr int K8 = CK / 8;
    constexpr int K16 = CK / 16;
    constexpr int C4 = CK / 4;
    constexpr int NF = X3 ? TAPS * K16 : TAPS * K8;       // MFMA steps per chunk: (tap, k8), or (tap, 16-channel slab)
    // row pad of the staged tile (floats): lanes 0-15 / 16-31 of an A-fragment read own consecutive pixels of two ROWS of the tile;
    // with a halo the row stride IW * LDA puts the second row on the banks of the first (2-way conflicts on every ds_read_b128 of
    // the 16-wide tile, 3-way on the 8-wide one); these pads make the reads conflict-free (bank search: DESIGN section 3)
    constexpr int RP = (X3 && KS == 3 && STRIDE == 1) ? (WR ? (64 - (IW * LDA) % 64) % 64 : (NPL == 2 ? 24 : (NPL == 1 ? 40 : 8))) : 0;  // WR (CK = 32, 16-wide tiles): row stride = 0 mod 64 banks
    constexpr int LDR = IW * LDA + RP;                    // floats per staged tile row
    constexpr int ABUF = IH * LDR;                        // floats per LDS buffer
    // PRO_GN_DW (1x1 conv over depthwise3x3(GroupNorm(x))): the LOAD tile has a one-pixel halo and goes to a scratch
    // LDS region; the depthwise conv turns it into the A tile of the 1x1 contraction.
    constexpr bool DWM = (PRO == PRO_GN_DW);
    constexpr bool GNP = (PRO == PRO_GN || PRO == PRO_GN_SILU || PRO == PRO_GN_DW);
    constexpr bool FILM = (EPI & EPI_FILM) != 0, SOUT = (EPI & EPI_SOUT) != 0, RES = (EPI & EPI_RES) != 0, SILU = (EPI & EPI_SILU) != 0, TBS = (EPI & EPI_TBS) != 0;
    constexpr bool SAMP = (EPI & EPI_SAMP) != 0;
    constexpr int NBX = (EPI & EPI_XF1) ? 1 : ((EPI & EPI_XF2) ? 2 : 0);  // 32-cout blocks of the folded x_conv
    constexpr bool XF = NBX > 0;
    static_assert(!SAMP || (!SOUT && !RES && !FILM), "sampler epilogue: the plain vector epilogue of the final conv");
    static_assert(!XF || (MB == 1 && NB == 1 && WN == 1 && !SOUT && !SAMP && !FILM && !SILU && PRO != PRO_GN_DW), "x_conv + FiLM fold: one wave holds all 32 couts of its pixels");
    static_assert(!DWM || (KS == 1 && STRIDE == 1 && !UPS && VEC == 1), "depthwise staging is for plain 1x1 convs");
    static_assert(PRO != PRO_COLSM || VEC == 1, "column-softmax prologue needs uniform-source float4 staging");
    constexpr int LPAD = DWM ? 1 : PAD;
    constexpr int LH = DWM ? TH + 2 : IH, LW = DWM ? TW + 2 : IW;   // extent of the loaded tile
    constexpr int HBUF = DWM ? LH * LW * LDH : 0;                  // scratch for the normalised halo tile (fp32)
    constexpr int DWMAX = DWM ? 9 * 256 : 0;                       // depthwise weights of up to 256 channels
    constexpr int DITEMS = (TH * TW * C4 + NTHR - 1) / NTHR;
    constexpr int NITEMS = (LH * LW * C4 + NTHR - 1) / NTHR;  // float4 input-staging items per thread and chunk
    constexpr int WCHUNK = X3 ? NF * NPL * 256 : NF * 256;  // floats per (32-cout block, chunk): X3 = NPL 16-bit planes of 1 KiB per step
    constexpr int WBUF = NB * WN * WCHUNK;                // floats of one weight chunk (all n-blocks of the cout tile)
    constexpr int WITEMS = WR ? 0 : (WBUF / 4 + NTHR - 1) / NTHR;  // float4 weight-staging items per thread and chunk (WR: none, the prologue copies them once)
    constexpr int WITEMS_A = WITEMS ? WITEMS : 1;
    constexpr int WRI = WR ? (WBUF / 4 + NTHR - 1) / NTHR : 1;  // WR: float4 items per thread of the one-time weight copy
    constexpr int DUMMY = DWM ? CK : (X3 ? NPL * PS : CK);  // pad slot of pixel 0 of the buffer the staging items go to (Hs for the
                                                          // depthwise prologue, else the A buffer): items past the end write here
    static_assert(NW == 4 || NW == 8, "4 or 8 wavefronts per workgroup");
    static_assert(!X3 || (CK % 16 == 0 && (VEC == 1 || (VEC == 2 && F16 && PRO == PRO_NONE))), "split-operand paths: 16-channel slabs, float4 staging (per-thread source select: the stem on f16x2)");
    static_assert(TH * TW == 32 * MB * WM, "pixel tile must match the wave layout");
    static_assert(CK % 8 == 0 && NTHR % C4 == 0, "chunk size");
    static_assert(NITEMS <= 32, "valid mask is 32 bits");
    static_assert(!F16 || PRO != PRO_COLSM, "f16x2: the column-softmax prologue stays on bf16x3 / fp32 (probabilities far below 2^-7)");
    static_assert(RP == 0 || !DWM, "row pad: 3x3 tiles only");
    static_assert(!WR || (KS == 3 && STRIDE == 1 && !UPS && CK == 32 && TW == 16 && !DWM), "resident weights: plain 3x3 convs with one 32-channel stage");

    dd_touch_kernargs<sizeof(ConvArgs)>();  // every line of the argument block in ONE round trip (ddif_dev.h)
    DDIF_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);   // [2][ABUF]  input halo tile of one channel chunk
    float* Ws = As + 2 * ABUF;                    // [NWB][WBUF]  weight chunk in B-fragment order
    double* red = reinterpret_cast<double*>(smem + (size_t)(2 * ABUF + NWB * WBUF) * sizeof(float));  // [2][2 * NW]
    float* Hs = reinterpret_cast<float*>(smem + (size_t)(2 * ABUF + NWB * WBUF) * sizeof(float) + 4 * NW * sizeof(double));  // [HBUF]
    float* DWs = Hs + HBUF;                       // [9][Ctot]
    [[maybe_unused]] double* redx = reinterpret_cast<double*>(DWs + DWMAX);  // XF: [2][2 * NW] partials of the second output
    float* GBs = DWs + DWMAX + (XF ? 8 * NW : 0); // GroupNorm gamma | beta, [2][n_chunks * CK] (host adds the bytes)
    float* BTs = GBs + (GNP ? 2 * a.n_chunks * CK : 0);  // bias (+ the step's time-bias row) of all n_ct * NT couts
    [[maybe_unused]] float* BXs = BTs + a.n_ct * (32 * NB * WN);  // XF: x_conv bias [32 * NBX]

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DDIF_EMU
    const int wave = tid >> 6;
    [[maybe_unused]] const long long t_entry = 0;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // SGPR: everything derived from it is scalar math
    [[maybe_unused]] const long long t_entry = (ABL & 16) ? (long long)__builtin_amdgcn_s_memtime() : 0;
#endif
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, j = lane & 31;
    const int tiles = a.tiles_x * a.tiles_y;
    const int ntiles = a.B * tiles;
    const int nwork = ntiles * a.n_ct;
    int w0, w1;
    wg_work_range(nwork, &w0, &w1, a.xcd);
    if (w0 >= w1) return;  // whole workgroup leaves together
    const int Hc = UPS ? a.Hin * 2 : a.Hin, Wc = UPS ? a.Win * 2 : a.Win;
    const int Ctot = a.c0 + a.c1;
    const int GBN = a.n_chunks * CK;
    const int c4 = tid % C4;
    const int stepv = a.step_ptr ? *a.step_ptr : 0;  // the load goes out here; first used at the table fill below (dd_late)
    const float* tbrow = a.tbias;                    // (+ the step's row offset from there on)
    // sampler epilogue: this step's coefficients (scalar loads, once per workgroup)
    [[maybe_unused]] SamplerRun s_run{};
    [[maybe_unused]] int s_k = 0;
    [[maybe_unused]] float s_c0 = 0.f, s_c1 = 0.f, s_c2 = 0.f, s_c3 = 0.f, s_c4 = 0.f;
    [[maybe_unused]] const float* s_noise = nullptr;
    if constexpr (SAMP) {
        s_run = *a.s_run;
        s_k = *a.step_ptr;
        s_c0 = s_run.tab[0][s_k];
        s_c1 = s_run.tab[1][s_k];
        s_c2 = s_run.tab[2][s_k];
        if (a.s_kind == 1) {
            s_c3 = s_run.tab[3][s_k];
            s_c4 = s_run.tab[4][s_k];
        }
        s_noise = s_run.noise ? s_run.noise + (size_t)s_k * ((size_t)a.B * a.Cout * a.Hout * a.Wout) : nullptr;
        if (blockIdx.x == 0 && tid == 0) *a.s_step_next = s_k + 1;  // the next step's kernels read the OTHER counter (nobody reads this one during this step)
    }

    int abase[MB], e_my[MB], e_mx[MB];
    unsigned e_off[MB], e_foff[MB];  // byte offset of this lane's pixel (+ 4h couts) from the tile's first output / FiLM element
    [[maybe_unused]] unsigned e_xoff[MB], e_xfoff[MB];  // XF: the same for the second output / its scale | shift tensor
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = (wm * MB + mb) * 32 + j;
        abase[mb] = (m / TW) * STRIDE * LDR + (m % TW) * STRIDE * LDA + 4 * h;
        e_my[mb] = m / TW;  // the pixel this lane owns in the (transposed) accumulator block mb
        e_mx[mb] = m % TW;
        e_off[mb] = (unsigned)(((e_my[mb] * a.Wout + e_mx[mb]) * a.Cout + 4 * h) * 4);
        e_foff[mb] = (unsigned)(((e_my[mb] * a.Wout + e_mx[mb]) * 2 * a.Cout + 4 * h) * 4);
        if constexpr (XF) {
            e_xoff[mb] = (unsigned)(((e_my[mb] * a.Wout + e_mx[mb]) * (32 * NBX) + 4 * h) * 4);
            e_xfoff[mb] = (unsigned)(((e_my[mb] * a.Wout + e_mx[mb]) * 2 * (32 * NBX) + 4 * h) * 4);
        }
    }
    // XF: this lane's A fragments of the register GEMM, resident for the whole launch: step s = 4g + i contracts cout 8g + 4h + i (k = h); lane (j, h) supplies W_x[32 nx + j][that cout]
    [[maybe_unused]] float xw[XF ? NBX : 1][16];
    if constexpr (XF) {
#pragma unroll
        for (int nx = 0; nx < NBX; ++nx)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(a.xf_w + ((size_t)(nx * 64 + lane) * 16 + 4 * q));
                xw[nx][4 * q + 0] = t.x;
                xw[nx][4 * q + 1] = t.y;
                xw[nx][4 * q + 2] = t.z;
                xw[nx][4 * q + 3] = t.w;
            }
    }

    // ---- per-thread staging geometry: constant for the whole kernel (no divisions inside the stage loop) ----
    int a_py[NITEMS], a_px[NITEMS], a_lds[NITEMS];
    unsigned a_in = 0;
#pragma unroll
    for (int it = 0; it < NITEMS; ++it) {
        const int pixr = (tid + it * NTHR) / C4;
        const bool in = pixr < LH * LW;
        const int pix = in ? pixr : LH * LW - 1;
        a_py[it] = pix / LW;
        a_px[it] = pix % LW;
        a_lds[it] = in ? (DWM ? pix * LDH + c4 * 4 : (X3 ? a_py[it] * LDR + a_px[it] * LDA + c4 * 2 : pix * LDA + c4 * 4)) : DUMMY;
        a_in |= (in ? 1u : 0u) << it;
    }
    unsigned w_boff[WITEMS_A];
    int w_lds[WITEMS_A];  // relative to Ws[buf]; items past the end go to the A buffer's dummy slot (negative)
#pragma unroll
    for (int it = 0; it < WITEMS; ++it) {
        const int qr = tid + it * NTHR;
        const bool in = qr < WBUF / 4;
        const int q = in ? qr : WBUF / 4 - 1;
        w_boff[it] = (unsigned)(((q / (WCHUNK / 4)) * (a.n_chunks * WCHUNK) + (q % (WCHUNK / 4)) * 4) * 4);
        w_lds[it] = in ? q * 4 : -1;
    }

    // ---- work-item positions (workgroup-uniform); one division set per ITEM, none per chunk ----
    struct Pos { int work, ct, b, oy0, ox0; };
    auto locate = [&](int work) {
        Pos p;
        p.work = work;
        const int pt = work / a.n_ct;  // cout tile FASTEST: consecutive items re-read the same input tile out of L2
        p.ct = work - pt * a.n_ct;
        const int bl = pt / tiles;
        p.b = a.b0 + bl;
        const int t = pt - bl * tiles;
        const int ty = t / a.tiles_x;
        p.oy0 = ty * TH;
        p.ox0 = (t - ty * a.tiles_x) * TW;
        return p;
    };
    auto next_pos = [&](Pos p) {  // work + 1 without divisions
        p.work += 1;
        if (++p.ct < a.n_ct) return p;
        p.ct = 0;
        p.ox0 += TW;
        if (p.ox0 >= a.tiles_x * TW) {
            p.ox0 = 0;
            p.oy0 += TH;
            if (p.oy0 >= a.tiles_y * TH) {
                p.oy0 = 0;
                ++p.b;
            }
        }
        return p;
    };

    // ---- loading side: runs TWO stages (channel chunks) ahead of the MFMAs, into two register sets ----
    // Measured on MI355X (tools/mbench.cpp, in-kernel clock stamps): with all workgroups of a launch prefetching in
    // bursts a global load takes ~5 us to land, a stage of MFMAs ~2-3 us: one stage of lookahead is not enough.
    Pos L = locate(w0);
    int l_ch = 0;
    [[maybe_unused]] int l_sp[(VEC == 1) ? 1 : NITEMS];  // VEC != 1: clamped source pixel index of every staging item of work item L
    [[maybe_unused]] unsigned l_o0[NITEMS], l_o1[NITEMS];  // VEC == 1: byte offset of the item's source pixel in source 0 / source 1
    unsigned l_cso[PRO == PRO_COLSM ? NITEMS : 1];  // PRO_COLSM: byte offset of (b, x) in the column-softmax statistics
    unsigned l_ok = 0;  // which items are real (inside the image): zero padding otherwise
    auto item_geometry = [&]() {
        const int iy0 = L.oy0 * STRIDE - LPAD, ix0 = L.ox0 * STRIDE - LPAD;
        l_ok = 0;
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) {
            const int iy = iy0 + a_py[it], ix = ix0 + a_px[it];
            const bool ok = (iy >= 0) & (iy < Hc) & (ix >= 0) & (ix < Wc);
            l_ok |= (ok ? 1u : 0u) << it;
            const int iyc = iy < 0 ? 0 : (iy >= Hc ? Hc - 1 : iy), ixc = ix < 0 ? 0 : (ix >= Wc ? Wc - 1 : ix);
            const int sp = (L.b * a.Hin + (UPS ? (iyc >> 1) : iyc)) * a.Win + (UPS ? (ixc >> 1) : ixc);
            if constexpr (VEC == 1) {
                l_o0[it] = (unsigned)sp * (unsigned)(a.c0 * 4);
                if (a.c1) l_o1[it] = (unsigned)sp * (unsigned)(a.c1 * 4);
            } else {
                l_sp[it] = sp;
            }
            if (PRO == PRO_COLSM) l_cso[it] = (unsigned)(L.b * a.Win + ixc) * (unsigned)(a.c0 * 4);
        }
        l_ok &= a_in;
    };

    struct StageRegs {
        float4 sv[NITEMS], wv[WITEMS_A];
        float4 mxv[PRO == PRO_COLSM ? NITEMS : 1], smv[PRO == PRO_COLSM ? NITEMS : 1];
        unsigned ok;
        int cb, ch;  // first channel of the chunk, chunk index
        Pos pos;
    };
    StageRegs R0, R1;
    Pos bufpos[2];  // work item staged in LDS buffer 0 / 1
    int bufch[2] = {0, 0};
    int gn_b = -1;
    float mean = 0.f, rstd = 1.f;

    // Branch-free and UNCONDITIONAL: every item loads from a clamped (always valid) address; validity is a bit mask
    // applied when the tile is written to LDS.  Past the end of the work range the loader re-reads the last item (L2
    // hits, never consumed): a conditional prefetch would make the number of loads in flight path-dependent, and every
    // later counted vmcnt wait would degrade to vmcnt(0).
    auto issue_loads = [&](StageRegs& R) {
        const int cb = l_ch * CK;
        if (ABL & 2) {
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) R.sv[it] = make_float4(0.5f, 0.25f, -0.5f, 0.125f);
        } else if constexpr (VEC == 1) {
            // uniform source: SGPR base + 32-bit VGPR offset (tensors are < 4 GiB, checked by the host)
            const bool s0 = cb < a.c0;
            const int nvalid = (s0 ? a.c0 : Ctot) - cb;  // channels of this chunk that exist (>= 4, multiple of 4)
            const int c4c = c4 * 4 < nvalid ? c4 * 4 : nvalid - 4;  // channels past the end re-read valid ones (their weights are 0)
            const char* base = reinterpret_cast<const char*>(s0 ? a.in0 + cb : a.in1 + (cb - a.c0));
            const unsigned co = (unsigned)c4c * 4u;
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) {
                const unsigned off = ((s0 || !a.c1) ? l_o0[it] : l_o1[it]) + co;
                R.sv[it] = *reinterpret_cast<const float4*>(base + off);
            }
        } else if constexpr (VEC == 2) {
            const int cbase = cb + c4 * 4;
            const int cc = cbase < Ctot ? cbase : Ctot - 4;
            const bool s0 = cc < a.c0;
            const float* base = s0 ? a.in0 + cc : a.in1 + (cc - a.c0);
            const int cs = s0 ? a.c0 : a.c1;
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) R.sv[it] = *reinterpret_cast<const float4*>(base + (size_t)l_sp[it] * cs);
        } else {
            const int cbase = cb + c4 * 4;
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) {
                float e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = cbase + i < Ctot ? cbase + i : Ctot - 1;
                    const bool s0 = c < a.c0;
                    const float* base = s0 ? a.in0 + c : a.in1 + (c - a.c0);
                    e[i] = base[(size_t)l_sp[it] * (s0 ? a.c0 : a.c1)];
                }
                R.sv[it] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        if (PRO == PRO_COLSM) {  // softmax_H(q) statistics of the channels of source 0 (c0 is a multiple of CK)
            const int cbc = cb < a.c0 ? cb : 0;
            const char* bmx = reinterpret_cast<const char*>(a.cs_mx + cbc);
            const char* bsm = reinterpret_cast<const char*>(a.cs_sm + cbc);
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) {
                R.mxv[it] = *reinterpret_cast<const float4*>(bmx + (l_cso[it] + (unsigned)c4 * 16u));
                R.smv[it] = *reinterpret_cast<const float4*>(bsm + (l_cso[it] + (unsigned)c4 * 16u));
            }
        }
        const char* wbase = reinterpret_cast<const char*>(a.w + (size_t)L.b * a.w_bstride + ((size_t)L.ct * (NB * WN) * a.n_chunks + l_ch) * WCHUNK);
#pragma unroll
        for (int it = 0; it < WITEMS; ++it) {
            if (ABL & 8) R.wv[it] = make_float4(0.01f, 0.02f, 0.03f, 0.04f);
            else R.wv[it] = *reinterpret_cast<const float4*>(wbase + w_boff[it]);
        }
        R.ok = l_ok;
        R.cb = cb;
        R.ch = l_ch;
        R.pos = L;
        // advance the loader
        if (++l_ch == a.n_chunks) {
            l_ch = 0;
            if (L.work + 1 < w1) {
                const bool same_tile = L.ct + 1 < a.n_ct;
                L = next_pos(L);
                if (!same_tile) item_geometry();
            }
        }
    };
    // finish_stage = fs_begin; fs_item x NITEMS; [depthwise pass]; fs_witem x WITEMS; fs_end.  The pieces exist so that the bf16x3
    // MFMA loop can carry them BETWEEN its taps: v_mfma_f32_32x32x16_bf16 runs on the matrix pipe beside the VALU / LDS work of the same
    // and of the partner wave (tools/probes/mfma_coexec.cpp: a bf16 MFMA stream slows a v_fma / v_exp / ds_read stream on the same SIMD
    // by < 15 % and is not slowed itself), and the six dependent MFMAs of a tap leave ~40 idle issue slots per wave.
    float fs_ga[4] = {0.f, 0.f, 0.f, 0.f}, fs_gb[4] = {0.f, 0.f, 0.f, 0.f};
    bool fs_colsm = false;
    constexpr bool RANGE = F16 && !GNP && !DWM;  // raw activations x 2^4 into halves: watch the range (ConvArgs::range_flag)
    [[maybe_unused]] float r_max = 0.f;         // largest |scaled value| this thread staged
    auto fs_begin = [&](StageRegs& R) {
        if (GNP) {
            if (R.pos.b != gn_b) {  // workgroup-uniform; every wavefront reduces the partials itself (no barrier)
                gn_finalize_wave(a.st0, a.np0, a.st1, a.np1, R.pos.b, (double)Ctot * a.Hin * a.Win, &mean, &rstd);
                gn_b = R.pos.b;
            }
            const float4 gq = *reinterpret_cast<const float4*>(&GBs[R.cb + c4 * 4]);
            const float4 bq = *reinterpret_cast<const float4*>(&GBs[GBN + R.cb + c4 * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fs_ga[i] = (&gq.x)[i] * rstd;
                fs_gb[i] = (&bq.x)[i] - mean * fs_ga[i];
            }
        }
        fs_colsm = (PRO == PRO_COLSM) && R.cb < a.c0;
    };
    // channels past the end of a partial last chunk hold duplicates of real (finite) data: their packed weights are 0
    auto fs_item = [&](StageRegs& R, int buf, int it) {
        if ((ABL & 256) && R.sv[it].x != 12345.678f) return;  // ablation: no prologue math, no LDS writes of the tile
        float* dst = As + buf * ABUF;
        const bool ok = (R.ok >> it) & 1u;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x = (&R.sv[it].x)[i];
            if (GNP) {
                x = fmaf(x, fs_ga[i], fs_gb[i]);
                if (PRO == PRO_GN_SILU) x = (F16 && !DWM) ? dd_silu_scaled(x, 1.0f / DDIF_F16_ASCALE) : dd_silu(x);
                else if (F16 && !DWM) x *= DDIF_F16_ASCALE;
            } else if (F16 && !DWM) {
                x *= DDIF_F16_ASCALE;
            }
            if (PRO == PRO_COLSM) {
                if (fs_colsm) x = dd_exp2_fast((x - (&R.mxv[it].x)[i]) * 1.4426950408889634f) * dd_rcp_fast((&R.smv[it].x)[i]);
            }
            v[i] = ok ? x : 0.f;  // zero padding comes AFTER the activation
        }
        if constexpr (RANGE) r_max = fmaxf(fmaxf(r_max, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        if (DWM) {
            *reinterpret_cast<float4*>(&Hs[a_lds[it]]) = make_float4(v[0], v[1], v[2], v[3]);
            // centre pixels inside the image: this IS xn = GroupNorm(cat[h, skip]) (attn_res input)
            if (a.out_xn && ok && a_py[it] >= 1 && a_py[it] <= TH && a_px[it] >= 1 && a_px[it] <= TW)
                *reinterpret_cast<float4*>(a.out_xn + ((size_t)((R.pos.b * a.Hin + R.pos.oy0 + a_py[it] - 1) * a.Win + R.pos.ox0 + a_px[it] - 1)) * Ctot + R.cb + c4 * 4) =
                    make_float4(v[0], v[1], v[2], v[3]);
        } else if constexpr (B1) {
            *reinterpret_cast<uint2*>(&dst[a_lds[it]]) = make_uint2(dd_bf16_pair(v[0], v[1]), dd_bf16_pair(v[2], v[3]));
        } else if constexpr (F16) {
            unsigned h01, l01, h23, l23;
            dd_split2_pair(v[0], v[1], &h01, &l01);
            dd_split2_pair(v[2], v[3], &h23, &l23);
            const int ps = ((a_in >> it) & 1u) ? PS : 0;  // plane stride in floats (the dummy slot takes both)
            *reinterpret_cast<uint2*>(&dst[a_lds[it]]) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(&dst[a_lds[it] + ps]) = make_uint2(l01, l23);
        } else if constexpr (X3) {
            unsigned h01, m01, l01, h23, m23, l23;
            dd_split3_pair(v[0], v[1], &h01, &m01, &l01);
            dd_split3_pair(v[2], v[3], &h23, &m23, &l23);
            const int ps = ((a_in >> it) & 1u) ? PS : 0;  // plane stride in floats (the dummy slot takes all three)
            *reinterpret_cast<uint2*>(&dst[a_lds[it]]) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(&dst[a_lds[it] + ps]) = make_uint2(m01, m23);
            *reinterpret_cast<uint2*>(&dst[a_lds[it] + 2 * ps]) = make_uint2(l01, l23);
        } else {
            *reinterpret_cast<float4*>(&dst[a_lds[it]]) = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    auto fs_depthwise = [&](StageRegs& R, int buf) {
        float* dst = As + buf * ABUF;
        if (DWM) {
            __syncthreads();  // halo tile complete in Hs
#pragma unroll
            for (int it = 0; it < DITEMS; ++it) {
                const int item = tid + it * NTHR;
                const int p = item / C4;
                if (p < TH * TW) {
                    const int ty = p / TW, tx = p % TW;
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const float4 hv = *reinterpret_cast<const float4*>(&Hs[((ty + k / 3) * LW + tx + k % 3) * LDH + c4 * 4]);
                        const float4 wk = *reinterpret_cast<const float4*>(&DWs[k * Ctot + R.cb + c4 * 4]);
                        s0 = fmaf(hv.x, wk.x, s0);
                        s1 = fmaf(hv.y, wk.y, s1);
                        s2 = fmaf(hv.z, wk.z, s2);
                        s3 = fmaf(hv.w, wk.w, s3);
                    }
                    if constexpr (B1) {
                        *reinterpret_cast<uint2*>(&dst[p * LDA + c4 * 2]) = make_uint2(dd_bf16_pair(s0, s1), dd_bf16_pair(s2, s3));
                    } else if constexpr (F16) {
                        unsigned h01, l01, h23, l23;
                        dd_split2_pair(s0 * DDIF_F16_ASCALE, s1 * DDIF_F16_ASCALE, &h01, &l01);
                        dd_split2_pair(s2 * DDIF_F16_ASCALE, s3 * DDIF_F16_ASCALE, &h23, &l23);
                        *reinterpret_cast<uint2*>(&dst[p * LDA + c4 * 2]) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(&dst[p * LDA + PS + c4 * 2]) = make_uint2(l01, l23);
                    } else if constexpr (X3) {
                        unsigned h01, m01, l01, h23, m23, l23;
                        dd_split3_pair(s0, s1, &h01, &m01, &l01);
                        dd_split3_pair(s2, s3, &h23, &m23, &l23);
                        *reinterpret_cast<uint2*>(&dst[p * LDA + c4 * 2]) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(&dst[p * LDA + PS + c4 * 2]) = make_uint2(m01, m23);
                        *reinterpret_cast<uint2*>(&dst[p * LDA + 2 * PS + c4 * 2]) = make_uint2(l01, l23);
                    } else {
                        *reinterpret_cast<float4*>(&dst[p * LDA + c4 * 4]) = make_float4(s0, s1, s2, s3);
                    }
                }
            }
        }
    };
    auto fs_witem = [&](StageRegs& R, int buf, int it) {
        float* dst = As + buf * ABUF;
        float* wdst = Ws + ((WSB || WR) ? 0 : buf) * WBUF;
        float* wp = w_lds[it] >= 0 ? wdst + w_lds[it] : dst + DUMMY;
        *reinterpret_cast<float4*>(wp) = R.wv[it];
    };
    auto fs_end = [&](StageRegs& R, int buf) {
        bufpos[buf] = R.pos;
        bufch[buf] = R.ch;
    };
    auto finish_stage = [&](StageRegs& R, int buf) {
        fs_begin(R);
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) fs_item(R, buf, it);
        fs_depthwise(R, buf);
#pragma unroll
        for (int it = 0; it < WITEMS; ++it) fs_witem(R, buf, it);
        fs_end(R, buf);
    };
    // interleaved form: piece p of NP = NITEMS + WITEMS follows MFMA step p * NF / NP
#ifdef DDIF_NO_ILV
    constexpr bool ILV = false;
#else
    constexpr bool ILV = X3 && !DWM && !WSB;
#endif
    constexpr int NP = NITEMS + WITEMS;

    bool pend = false;  // a statistics partial of work item pend_pos sits in red[pend_par]
    Pos pend_pos = L;
    int pend_par = 0;
    auto flush_stats = [&]() {
        if (a.st_out && pend && tid == 0) {
            const Pos p = pend_pos;
            const int t = (p.oy0 / TH) * a.tiles_x + p.ox0 / TW;
            const double* r = red + pend_par * 2 * NW;
            if (NW == 8 && TH == 16 && a.st_halves) {
                // one partial per 8 x 16 half tile: waves 0-3 = rows 0-7, waves 4-7 = rows 8-15, each summed as the four-wave tiling sums its tile
                const int ty = p.oy0 / TH, tx = p.ox0 / TW;
                const size_t np8 = (size_t)a.tiles_x * a.tiles_y8 * a.n_ct;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int ty8 = 2 * ty + hf;
                    if (ty8 < a.tiles_y8) {  // (a bottom half below the image has no tile in the 8 x 16 tiling)
                        const size_t pi = ((size_t)p.b * np8 + ((size_t)ty8 * a.tiles_x + tx) * a.n_ct + p.ct) * 2;
                        const double* q = r + 8 * hf;
                        a.st_out[pi + 0] = (q[0] + q[2]) + (q[4] + q[6]);
                        a.st_out[pi + 1] = (q[1] + q[3]) + (q[5] + q[7]);
                        if constexpr (XF) {
                            const double* qx = redx + pend_par * 2 * NW + 8 * hf;
                            a.xf_st[pi + 0] = (qx[0] + qx[2]) + (qx[4] + qx[6]);
                            a.xf_st[pi + 1] = (qx[1] + qx[3]) + (qx[5] + qx[7]);
                        }
                    }
                }
            } else {
            const size_t pi = ((size_t)p.b * (tiles * a.n_ct) + (size_t)t * a.n_ct + p.ct) * 2;
            double t0 = (r[0] + r[2]) + (r[4] + r[6]), t1 = (r[1] + r[3]) + (r[5] + r[7]);
            if (NW == 8) {
                t0 += (r[8] + r[10]) + (r[12] + r[14]);
                t1 += (r[9] + r[11]) + (r[13] + r[15]);
            }
            a.st_out[pi + 0] = t0;
            a.st_out[pi + 1] = t1;
            if constexpr (XF) {  // the second output's partial of the same item (n_ct = 1)
                const double* rx = redx + pend_par * 2 * NW;
                double u0 = (rx[0] + rx[2]) + (rx[4] + rx[6]), u1 = (rx[1] + rx[3]) + (rx[5] + rx[7]);
                if (NW == 8) {
                    u0 += (rx[8] + rx[10]) + (rx[12] + rx[14]);
                    u1 += (rx[9] + rx[11]) + (rx[13] + rx[15]);
                }
                a.xf_st[pi + 0] = u0;
                a.xf_st[pi + 1] = u1;
            }
            }
        }
        pend = false;
    };

    [[maybe_unused]] int dbg_n = 0;
    auto stamp = [&]() {
#ifndef DDIF_EMU
        if ((ABL & 16) && a.dbg && tid == 0 && dbg_n < 126) a.dbg[blockIdx.x * 128 + dbg_n++] = (long long)__builtin_amdgcn_s_memtime();
#endif
    };
#ifndef DDIF_EMU
    if ((ABL & 16) && a.dbg && tid == 0) a.dbg[blockIdx.x * 128 + dbg_n++] = t_entry;  // kernel entry; the next stamp = per-thread geometry set up
#endif
    f32x16 acc[MB][NB];
    const int nflat = (w1 - w0) * a.n_chunks;

    // One pipeline stage out of LDS buffer `cur`:
    //   [LAST: epilogue-operand loads]  prefetch of stage +2 into Rf  MFMAs  [LAST: epilogue]  stage +1 (Rn, loaded one
    //   stage ago) -> prologue -> LDS buffer cur^1.  The caller puts one barrier after it.
    auto stage = [&](auto kind, const int cur, StageRegs& Rn, StageRegs& Rf) {
        constexpr bool LAST = decltype(kind)::LAST, TAIL = decltype(kind)::TAIL;
        stamp();
        const float* Ac = As + cur * ABUF;
        const Pos Cp = bufpos[cur];
        const int c_ch = bufch[cur];
        const int nbg0 = (Cp.ct * WN + wn) * NB;
        // (1) epilogue operands FIRST: older than the prefetch below, so the epilogue's counted vmcnt wait leaves the
        //     prefetch in flight.  Nothing here may be COMPUTED on before the MFMAs (that would pull the wait up).
        float4 e_t[(LAST && TBS) ? NB : 1][4];
        float4 e_res[(LAST && RES) ? MB : 1][(LAST && RES) ? NB : 1][4];
        float4 e_fs[(LAST && FILM) ? MB : 1][(LAST && FILM) ? NB : 1][4], e_fh[(LAST && FILM) ? MB : 1][(LAST && FILM) ? NB : 1][4];
        [[maybe_unused]] float4 e_xi[(LAST && SAMP) ? MB : 1][(LAST && SAMP) ? NB : 1][4], e_xl[(LAST && SAMP) ? MB : 1][(LAST && SAMP) ? NB : 1][4];  // x_t, lms
        [[maybe_unused]] float4 e_xs[(LAST && XF) ? NBX : 1][4], e_xh[(LAST && XF) ? NBX : 1][4];  // XF: scale / shift of the second output (MB = 1)
        [[maybe_unused]] unsigned e_pxo = 0;  // XF: this item's byte offset of the lane's pixel in the second output
        bool full = true;
        unsigned e_po[MB], e_pf[MB];  // this item's byte offsets (clamped to the tile origin for pixels outside the image)
        bool e_pok[MB];
        size_t tile_el = 0;           // element index of (tile origin pixel, first cout of this wave)
        if constexpr (LAST) {
            full = (Cp.oy0 + TH <= a.Hout) & (Cp.ox0 + TW <= a.Wout);
            const size_t tile_pix = (size_t)((Cp.b * a.Hout + Cp.oy0) * a.Wout + Cp.ox0);
            tile_el = tile_pix * a.Cout + nbg0 * 32;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                e_pok[mb] = full || ((Cp.oy0 + e_my[mb] < a.Hout) & (Cp.ox0 + e_mx[mb] < a.Wout));
                e_po[mb] = e_pok[mb] ? e_off[mb] : (unsigned)(16 * h);
                e_pf[mb] = e_pok[mb] ? e_foff[mb] : (unsigned)(16 * h);
            }
            if constexpr (!SOUT) {
                // a wave whose whole cout range lies past Cout (Cout = 32 under a 64-cout tile) loads from block 0 instead
                const int nbl = nbg0 * 32 < a.Cout ? nbg0 * 32 : 0;
                [[maybe_unused]] const char* tb = reinterpret_cast<const char*>(tbrow + (size_t)Cp.b * a.tbias_stride + nbl);
                [[maybe_unused]] const char* rbase = reinterpret_cast<const char*>(a.res + tile_pix * a.Cout + nbl);
                [[maybe_unused]] const char* fbase = reinterpret_cast<const char*>(a.film + tile_pix * 2 * a.Cout + nbl);
                [[maybe_unused]] const char* xibase = reinterpret_cast<const char*>(a.s_img + tile_pix * a.Cout + nbl);
                [[maybe_unused]] const char* xlbase = reinterpret_cast<const char*>(a.s_lms + tile_pix * a.Cout + nbl);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // couts past Cout re-read the first quad of the (clamped) base block: always inside the tensor
                        const unsigned cq = ((nbg0 + nb) * 32 + 8 * g + 4 * h < a.Cout) ? (unsigned)((nb * 32 + 8 * g) * 4) : (unsigned)(-16 * h);
                        if constexpr (TBS) e_t[nb][g] = *reinterpret_cast<const float4*>(tb + (cq + 16u * h));
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) {
                            if constexpr (RES) e_res[mb][nb][g] = *reinterpret_cast<const float4*>(rbase + (e_po[mb] + cq));
                            if constexpr (SAMP) {
                                e_xi[mb][nb][g] = *reinterpret_cast<const float4*>(xibase + (e_po[mb] + cq));
                                e_xl[mb][nb][g] = *reinterpret_cast<const float4*>(xlbase + (e_po[mb] + cq));
                            }
                            if constexpr (FILM) {
                                e_fs[mb][nb][g] = *reinterpret_cast<const float4*>(fbase + (e_pf[mb] + cq));
                                e_fh[mb][nb][g] = *reinterpret_cast<const float4*>(fbase + (size_t)a.Cout * 4 + (e_pf[mb] + cq));
                            }
                        }
                    }
                if constexpr (XF) {
                    const char* xfb = reinterpret_cast<const char*>(a.xf_film + tile_pix * 2 * (32 * NBX));
                    const unsigned pf = e_pok[0] ? e_xfoff[0] : (unsigned)(16 * h);
                    e_pxo = e_pok[0] ? e_xoff[0] : (unsigned)(16 * h);
#pragma unroll
                    for (int nx = 0; nx < NBX; ++nx)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            e_xs[nx][g] = *reinterpret_cast<const float4*>(xfb + (pf + (unsigned)((nx * 32 + 8 * g) * 4)));
                            e_xh[nx][g] = *reinterpret_cast<const float4*>(xfb + (size_t)(32 * NBX) * 4 + (pf + (unsigned)((nx * 32 + 8 * g) * 4)));
                        }
                }
            }
        }
        // (2) prefetch stage +2
        if constexpr (!TAIL) issue_loads(Rf);
        flush_stats();
        stamp();
        if (c_ch == 0) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
        }
        if constexpr (ILV && !TAIL) fs_begin(Rn);
        // (3) contraction over taps x chunk channels: both fragments from LDS, no global load in here.  Weights are
        //     the MFMA's FIRST operand: D[cout][pixel] (see the header comment).
        const float* Wc = Ws + ((WSB || WR) ? 0 : cur) * WBUF + (wn * NB) * WCHUNK + h * 128 + j * 4;
        if constexpr (X3) {
            //     bf16x3: per tap one 16-channel slab; three planes per operand, six cross products, small terms first;
            //     the fragments of tap t+1 are read before the MFMAs of tap t (register double buffer)
            constexpr int FB = (MB * NB >= 2) ? 1 : 2;  // register double buffer only for the single-tile shapes (wide cout tiles would spill)
            float4 xa[FB][MB][NPL], wb[FB][NB][NPL];
            auto load_frags3 = [&](int f, int slot) {  // step f = (tap, 16-channel slab)
                // WR: the pack is [16-channel chunk][tap] (the MATH = 3, CK = 16 pack), so step f walks slab-major -- the accumulation order of MATH = 3
                const int tap = WR ? f % TAPS : f / K16, k16 = WR ? f / TAPS : f % K16;
                const int aoff = (tap / KS) * LDR + (tap % KS) * LDA + k16 * 8;
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        xa[slot][mb][q] = (ABL & 128) ? make_float4(1.f, 2.f, 3.f, (float)(f + q)) : *reinterpret_cast<const float4*>(&Ac[abase[mb] + aoff + q * PS]);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        wb[slot][nb][q] = (ABL & 128) ? make_float4(4.f, 3.f, 2.f, (float)(f - q)) : *reinterpret_cast<const float4*>(&Wc[nb * WCHUNK + (f * NPL + q) * 256]);
                }
            };
            if (FB == 2) load_frags3(0, 0);
#pragma unroll
            for (int tap = 0; tap < NF; ++tap) {
                if (FB == 2) {
                    if (tap + 1 < NF) load_frags3(tap + 1, (tap + 1) & 1);
                } else {
                    load_frags3(tap, 0);
                }
                DDIF_SCHED_FENCE();
                const int sl = FB == 2 ? (tap & 1) : 0;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        f32x16 c = acc[mb][nb];
                        if (ABL & 1) {  // no matrix work: keep the fragments alive
                            c[0] += wb[sl][nb][0].x + xa[sl][mb][0].y + wb[sl][nb][NPL / 2].z + xa[sl][mb][NPL / 2].w + wb[sl][nb][NPL - 1].x + xa[sl][mb][NPL - 1].y;
                            acc[mb][nb] = c;
                            continue;
                        }
                        if constexpr (B1) {
                            acc[mb][nb] = DDIF_MFMA_32x32x16_BF16(wb[sl][nb][0], xa[sl][mb][0], c);
                            continue;
                        }
                        if constexpr (F16) {
                            c = DDIF_MFMA_32x32x16_F16(wb[sl][nb][NPL - 1], xa[sl][mb][0], c);  // lo * hi
                            c = DDIF_MFMA_32x32x16_F16(wb[sl][nb][0], xa[sl][mb][NPL - 1], c);  // hi * lo
                            c = DDIF_MFMA_32x32x16_F16(wb[sl][nb][0], xa[sl][mb][0], c);  // hi * hi
                            acc[mb][nb] = c;
                            continue;
                        }
                        if constexpr (NPL == 3) {
                            c = DDIF_MFMA_32x32x16_BF16(wb[sl][nb][2], xa[sl][mb][0], c);  // lo * hi
                            c = DDIF_MFMA_32x32x16_BF16(wb[sl][nb][0], xa[sl][mb][2], c);  // hi * lo
                            c = DDIF_MFMA_32x32x16_BF16(wb[sl][nb][1], xa[sl][mb][1], c);  // mid * mid
                            c = DDIF_MFMA_32x32x16_BF16(wb[sl][nb][1], xa[sl][mb][0], c);  // mid * hi
                            c = DDIF_MFMA_32x32x16_BF16(wb[sl][nb][0], xa[sl][mb][1], c);  // hi * mid
                            c = DDIF_MFMA_32x32x16_BF16(wb[sl][nb][0], xa[sl][mb][0], c);  // hi * hi
                        }
                        acc[mb][nb] = c;
                    }
                DDIF_SCHED_FENCE();
                if constexpr (ILV && !TAIL) {  // stage +1's pieces that belong behind this step
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        if (p * NF / NP == tap) {
                            if (p < NITEMS) fs_item(Rn, cur ^ 1, p);
                            else fs_witem(Rn, cur ^ 1, p - NITEMS);
                        }
                    DDIF_SCHED_FENCE();
                }
            }
        } else {
        //     Fragments are double-buffered in registers: the ds_reads of step f+1 are issued BEFORE the MFMAs of step f
        //     (an in-order wave otherwise issues them only after the last MFMA of step f has issued, and the LDS
        //     latency beyond that MFMA's 64 cycles is a bubble in the matrix pipe).
        float4 af[2][MB], bf[2][NB];
        auto load_frags = [&](int f, int slot) {
            const int tap = f / K8, k8 = f % K8;
            const int aoff = ((tap / KS) * IW + (tap % KS)) * LDA;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) af[slot][mb] = *reinterpret_cast<const float4*>(&Ac[abase[mb] + aoff + k8 * 8]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bf[slot][nb] = *reinterpret_cast<const float4*>(&Wc[nb * WCHUNK + f * 256]);
        };
        load_frags(0, 0);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (f + 1 < NF) load_frags(f + 1, (f + 1) & 1);
            DDIF_SCHED_FENCE();  // keep the reads above the MFMAs (the scheduler otherwise sinks them to their first use)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        if (ABL & 1) acc[mb][nb][i] += (&af[f & 1][mb].x)[i] * (&bf[f & 1][nb].x)[i];
                        else acc[mb][nb] = DDIF_MFMA_32x32x2((&bf[f & 1][nb].x)[i], (&af[f & 1][mb].x)[i], acc[mb][nb]);
            DDIF_SCHED_FENCE();
        }
        }
        if constexpr (WSB && !TAIL) __syncthreads();  // every wave is done reading the single weight buffer
        stamp();
        if constexpr (LAST) {
            // (4) epilogue of work item Cp: lane (j, h) owns pixel j of each 32-pixel block and, per accumulator quad g,
            //     the 4 consecutive couts 8g + 4h .. +3 of each 32-cout block
            float s1 = 0.f, s2 = 0.f;
            [[maybe_unused]] float sx1 = 0.f, sx2 = 0.f;  // XF: statistics of the second output
            if constexpr (!SOUT) {
                char* obase = reinterpret_cast<char*>(a.out + tile_el);
                auto epi = [&](auto guard) {
                    constexpr bool GUARD = decltype(guard)::LAST;  // StageKind<1> = bounds-checked stores
                    [[maybe_unused]] float vx[16];  // XF: the 16 output values of this lane = the B operand of the register GEMM
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int co = (nbg0 + nb) * 32 + 8 * g + 4 * h;
                                const float4 bt = *reinterpret_cast<const float4*>(&BTs[co]);
                                float v[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    float x = F16 ? fmaf(acc[mb][nb][4 * g + i], DDIF_F16_OSCALE, (&bt.x)[i]) : acc[mb][nb][4 * g + i] + (&bt.x)[i];
                                    if constexpr (TBS) x += (&e_t[nb][g].x)[i];
                                    if constexpr (FILM) x = x * (1.f + (&e_fs[mb][nb][g].x)[i]) + (&e_fh[mb][nb][g].x)[i];
                                    if constexpr (SILU) x = dd_silu(x);
                                    if constexpr (RES) x += (&e_res[mb][nb][g].x)[i];
                                    v[i] = x;
                                    if constexpr (XF) vx[4 * g + i] = x;
                                }
                                if constexpr (SAMP) {
                                    if (e_pok[mb] && co < a.Cout) {
#pragma clang fp contract(off)
                                        // the sampler update on x0 = v (same expressions, same order as ddpm_step_kernel / ddim_step_kernel: no contraction)
                                        const size_t pix = (size_t)((Cp.b * a.Hout + Cp.oy0 + e_my[mb]) * a.Wout + Cp.ox0 + e_mx[mb]);
                                        const size_t hw = (size_t)a.Hout * a.Wout, pin = pix - (size_t)Cp.b * hw;
                                        float o4[4];
#pragma unroll
                                        for (int i = 0; i < 4; ++i) {
                                            float x0 = v[i];
                                            const float l = (&e_xl[mb][nb][g].x)[i], xi = (&e_xi[mb][nb][g].x)[i];
                                            if (s_run.do_clamp) x0 = fminf(fmaxf(x0 + l, s_run.lo), s_run.hi) - l;
                                            const size_t e = ((size_t)Cp.b * a.Cout + co + i) * hw + pin;  // NCHW element index: the noise layout / Philox key
                                            if (a.s_kind == 0) {
                                                const float z = s_noise ? s_noise[e] : philox_normal(s_run.seed, (unsigned)(s_k + 1), (s_run.tile0 * a.Cout * hw) + e);
                                                const float mean = s_c0 * x0 + s_c1 * xi;
                                                o4[i] = mean + s_c2 * z;
                                            } else {
                                                const float eps = (s_c0 * xi - x0) / s_c1;
                                                float r = x0 * s_c2 + s_c3 * eps;
                                                if (s_c4 != 0.f) {
                                                    const float z = s_noise ? s_noise[e] : philox_normal(s_run.seed, (unsigned)(s_k + 1), (s_run.tile0 * a.Cout * hw) + e);
                                                    r += s_c4 * z;
                                                }
                                                o4[i] = r;
                                            }
                                        }
                                        *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.s_out + tile_el) + (e_po[mb] + (unsigned)((nb * 32 + 8 * g) * 4))) = make_float4(o4[0], o4[1], o4[2], o4[3]);
                                    }
                                } else if (!GUARD || (e_pok[mb] && co < a.Cout)) {
                                    // (streaming / nontemporal stores measured slower: 1x1 64->64 @64^2 63 vs 38 us)
                                    if (!(ABL & 4) || v[0] == 12345.678f)
                                        *reinterpret_cast<float4*>(obase + (e_po[mb] + (unsigned)((nb * 32 + 8 * g) * 4))) = make_float4(v[0], v[1], v[2], v[3]);
                                    s1 += (v[0] + v[1]) + (v[2] + v[3]);
                                    s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                                }
                            }
                    if constexpr (XF) {
                        // y = (W_x out + b_x) * (1 + scale) + shift on the values just stored (pixels outside the image carry finite garbage in their own column only)
                        char* xobase = reinterpret_cast<char*>(a.xf_out + (size_t)((Cp.b * a.Hout + Cp.oy0) * a.Wout + Cp.ox0) * (32 * NBX));
#pragma unroll
                        for (int nx = 0; nx < NBX; ++nx) {
                            f32x16 c2;
#pragma unroll
                            for (int r = 0; r < 16; ++r) c2[r] = 0.f;
#pragma unroll
                            for (int sx = 0; sx < 16; ++sx) c2 = DDIF_MFMA_32x32x2(xw[nx][sx], vx[sx], c2);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const float4 bx = *reinterpret_cast<const float4*>(&BXs[nx * 32 + 8 * g + 4 * h]);
                                float y[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) y[i] = (c2[4 * g + i] + (&bx.x)[i]) * (1.f + (&e_xs[nx][g].x)[i]) + (&e_xh[nx][g].x)[i];
                                if (!GUARD || e_pok[0]) {
                                    *reinterpret_cast<float4*>(xobase + (e_pxo + (unsigned)((nx * 32 + 8 * g) * 4))) = make_float4(y[0], y[1], y[2], y[3]);
                                    sx1 += (y[0] + y[1]) + (y[2] + y[3]);
                                    sx2 += (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
                                }
                            }
                        }
                    }
                };
                if (full && (nbg0 + NB) * 32 <= a.Cout) epi(StageKind<0>{});
                else epi(StageKind<1>{});
            } else {
                // scalar slow path (Cout not a multiple of 4): operands are loaded here, latency exposed -- rare
                [[maybe_unused]] const float* tb = tbrow + (size_t)Cp.b * a.tbias_stride;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int co = (nbg0 + nb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (e_pok[mb] && co < a.Cout) {
                                const size_t pix = (size_t)((Cp.b * a.Hout + Cp.oy0 + e_my[mb]) * a.Wout + Cp.ox0 + e_mx[mb]);
                                float x = F16 ? fmaf(acc[mb][nb][r], DDIF_F16_OSCALE, BTs[co]) : acc[mb][nb][r] + BTs[co];
                                if constexpr (TBS) x += tb[co];
                                if constexpr (FILM) x = x * (1.f + a.film[pix * 2 * a.Cout + co]) + a.film[pix * 2 * a.Cout + a.Cout + co];
                                if constexpr (SILU) x = dd_silu(x);
                                if constexpr (RES) x += a.res[pix * a.Cout + co];
                                if (!(ABL & 4) || x == 12345.678f) a.out[pix * a.Cout + co] = x;
                                s1 += x;
                                s2 += x * x;
                            }
                        }
            }
            if (a.st_out) {
                const double d1 = (double)wave_sum_fast(s1), d2 = (double)wave_sum_fast(s2);  // fp32 tree in-wave, fp64 beyond
                pend_par ^= 1;
                if (lane == 63) {
                    red[pend_par * 2 * NW + wave * 2 + 0] = d1;
                    red[pend_par * 2 * NW + wave * 2 + 1] = d2;
                }
                if constexpr (XF) {
                    const double x1 = (double)wave_sum_fast(sx1), x2 = (double)wave_sum_fast(sx2);
                    if (lane == 63) {
                        redx[pend_par * 2 * NW + wave * 2 + 0] = x1;
                        redx[pend_par * 2 * NW + wave * 2 + 1] = x2;
                    }
                }
                pend = true;
                pend_pos = Cp;
            }
        }
        stamp();
        // (5) stage +1 (loaded one stage ago) -> the other LDS buffer
        //     (after the very last stage this writes a never-consumed copy of the last item's chunk 0: cheaper than a
        //     `has_next` branch, whose skip path would leave Rn pending at the loop-head merge)
        if constexpr (!TAIL) {
            if constexpr (ILV) fs_end(Rn, cur ^ 1);
            else finish_stage(Rn, cur ^ 1);
        }
        stamp();
#ifndef DDIF_EMU
        // keep the tails of stage<LAST> and stage<!LAST> distinct: the optimiser otherwise sinks their common last LDS
        // writes below the join of the two, where the merged wait-count state degrades to vmcnt(0) (= waits for the prefetch)
        asm volatile("; end of stage kind %0" ::"n"(LAST ? 1 : 0));
#endif
    };

    stamp();
    // the first two stages' loads go out FIRST: their HBM latency overlaps the table fills below
    item_geometry();
    issue_loads(R0);
    issue_loads(R1);
    stamp();
    // Every prologue load is issued before any of them is used (fixed unroll counts, clamped indices): the tables, the
    // GroupNorm partials of the first sample and the two stage loads above share ONE memory latency.
    constexpr int TBL = 512 / NTHR > 0 ? 512 / NTHR : 1;  // 512 table entries per pass of the unrolled part
    constexpr int DWI = DWM ? (9 * 256 + NTHR - 1) / NTHR : 1;
    [[maybe_unused]] float t_g[TBL], t_b[TBL], t_dw[DWI];
    float t_bias[TBL], t_tb[TBL];
    [[maybe_unused]] GnPartials gp;
    if (GNP) {
#pragma unroll
        for (int k = 0; k < TBL; ++k) {
            const int i = tid + k * NTHR, c = i < Ctot ? i : Ctot - 1;
            t_g[k] = a.gamma[c];
            t_b[k] = a.beta[c];
        }
    }
#pragma unroll
    for (int k = 0; k < TBL; ++k) {
        const int i = tid + k * NTHR, c = i < a.Cout ? i : a.Cout - 1;
        t_bias[k] = a.bias[c];
    }
    if (DWM) {
#pragma unroll
        for (int k = 0; k < DWI; ++k) {
            const int i = tid + k * NTHR;
            t_dw[k] = a.dw_w[i < 9 * Ctot ? i : 9 * Ctot - 1];
        }
    }
    // WR: the cout tile's weights, once per workgroup (L2 hits for all but the first workgroups of an XCD); five named registers, not an array
    // (hipcc leaves a float4 array that is live across the branches below in scratch)
    [[maybe_unused]] float4 t_w0, t_w1, t_w2, t_w3, t_w4;
    static_assert(WRI <= 5, "resident weights: at most five float4 per thread");
    if constexpr (WR) {
        const float4* wsrc = reinterpret_cast<const float4*>(a.w);
        auto wld = [&](int k) { const int i = tid + k * NTHR; return wsrc[i < WBUF / 4 ? i : WBUF / 4 - 1]; };
        t_w0 = wld(0);
        if (WRI > 1) t_w1 = wld(1);
        if (WRI > 2) t_w2 = wld(2);
        if (WRI > 3) t_w3 = wld(3);
        if (WRI > 4) t_w4 = wld(4);
    }
    if (GNP) gn_load_partials(a.st0, a.np0, a.st1, a.np1, R0.pos.b, &gp);
    // the step's time-bias row LAST: its address waits for the step counter (a dependent scalar load); issued earlier, every load behind it in program order waits too
    tbrow = a.tbias + (size_t)dd_late(stepv) * a.tb_rowstride;
#pragma unroll
    for (int k = 0; k < TBL; ++k) {
        const int i = tid + k * NTHR, c = i < a.Cout ? i : a.Cout - 1;
        t_tb[k] = TBS ? 0.f : tbrow[c];
    }
    // ... and now the uses
    if constexpr (WR) {
        auto wst = [&](int k, const float4& v) { const int i = tid + k * NTHR; *reinterpret_cast<float4*>(i < WBUF / 4 ? &Ws[i * 4] : As + DUMMY) = v; };  // past the end: the A buffer's pad slot
        wst(0, t_w0);
        if (WRI > 1) wst(1, t_w1);
        if (WRI > 2) wst(2, t_w2);
        if (WRI > 3) wst(3, t_w3);
        if (WRI > 4) wst(4, t_w4);
    }
    if (GNP) {
#pragma unroll
        for (int k = 0; k < TBL; ++k) {
            const int i = tid + k * NTHR;
            if (i < GBN) {
                GBs[i] = t_g[k];
                GBs[GBN + i] = t_b[k];
            }
        }
        for (int i = tid + TBL * NTHR; i < GBN; i += NTHR) {  // more than 512 input channels
            const int c = i < Ctot ? i : Ctot - 1;
            GBs[i] = a.gamma[c];
            GBs[GBN + i] = a.beta[c];
        }
    }
    if (DWM) {
#pragma unroll
        for (int k = 0; k < DWI; ++k) {
            const int i = tid + k * NTHR;
            if (i < 9 * Ctot) DWs[i] = t_dw[k];
        }
    }
    if (GNP) {
        gn_reduce_partials(gp, a.st0, a.np0, a.st1, a.np1, R0.pos.b, (double)Ctot * a.Hin * a.Win, &mean, &rstd);
        gn_b = R0.pos.b;
    }
    __syncthreads();
    stamp();
    finish_stage(R0, 0);
    {   // bias + the step's time-bias row: needed by the first EPILOGUE only, so the fill comes behind the first stage's staging -- the row's address hangs on the step counter
        // (two dependent round trips on cold caches), and in front of the staging the first MFMA would wait for them
        const int nbt = a.n_ct * (32 * NB * WN);
#pragma unroll
        for (int k = 0; k < TBL; ++k) {
            const int i = tid + k * NTHR;
            if (i < nbt) BTs[i] = t_bias[k] + t_tb[k];
        }
        for (int i = tid + TBL * NTHR; i < nbt; i += NTHR) {
            const int c = i < a.Cout ? i : a.Cout - 1;
            BTs[i] = a.bias[c] + (TBS ? 0.f : tbrow[c]);
        }
        if constexpr (XF) {
            if (tid < 32 * NBX) BXs[tid] = a.xf_b[tid];
        }
    }
    __syncthreads();
    stamp();
#ifndef DDIF_EMU
    if (ABL & 32) {  // experiment: stagger the workgroups of a launch in time so that their memory phases do not coincide
        const int k = (ABL & 64) ? (blockIdx.x & 15) : (blockIdx.x & 7);
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep((ABL & 64) ? 4 : 8);
    }
#endif
    const int npairs = nflat >> 1;
    for (int pr = 0; pr < npairs; ++pr) {
        if (WR || bufch[0] == a.n_chunks - 1) stage(StageKind<1>{}, 0, R1, R0);   // WR: every stage is a whole item
        else stage(StageKind<0>{}, 0, R1, R0);
        __syncthreads();
        if (WR || bufch[1] == a.n_chunks - 1) stage(StageKind<1>{}, 1, R0, R1);
        else stage(StageKind<0>{}, 1, R0, R1);
        __syncthreads();
    }
    if (nflat & 1) stage(StageKind<3>{}, 0, R1, R0);  // odd tail: always the last chunk of the last item
    __syncthreads();
    flush_stats();
    stamp();
    if constexpr (RANGE) {
        if (a.range_flag && !(r_max < 65520.f)) *a.range_flag = 1;  // a scaled half of this launch overflowed (inf included; a staged NaN does NOT raise it -- fmaxf drops NaN operands -- and propagates to the output as it would in the reference)
    }
}


template <int KS, int STRIDE, int UPS, int TH, int TW, int CK, int NBT, int PRO = 0, int NW = 4, int MATH = 0>
constexpr size_t conv_smem_bytes() {  // NBT = n-blocks (of 32 couts) per workgroup = NB * WN; + conv_smem_extra() at launch
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr size_t dw = PRO == PRO_GN_DW ? (size_t)((TH + 2) * (TW + 2) * (CK + 4) + 9 * 256) : 0;
    constexpr int npl = (MATH == 3 || MATH == 5) ? 2 : (MATH == 4 ? 1 : 3);
    constexpr int lda = MATH >= 1 ? npl * CK / 2 + 4 : CK + 4, wchunk = MATH >= 1 ? KS * KS * (CK / 16) * npl * 256 : KS * KS * (CK / 8) * 256;
    constexpr int rp = (MATH >= 1 && KS == 3 && STRIDE == 1) ? (MATH == 5 ? (64 - (IW * lda) % 64) % 64 : (npl == 2 ? 24 : (npl == 1 ? 40 : 8))) : 0;
    return (size_t)(2 * IH * (IW * lda + rp) + ((MATH == 2 || MATH == 5) ? 1 : 2) * NBT * wchunk + dw) * sizeof(float) + 4 * NW * sizeof(double);
}
// GroupNorm prologues keep gamma | beta of all input channels in LDS; every kernel keeps bias (+ time bias) of all couts
inline size_t conv_smem_extra(int pro, int n_chunks, int ck, int cout_pad, int xf_cout = 0) {
    // (xf_cout: the folded x_conv's bias + the second output's statistics partials, [2][2 * 8 waves] doubles)
    return ((pro == PRO_GN || pro == PRO_GN_SILU || pro == PRO_GN_DW) ? (size_t)2 * n_chunks * ck : 0) * sizeof(float) + (size_t)cout_pad * sizeof(float) +
           (xf_cout ? (size_t)xf_cout * sizeof(float) + 32 * sizeof(double) : 0);
}

}  // namespace ddif

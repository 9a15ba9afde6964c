// Device-side refresh of the packed weight blob (training: the parameters change every iteration and live on the device;
// ddif_net_commit's host repack + upload of 10 M parameters per step would dominate the iteration).  One table-driven launch
// rewrites every packed tensor IN PLACE from the parameter tensors (reference layouts: OIHW conv weights, plain vectors),
// bit-identical to what Net::commit() (ddif_net.cpp pack_conv / pack_conv_x3) produces on the host.
#pragma once
#include "ddif_dev.h"

namespace ddif {

enum { RF_COPY = 0, RF_DW = 1, RF_SUM = 2, RF_PACK_F32 = 3, RF_PACK_X3 = 4 };

struct RefreshRec {
    int kind;
    const float* src0;   // OIHW weights / vector
    const float* src1;   // second source: concatenated along cin (attn_out | attn_res) or summed (bias pair); may be null
    float* dst;
    int cout, cin0, cin1, ks, ck, n_chunks;
    int tr, fc0, fc1;    // tr = 1: the pack of the DGRAD conv of a forward conv W (Cout_f = cin0, Cin_f = fc0 [+ fc1 from src1]):
                         //   W'[co'][ci'][tap] = W[ci'][co'][taps - 1 - tap]  (transposed, taps flipped); co' >= fc0 + fc1: zero padding
    long long n_out;     // floats written
    long long blk0;      // first thread block of this record (256 floats per block)
};

__global__ __launch_bounds__(256) void refresh_blob_kernel(const RefreshRec* recs, int n_recs) {
    // binary search: last record with blk0 <= blockIdx.x
    int lo = 0, hi = n_recs - 1;
    const long long blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (recs[mid].blk0 <= blk) lo = mid;
        else hi = mid - 1;
    }
    const RefreshRec r = recs[lo];
    const long long idx64 = (blk - r.blk0) * 256 + threadIdx.x;
    if (idx64 >= r.n_out) return;
    const unsigned idx = (unsigned)idx64;  // a record writes < 2^31 floats: 32-bit index arithmetic below (a 64-bit division is ~100 instructions, six per element)
    if (r.kind == RF_COPY) {
        r.dst[idx] = r.src0[idx];
    } else if (r.kind == RF_SUM) {
        r.dst[idx] = r.src0[idx] + r.src1[idx];
    } else if (r.kind == RF_DW) {  // (C,1,3,3) -> [9][C]
        const int C = r.cout;
        const unsigned k = idx / (unsigned)C, c = idx - k * (unsigned)C;
        r.dst[idx] = r.src0[(size_t)c * 9 + k];
    } else {
        const int taps = r.ks * r.ks, cin = r.cin0 + r.cin1;
        auto wval = [&](int co, int ci, int tap) -> float {
            if (r.tr) {
                if (ci >= r.cin0) return 0.f;
                if (co < r.fc0) return r.src0[((size_t)ci * r.fc0 + co) * taps + (taps - 1 - tap)];
                if (co < r.fc0 + r.fc1) return r.src1[((size_t)ci * r.fc1 + (co - r.fc0)) * taps + (taps - 1 - tap)];
                return 0.f;
            }
            if (co >= r.cout || ci >= cin) return 0.f;
            if (ci < r.cin0) return r.src0[((size_t)co * r.cin0 + ci) * taps + tap];
            return r.src1[((size_t)co * r.cin1 + (ci - r.cin0)) * taps + tap];
        };
        if (r.kind == RF_PACK_F32) {
            // [n-block][chunk][tap][k8][half h][cout j][4 cins]   (ddif_net.cpp pack_conv)
            const int K8 = r.ck / 8;
            unsigned t = idx;
            const int i = (int)(t & 3);
            t >>= 2;
            const int j = (int)(t & 31);
            t >>= 5;
            const int h = (int)(t & 1);
            t >>= 1;
            const int k8 = (int)(t % (unsigned)K8);
            t /= (unsigned)K8;
            const int tap = (int)(t % (unsigned)taps);
            t /= (unsigned)taps;
            const int ch = (int)(t % (unsigned)r.n_chunks);
            const int nbi = (int)(t / (unsigned)r.n_chunks);
            r.dst[idx] = wval(nbi * 32 + j, ch * r.ck + k8 * 8 + 4 * h + i, tap);
        } else {
            // [n-block][chunk][tap][k16][plane][half h][cout j][8 bf16]: one float = two bf16 (t = 2q, 2q + 1)   (pack_conv_x3)
            const int K16 = r.ck / 16;
            unsigned t = idx;
            const int q = (int)(t & 3);
            t >>= 2;
            const int j = (int)(t & 31);
            t >>= 5;
            const int h = (int)(t & 1);
            t >>= 1;
            const int pl = (int)(t % 3u);
            t /= 3u;
            const int k16 = (int)(t % (unsigned)K16);
            t /= (unsigned)K16;
            const int tap = (int)(t % (unsigned)taps);
            t /= (unsigned)taps;
            const int ch = (int)(t % (unsigned)r.n_chunks);
            const int nbi = (int)(t / (unsigned)r.n_chunks);
            unsigned out = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float v = wval(nbi * 32 + j, ch * r.ck + 16 * k16 + 8 * h + 2 * q + e, tap);
                unsigned p3[3];
                dd_split3(v, &p3[0], &p3[1], &p3[2]);
                out |= (p3[pl] & 0xffffu) << (16 * e);
            }
            reinterpret_cast<unsigned*>(r.dst)[idx] = out;
        }
    }
}

}  // namespace ddif

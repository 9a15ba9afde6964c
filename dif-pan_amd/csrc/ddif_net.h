// Host-side structures of libddif: the network (weights repacked for the kernels) and helpers.
#pragma once
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ddif.h"
#include "../../include/ddif_testops.h"
#include "ddif_dev.h"

namespace ddif {

extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...);

#define DDIF_HIPCHK(x)                                                                                      \
    do {                                                                                                    \
        hipError_t e__ = (x);                                                                               \
        if (e__ != hipSuccess)                                                                              \
            return ::ddif::fail(DDIF_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e__), __FILE__, \
                                __LINE__);                                                                  \
    } while (0)

// conv prologue applied while staging the input tile (kernels_conv.h)
enum { PRO_NONE = 0, PRO_GN = 1, PRO_GN_SILU = 2, PRO_COLSM = 3, PRO_GN_DW = 4 };

enum LayerKind { L_STEM, L_ENC, L_DOWN, L_MID, L_DEC, L_UP };

struct Layer {
    LayerKind kind;
    int cin = 0, cout = 0, cx = 0, cskip = 0;
    bool attn = false;
    std::string p;  // state-dict prefix, e.g. "downs.3"
};

// A convolution's weights in kernel order (see pack_conv in ddif_net.cpp) + its bias, both device pointers.
struct PackedConv {
    const float* w = nullptr;
    const float* w_x3 = nullptr;  // 3x3 convs: the same weights as three bf16 planes (kernels_conv.h MATH = 1)
    const float* w_f16 = nullptr; // the same weights x 2^10 as two half planes (kernels_conv.h MATH = 3); inference plans only, null when a
                                  // weight exceeds the scaled half range
    const float* w_b1 = nullptr;  // the hi bf16 plane alone (kernels_conv.h MATH = 4: the throughput variant); inference plans only
    const float* bias = nullptr;
    int cin = 0, cout = 0, ks = 1, ck = 32, n_chunks = 0;
};

struct HostTensor {
    std::vector<float> v;
    std::vector<int64_t> shape;
};

struct Net {
    ddif_net_cfg cfg{};
    int device = 0;
    std::vector<Layer> downs, mid, ups;
    int final_in = 0;
    std::map<std::string, HostTensor> host;
    bool committed = false;
    unsigned long long generation = 0;  // bumped by every commit(): plans built against an older blob are refused

    float* blob = nullptr;  // one device allocation holding every repacked tensor
    size_t blob_floats = 0;
    std::map<std::string, PackedConv> conv;       // key = conv weight key without ".weight"
    std::map<std::string, const float*> vec;      // GroupNorm gammas/betas and depthwise weights by full key
    std::map<const float*, float> vec_absmax;     // max |v| of each of them (the f16x2 path bounds GroupNorm outputs with it)
    // time embedding
    const float *freqs = nullptr, *w1 = nullptr, *b1 = nullptr, *w3 = nullptr, *b3 = nullptr, *wall = nullptr,
                *ball = nullptr;
    int nslots = 0;                               // total FeatureWiseAffine output channels
    std::map<std::string, int> slot_off;          // res_block prefix -> offset into a time-bias row

    // device-side refresh (training): how every packed tensor of the blob derives from the parameter tensors, recorded by commit()
    struct Recipe {
        int kind = 0;                 // RF_* (kernels_refresh.h)
        std::string src0, src1;       // state-dict keys
        size_t dst_off = 0;           // floats into the blob
        int cout = 0, cin0 = 0, cin1 = 0, ks = 1, ck = 32, n_chunks = 0;
        int tr = 0, fc0 = 0, fc1 = 0; // dgrad packs (kernels_refresh.h)
        bool dblob = false;           // destination lives in the dgrad blob
        size_t n_out = 0;
    };
    std::vector<Recipe> recipes;
    // training only: the weights of every conv's DGRAD conv (transposed, taps flipped), packed by the same refresh launch.  Built on
    // demand by the first train-mode plan (build_dgrad_packs); key = forward conv key (".attn_mix": dgrad over cat[o, xn])
    std::map<std::string, PackedConv> dconv;
    float* dgrad_blob = nullptr;
    size_t dgrad_floats = 0;
    bool dgrad_filled = false;        // a refresh_device() has run since the dgrad packs were allocated (they start zeroed: a backward pass before that
                                      // would return zero input gradients upstream of the last layer without any error)
    // plans built on this net hold raw pointers into it: ddif_net_destroy defers to the last plan's destruction.  Both fields under `life_mu` (plans and
    // nets are destroyed from Python finalizers, possibly on different threads); `owner` = the C-ABI handle that contains this Net.
    std::mutex life_mu;
    int live_plans = 0;
    bool orphaned = false;
    void* owner = nullptr;
    int build_dgrad_packs();
    bool merged_stale = false;        // the eval-only merged ffn[3] o ffn[2] weights were NOT refreshed (train-mode plans do not use them)
    void* d_recs = nullptr;           // device RefreshRec table of the last refresh
    std::vector<const float*> last_ptrs;
    std::vector<std::string> refresh_keys;         // arguments of the last successful refresh (the map is rebuilt only when they change)
    std::vector<const float*> refresh_in_ptrs;
    int n_recs = 0;
    long long refresh_blocks = 0;
    int refresh_device(int n, const char* const* keys, const float* const* ptrs, hipStream_t stream);
    std::vector<hipEvent_t> reader_events;  // train-mode plans: 'my side stream has read the weights' -- a refresh waits for them before it rewrites the packs

    ~Net() {
        if (blob) (void)hipFree(blob);
        if (d_recs) (void)hipFree(d_recs);
        if (dgrad_blob) (void)hipFree(dgrad_blob);
    }
    int build_layers();
    int load(const char* key, const float* data, const int64_t* shape, int ndim);
    int commit(hipStream_t stream);
    int64_t num_params() const;
    const HostTensor* get(const std::string& key) const {
        auto it = host.find(key);
        return it == host.end() ? nullptr : &it->second;
    }
};

}  // namespace ddif

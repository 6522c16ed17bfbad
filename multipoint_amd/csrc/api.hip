// C ABI of libmultipoint_hip.so (declared in include/multipoint_hip.h): handle, weight repacking,
// workspace, and the launch sequences of the hot path.
#include "../../include/multipoint_hip.h"
#include "mp_common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct ConvLayer {            // one MFMA conv launch
    const char* name = "";
    float *wpack = nullptr, *bias = nullptr, *scale = nullptr, *shift = nullptr;
    float* u43pack = nullptr;     // 3x3 layers: F(4x4,3x3) weights (conv_wino43.hip)
    // enc.conv2 inside the fused conv1+conv2 launch, conv -> ReLU -> BN models: the first block's BatchNorm folded into this layer --
    // U from g2[o][c][tap] * s1[c] (in double, rounded once) and bias + sum_c t1[c] sum_tap g2[o][c][tap] (exact: with reflection
    // padding every tap of every output lands on a real pixel, so the shift's contribution is one constant per output channel)
    float *u43pack_f1 = nullptr, *bias_f1 = nullptr;
    _Float16* wpack_h = nullptr;  // mixed_precision: fp16 fragments (conv_f16.hip) and the fp16-rounded bias
    float* bias_h = nullptr;
    int cin = 0, cout = 0, taps = 9, nslices = 0;
    bool pool = false, relu = true;
};

struct FirstLayer {
    int channels = 64;            // output channels incl. zero padding (64 or 32)
    float *w = nullptr, *bias = nullptr, *scale = nullptr, *shift = nullptr;
    float *w_h = nullptr, *bias_h = nullptr;      // mixed_precision: fp16-representable copies
    // the fused F(4x4,3x3) conv1+conv2 launch produces relu(conv1) only: bn_first models get their BatchNorm folded into the block's own
    // weights (w s, b s + t), the others into conv2 (ConvLayer::u43pack_f1); without BatchNorm these are the plain weights
    float *w_f1 = nullptr, *bias_f1 = nullptr;
};

struct Encoder {
    FirstLayer first;
    ConvLayer conv[7];
    int nconv = 7;                  // 3x3 layers after the first one: 7, or 3 with double_convolution: false (MultiPoint.py:147-148)
    bool first_pool = false;        // ... where MaxPool2d(2,2) follows the first block directly
};

struct ProfEntry {
    const char* name;
    hipEvent_t a, b;
    double flop;
};

}  // namespace

struct mp_handle {
    int device = 0;
    int ncu = 256, xcd_shift = 3;   // machine shape derived in mp_create: compute units, log2(XCDs) (MP_DEBUG=ncu / MP_DEBUG=nxcd override)
    std::string err;
    bool loaded = false;
    mp_model_config cfg{};
    std::vector<void*> weight_allocs;
    Encoder enc[2];                 // [0] = encoder / encoder_thermal, [1] = encoder_optical
    ConvLayer heads3, det1, desc1;
    DevBuf ws;                      // forward workspace
    DevBuf ws2;                     // NMS work map + kept lists
    DevBuf ws3;                     // matching arg-min arrays
    DevBuf ws4;                     // pair metrics: warped keypoints + inverse match map
    DevBuf split_ws;                // F(4x4,3x3) split launches: the ranges' pre-bias output tiles
    DevBuf vin_ws;                  // F(4x4,3x3) layers with >= 4 output slices: the pre-transformed input (ConvParams::vglobal)
    int vin_min_slices = 4;         // ... from this many slices on (the 3x3 head convolutions: 8): GEMM pass 0.865 ms at 75 % of the matrix pipe +
                                    // 0.106 ms for the producer against 1.03 ms with the in-kernel transform per slice; MP_DEBUG=no_vin: never
    int splitk_max = 8;             // most ranges the input channels of a small launch are cut into (MP_DEBUG=splitk_max; 1: never)
    int fwd_batch = 0;              // images of the forward in flight: the split launches are gated on THIS, not on an encoder's share of it
    int splitk_env = 8;             // ... as mp_create set it (model.batch_invariant overrides it per loaded model)
    DevBuf nms_state;               // 64 round counters + tile flags
    DevBuf kp_scratch;              // segment counts + list totals of the keypoint compaction
    int* nms_total = nullptr;       // device: undecided candidates summed over all calls since the last read
    int last_nms_rounds = 0;
    int* tie_state = nullptr;       // device, 1 + MP_TIE_MAX_IMAGES ints: top-k tie guard (mp_topk_ambiguous)
    float tie_eps = 6e-5f;          // ... a survivor within this of the k-th score counts as 'at the cut' (mp_topk_tie_guard)
    int tie_min = 4;                // ... an image is flagged when at least this many sit at the cut on EACH side of it; 0: guard off
    int tie_last_B = 0;
    int* tie_pairs = nullptr;       // device int [tie_pairs_cap]: footprint tie guard, per-image counts of the latest call's NMS (nms.hip)
    int tie_pairs_cap = 0;
    int tie_pairs_min = 16;         // ... an image is flagged when at least this many of its NMS decisions fell between scores within tie_eps; 0: off
    int head_channels = 256;        // width of each 3x3 head convolution (MultiPoint.py:38-53)
    void* dummy = nullptr;          // scratch line for masked-off store lanes of the fp16 kernels
    bool wino = true;               // Winograd F(4x4,3x3) for the 3x3 layers (MP_DEBUG=no_winograd / conv_algorithm 'direct': the direct kernels)
    int persist = 8;                // persistent conv workgroups for launches with >= this many items per CU
                                    // (MP_DEBUG=no_persist: never; MP_DEBUG=persist_min_items=n overrides the threshold)
    bool fuse_first = true;         // fuse the Cin=1 block into the second convolution (MP_DEBUG=no_fuse disables)
    int planar = 1;                 // 0 (MP_DEBUG=no_planar): NHWC everywhere; 1: channel-quad-planar tensors where they pay (behind conv1 and pooled F(4x4,3x3) producers)
    int wino43 = 2;                 // MP_DEBUG=wino43: 0 off (direct kernels), 1 F(4x4,3x3) for the 3x3 layers with 64 input channels only, 2 (default) every 3x3 layer
    bool head_fuse = true;          // MP_DEBUG=no_head_fuse: separate 1x1 convolution / softmax / normalisation launches
    bool fuse43 = true;             // first block evaluated inside the F(4x4,3x3) conv2 kernel (MP_DEBUG=no_fuse43: its own launch)
    int wino43_gen = 0;             // MP_DEBUG=wino43_gen: 0 conv_wino43.hip where it applies, conv_wino43b.hip for every other shape; 1 / 2: only that kernel
    bool wino_env = true;           // wino / wino43 / wino43_gen as mp_create read them: mp_load_weights starts from these and
    int wino43_env = 2, wino43_gen_env = 0;   // applies the model's conv_algorithm on top (a reload never inherits the previous model's)
    int* pinned = nullptr;          // small pinned host scratch (img lists, counters)
    bool f16_res = true;            // MP_DEBUG=f16_no_res: the streaming kernel (conv_f16.hip) also for the 64 -> 64 layers
    bool f16_fuse1 = true;          // MP_DEBUG=f16_no_fuse1: the first block of the fp16 path as its own launch
    int f16_res_groups = 3;         // MP_DEBUG=f16_res_groups=2: two instead of three wave groups per CU in conv_f16_res.hip
    bool prof = false;
    bool head_fallback_noted = false;
    std::vector<ProfEntry> prof_entries;
    size_t prof_used = 0;
};

namespace {

int fail(mp_handle* h, int code, const std::string& msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define MP_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(h, MP_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
    } while (0)

int ensure(mp_handle* h, DevBuf& b, size_t bytes)
{
    if (b.bytes >= bytes) return MP_OK;
    if (b.p) { MP_HIP(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(h, MP_ENOMEM, "hipMalloc(" + std::to_string(bytes) + " B): " + hipGetErrorString(e));
    }
    b.bytes = bytes;
    return MP_OK;
}

int upload(mp_handle* h, const std::vector<float>& v, float** out)
{
    void* d = nullptr;
    MP_HIP(hipMalloc(&d, v.size() * sizeof(float)));
    h->weight_allocs.push_back(d);
    MP_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = static_cast<float*>(d);
    return MP_OK;
}

void free_weights(mp_handle* h)
{
    for (void* p : h->weight_allocs) (void)hipFree(p);
    h->weight_allocs.clear();
    h->loaded = false;
}

// IEEE binary16 <-> binary32 on the host, round-to-nearest-even (what tensor.half() does)
uint16_t f2h_bits(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          // >= 65520 rounds to inf
    if (x < 0x38800000u) {                                             // below 2^-14: fp16 subnormal
        if (x < 0x33000000u) return (uint16_t)sign;                    // below 2^-25: zero
        const int e = (int)(x >> 23);
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = (((x >> 23) - 112u) << 10) | ((x & 0x7fffffu) >> 13);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}

float h2f_bits(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) { x = sign; }
        else {
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 0x400u)) { mm <<= 1; ++sh; }
            x = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ffu) << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

float round_half(float f) { return h2f_bits(f2h_bits(f)); }

struct TensorMap {
    std::map<std::string, const mp_tensor*> m;
    std::map<std::string, bool> used;
    const float* get(const std::string& k, long long numel, std::string& err)
    {
        auto it = m.find(k);
        if (it == m.end()) { err = "missing key in state_dict: " + k; return nullptr; }
        if (it->second->numel != numel) {
            err = "size mismatch for " + k + ": got " + std::to_string(it->second->numel) +
                  " elements, expected " + std::to_string(numel);
            return nullptr;
        }
        used[k] = true;
        return it->second->data;
    }
};

// eval-mode BatchNorm2d(eps=1e-5) as y = x*scale + shift, evaluated like ATen's CPU kernel
// (batch_norm_cpu_collect_linear_and_constant_terms): invstd = 1/sqrt(var+eps); alpha = invstd*gamma;
// beta' = beta - mean*alpha, all in fp32.
bool bn_terms(TensorMap& tm, const std::string& prefix, int c, int padded, std::vector<float>& scale,
              std::vector<float>& shift, std::string& err)
{
    const float* g = tm.get(prefix + ".weight", c, err); if (!g) return false;
    const float* b = tm.get(prefix + ".bias", c, err); if (!b) return false;
    const float* m = tm.get(prefix + ".running_mean", c, err); if (!m) return false;
    const float* v = tm.get(prefix + ".running_var", c, err); if (!v) return false;
    if (tm.m.count(prefix + ".num_batches_tracked")) tm.used[prefix + ".num_batches_tracked"] = true;
    scale.assign(padded, 1.f); shift.assign(padded, 0.f);
    for (int i = 0; i < c; ++i) {
        const float invstd = 1.0f / std::sqrt(v[i] + 1e-5f);
        const float alpha = invstd * g[i];
        scale[i] = alpha;
        shift[i] = b[i] - m[i] * alpha;
    }
    return true;
}

// Packed B-operand layout consumed by conv_mfma_kernel:
//   [slice][chunk][step = tap*4 + kgroup][nblock(2)][lane(64)][4]
//   element e of lane l = W[cout = slice*64 + nblock*32 + (l&31)][cin = chunk*32 + kgroup*8 + (l>>5)*4 + e][tap]
// srcs: list of OIHW tensors concatenated along O (the two 3x3 head convs share one launch).
// cin_real < cin: the input tensor carries zero padding channels up to a multiple of 32 (channel_version 1 / 2).
void pack_conv_weights(const std::vector<const float*>& srcs, const std::vector<int>& couts, int cin, int cin_real,
                       int taps, std::vector<float>& out)
{
    int cout = 0;
    for (int c : couts) cout += c;
    const int nslices = (cout + 63) / 64, nchunks = cin / 32;
    // + 2 steps of zero padding: the kernel's weight prefetch runs two steps past the last slice
    out.assign((size_t)nslices * nchunks * taps * 4 * 2 * 64 * 4 + 2 * 2 * 64 * 4, 0.f);
    size_t o = 0;
    for (int s = 0; s < nslices; ++s)
        for (int c = 0; c < nchunks; ++c)
            for (int tap = 0; tap < taps; ++tap)
                for (int g = 0; g < 4; ++g)
                    for (int nb = 0; nb < 2; ++nb)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 4; ++e, ++o) {
                                int co = s * 64 + nb * 32 + (l & 31);
                                const int ci = c * 32 + g * 8 + (l >> 5) * 4 + e;
                                if (co >= cout || ci >= cin_real) continue;
                                size_t t = 0;
                                while (co >= couts[t]) { co -= couts[t]; ++t; }
                                out[o] = srcs[t][((size_t)co * cin_real + ci) * taps + tap];
                            }
}

// Winograd F(4x4,3x3) weights for conv_wino43_kernel: U[pos = 6i+j] = (G g G^T)[i][j] for the interpolation points
// {0, +a, -a, +b, -b, inf} (a = MP_W43_A, b = MP_W43_B, mp_common.h): row of point p = [1, p, p^2] / prod_{q != p} (p - q), last
// row [0, 0, 1]; evaluated in double and rounded to fp32 ONCE.  Layout = the LDS image of a unit of 4 input channels:
//   [slice64][unit = cin/4][ch(4)][cout(64)][pos(36)]
void pack_wino43_weights(const std::vector<const float*>& srcs, const std::vector<int>& couts, int cin, int cin_real,
                         std::vector<float>& out, const float* in_scale = nullptr)
{
    const double pts[5] = {0.0, MP_W43_A, -MP_W43_A, MP_W43_B, -MP_W43_B};
    double G[6][3];
    for (int k = 0; k < 5; ++k) {
        double n = 1.0;
        for (int q = 0; q < 5; ++q)
            if (q != k) n *= pts[k] - pts[q];
        G[k][0] = 1.0 / n; G[k][1] = pts[k] / n; G[k][2] = pts[k] * pts[k] / n;
    }
    G[5][0] = 0.0; G[5][1] = 0.0; G[5][2] = 1.0;
    int cout = 0;
    for (int c : couts) cout += c;
    const int nslices = (cout + 63) / 64, nunits = cin / 4;
    out.assign((size_t)nslices * nunits * 4 * 64 * 36, 0.f);
    for (int s = 0; s < nslices; ++s)
        for (int u = 0; u < nunits; ++u)
            for (int ch = 0; ch < 4; ++ch)
                for (int co64 = 0; co64 < 64; ++co64) {
                    int co = s * 64 + co64;
                    const int ci = u * 4 + ch;
                    if (co >= cout || ci >= cin_real) continue;
                    size_t t = 0;
                    while (co >= couts[t]) { co -= couts[t]; ++t; }
                    const float* g = srcs[t] + ((size_t)co * cin_real + ci) * 9;
                    const double sc = in_scale ? (double)in_scale[ci] : 1.0;      // (a producer's BatchNorm scale folded into this layer)
                    double tmp[6][3];
                    for (int a = 0; a < 6; ++a)
                        for (int j = 0; j < 3; ++j) tmp[a][j] = sc * (G[a][0] * g[j] + G[a][1] * g[3 + j] + G[a][2] * g[6 + j]);
                    float* o = out.data() + ((((size_t)s * nunits + u) * 4 + ch) * 64 + co64) * 36;
                    for (int a = 0; a < 6; ++a)
                        for (int b = 0; b < 6; ++b) o[6 * a + b] = (float)(tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2]);
                }
}

// fp16 flavour for conv_f16_kernel: chunks of 64 input channels, steps of 16:
//   [slice][chunk][step = tap*4 + kgroup][nblock(2)][lane(64)][8]
//   element e of lane l = half(W[cout = slice*64 + nblock*32 + (l&31)][cin = chunk*64 + kgroup*16 + (l>>5)*8 + e][tap])
void pack_conv_weights_h(const std::vector<const float*>& srcs, const std::vector<int>& couts, int cin, int cin_real,
                         int taps, std::vector<uint16_t>& out)
{      // cin_real < cin: the input tensor carries zero padding channels (channel_version 1 / 2): their weights stay zero
    int cout = 0;
    for (int c : couts) cout += c;
    const int nslices = (cout + 63) / 64, nchunks = cin / 64;
    // + 5 steps of zero padding: the weight prefetch runs 5 steps past the last slice
    out.assign((size_t)nslices * nchunks * taps * 4 * 2 * 64 * 8 + 5 * 2 * 64 * 8, 0);
    size_t o = 0;
    for (int s = 0; s < nslices; ++s)
        for (int c = 0; c < nchunks; ++c)
            for (int tap = 0; tap < taps; ++tap)
                for (int g = 0; g < 4; ++g)
                    for (int nb = 0; nb < 2; ++nb)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e, ++o) {
                                int co = s * 64 + nb * 32 + (l & 31);
                                const int ci = c * 64 + g * 16 + (l >> 5) * 8 + e;
                                if (co >= cout || ci >= cin_real) continue;
                                size_t t = 0;
                                while (co >= couts[t]) { co -= couts[t]; ++t; }
                                out[o] = f2h_bits(srcs[t][((size_t)co * cin_real + ci) * taps + tap]);
                            }
}

// cin: channel count of the (zero-padded) input tensor, a multiple of 32; cin_real: channels of the reference conv.
// L.cout is rounded up to a multiple of 32: the extra output channels have zero weights/bias and identity BN, so
// the kernel writes zeros there -- exactly the padding the next layer expects.
int build_conv(mp_handle* h, TensorMap& tm, ConvLayer& L, const char* name,
               const std::vector<std::string>& conv_keys, const std::vector<std::string>& bn_keys,
               const std::vector<int>& couts, int cin, int taps, bool pool, bool relu, int cin_real = 0,
               bool pad_cout = false)
{
    if (cin_real <= 0) cin_real = cin;
    std::string err;
    int cout = 0;
    for (int c : couts) cout += c;
    const int padded = ((cout + 63) / 64) * 64;
    std::vector<const float*> srcs;
    std::vector<float> bias(padded, 0.f), scale(padded, 1.f), shift(padded, 0.f);
    int off = 0;
    for (size_t i = 0; i < conv_keys.size(); ++i) {
        const float* w = tm.get(conv_keys[i] + ".weight", (long long)couts[i] * cin_real * taps, err);
        if (!w) return fail(h, MP_EINVAL, err);
        const float* b = tm.get(conv_keys[i] + ".bias", couts[i], err);
        if (!b) return fail(h, MP_EINVAL, err);
        srcs.push_back(w);
        for (int c = 0; c < couts[i]; ++c) bias[off + c] = b[c];
        if (!bn_keys[i].empty()) {
            std::vector<float> s, t;
            if (!bn_terms(tm, bn_keys[i], couts[i], couts[i], s, t, err)) return fail(h, MP_EINVAL, err);
            for (int c = 0; c < couts[i]; ++c) { scale[off + c] = s[c]; shift[off + c] = t[c]; }
        }
        off += couts[i];
    }
    std::vector<float> packed;
    pack_conv_weights(srcs, couts, cin, cin_real, taps, packed);
    const int pm = h->cfg.mixed_precision ? 64 : 32;           // channel padding granule: the fp16 kernels walk K in chunks of 64
    L.name = name; L.cin = cin; L.cout = pad_cout ? ((cout + pm - 1) / pm) * pm : cout; L.taps = taps; L.nslices = padded / 64;
    L.pool = pool; L.relu = relu;
    int rc;
    if ((rc = upload(h, packed, &L.wpack))) return rc;
    if ((rc = upload(h, bias, &L.bias))) return rc;
    if ((rc = upload(h, scale, &L.scale))) return rc;
    if ((rc = upload(h, shift, &L.shift))) return rc;
    if (taps == 9 && h->wino && h->wino43 && cin % 8 == 0) {
        std::vector<float> u4;
        pack_wino43_weights(srcs, couts, cin, cin_real, u4);
        if ((rc = upload(h, u4, &L.u43pack))) return rc;
    }
    if (h->cfg.mixed_precision) {
        std::vector<uint16_t> ph;
        pack_conv_weights_h(srcs, couts, cin, cin_real, taps, ph);
        void* d = nullptr;
        MP_HIP(hipMalloc(&d, ph.size() * 2));
        h->weight_allocs.push_back(d);
        MP_HIP(hipMemcpy(d, ph.data(), ph.size() * 2, hipMemcpyHostToDevice));
        L.wpack_h = static_cast<_Float16*>(d);
        std::vector<float> bh(bias);
        for (float& v : bh) v = round_half(v);
        if ((rc = upload(h, bh, &L.bias_h))) return rc;
    }
    return MP_OK;
}

const char* kEncNames[7] = {"enc.conv2", "enc.conv3", "enc.conv4", "enc.conv5", "enc.conv6", "enc.conv7",
                            "enc.conv8"};

int build_encoder(mp_handle* h, TensorMap& tm, Encoder& E, const std::string& prefix)
{
    // MultiPoint: generate_encoder (MultiPoint.py:168-185): Sequential indices, 4 modules per conv block
    // (pad, conv, X, Y) and one MaxPool2d after blocks 2, 4, 6.
    // SuperPointMagicLeap (SuperPointMagicLeap.py:16-23): named convolutions, no BatchNorm.
    // double_convolution: false -- one (pad, conv, X, Y) group per stage, a pool after stages 1-3: indices 1, 6, 11, 16
    static const int conv_idx2[8] = {1, 5, 10, 14, 19, 23, 28, 32};
    static const int conv_idx1[4] = {1, 6, 11, 16};
    const bool dbl = h->cfg.double_convolution != 0;
    const int* conv_idx = dbl ? conv_idx2 : conv_idx1;
    static const char* ml_names[8] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b"};
    // MultiPoint.py:38-53: channel_version 0 [1,64,64,128,128], 1 [1,32,64,96,128], 2 [1,8,16,32,64]
    static const int stage_ch[3][5] = {{1, 64, 64, 128, 128}, {1, 32, 64, 96, 128}, {1, 8, 16, 32, 64}};
    const int* sc = stage_ch[h->cfg.channel_version];
    const int chan2[9] = {1, sc[1], sc[1], sc[2], sc[2], sc[3], sc[3], sc[4], sc[4]};
    const int chan1[9] = {1, sc[1], sc[2], sc[3], sc[4], 0, 0, 0, 0};
    const int* chan = dbl ? chan2 : chan1;
    // tensors carry zero padding channels up to a multiple of 32 (fp32 kernels) or 64 (fp16 kernels: their K chunk)
    const int pgran = h->cfg.mixed_precision ? 64 : 32;
    auto pad32 = [pgran](int c) { return ((c + pgran - 1) / pgran) * pgran; };
    static const bool pool2[8] = {false, true, false, true, false, true, false, false};
    static const bool pool1[8] = {true, true, true, false, false, false, false, false};
    const bool* pool = dbl ? pool2 : pool1;
    E.nconv = dbl ? 7 : 3;
    E.first_pool = pool[0];
    const int bn_off = h->cfg.bn_first ? 1 : 2;
    auto conv_key = [&](int i) {
        return h->cfg.key_layout == 1 ? std::string(ml_names[i]) : prefix + "." + std::to_string(conv_idx[i]);
    };
    auto bn_key = [&](int i) {
        return h->cfg.batchnorm ? prefix + "." + std::to_string(conv_idx[i] + bn_off) : std::string();
    };
    std::string err;
    {   // first layer (Cin = 1): [tap][cout], cout zero-padded to 32 / 64
        const std::string ck = conv_key(0), bk = bn_key(0);
        const int c1 = chan[1], c1p = pad32(c1);
        const float* w = tm.get(ck + ".weight", c1 * 9, err); if (!w) return fail(h, MP_EINVAL, err);
        const float* b = tm.get(ck + ".bias", c1, err); if (!b) return fail(h, MP_EINVAL, err);
        std::vector<float> wt(9 * c1p, 0.f), bias(c1p, 0.f), s(c1p, 1.f), t(c1p, 0.f);
        for (int co = 0; co < c1; ++co) {
            bias[co] = b[co];
            for (int k = 0; k < 9; ++k) wt[k * c1p + co] = w[co * 9 + k];
        }
        if (!bk.empty() && !bn_terms(tm, bk, c1, c1p, s, t, err)) return fail(h, MP_EINVAL, err);
        E.first.channels = c1p;
        int rc;
        if ((rc = upload(h, wt, &E.first.w))) return rc;
        if ((rc = upload(h, bias, &E.first.bias))) return rc;
        if ((rc = upload(h, s, &E.first.scale))) return rc;
        if ((rc = upload(h, t, &E.first.shift))) return rc;
        if (h->cfg.mixed_precision) {
            for (float& v : wt) v = round_half(v);
            for (float& v : bias) v = round_half(v);
            if ((rc = upload(h, wt, &E.first.w_h))) return rc;
            if ((rc = upload(h, bias, &E.first.bias_h))) return rc;
        }
    }
    for (int i = 1; i <= E.nconv; ++i) {
        int rc = build_conv(h, tm, E.conv[i - 1], kEncNames[i - 1], {conv_key(i)}, {bn_key(i)}, {chan[i + 1]}, pad32(chan[i]), 9,
                            pool[i], true, chan[i], true);
        if (rc) return rc;
    }
    // The fused F(4x4,3x3) conv1+conv2 launch (conv_wino43.hip F1: channel_version 0, double convolution, reflection padding) produces
    // relu(conv1) and nothing else per patch pixel: the first block's BatchNorm is folded at load time -- into the block's own weights
    // for bn_first models (conv -> BN -> ReLU), into conv2's Winograd-domain weights and bias otherwise (conv -> ReLU -> BN -> pad ->
    // conv2).  Exact in real arithmetic; in fp32 one rounding per activation fewer than the un-fused launches (equal within the tolerance
    // class of any two kernel variants: tests/test_gpu_parity.py::test_first_block_inside_f43_equals_standalone).
    if (dbl && h->cfg.channel_version == 0 && h->cfg.reflection_pad && E.conv[0].u43pack && E.conv[0].cin == 64 && chan[1] == 64) {
        const std::string ck = conv_key(0), bk = bn_key(0), ck2 = conv_key(1);
        const float* w1 = tm.get(ck + ".weight", 64 * 9, err); if (!w1) return fail(h, MP_EINVAL, err);
        const float* b1 = tm.get(ck + ".bias", 64, err); if (!b1) return fail(h, MP_EINVAL, err);
        const float* w2 = tm.get(ck2 + ".weight", 64LL * 64 * 9, err); if (!w2) return fail(h, MP_EINVAL, err);
        const float* b2 = tm.get(ck2 + ".bias", 64, err); if (!b2) return fail(h, MP_EINVAL, err);
        std::vector<float> s1(64, 1.f), t1(64, 0.f);
        if (!bk.empty() && !bn_terms(tm, bk, 64, 64, s1, t1, err)) return fail(h, MP_EINVAL, err);
        std::vector<float> wf(9 * 64), bf(64), b2f(64);
        const bool own = h->cfg.bn_first != 0;                      // fold into the block itself
        for (int co = 0; co < 64; ++co) {
            bf[co] = own ? (float)((double)b1[co] * s1[co] + t1[co]) : b1[co];
            for (int k = 0; k < 9; ++k) wf[k * 64 + co] = own ? (float)((double)w1[co * 9 + k] * s1[co]) : w1[co * 9 + k];
        }
        for (int o = 0; o < 64; ++o) {
            double acc = b2[o];
            if (!own)
                for (int c = 0; c < 64; ++c) {
                    double g = 0.0;
                    for (int k = 0; k < 9; ++k) g += w2[((size_t)o * 64 + c) * 9 + k];
                    acc += g * t1[c];
                }
            b2f[o] = (float)acc;
        }
        std::vector<float> u4;
        pack_wino43_weights({w2}, {64}, 64, 64, u4, own ? nullptr : s1.data());
        int rc;
        if ((rc = upload(h, wf, &E.first.w_f1))) return rc;
        if ((rc = upload(h, bf, &E.first.bias_f1))) return rc;
        if ((rc = upload(h, u4, &E.conv[0].u43pack_f1))) return rc;
        if ((rc = upload(h, b2f, &E.conv[0].bias_f1))) return rc;
    }
    return MP_OK;
}

int pick_mbw(int H, int W)
{
    long long best = -1;
    int arg = 32;
    for (int mbw : {32, 16, 8}) {
        const int tw = mbw, th = 256 / mbw;
        const long long area = (long long)((H + th - 1) / th) * th * ((W + tw - 1) / tw) * tw;
        if (best < 0 || area < best) { best = area; arg = mbw; }
    }
    return arg;
}

void prof_begin(mp_handle* h, const char* name, double flop, hipStream_t s)
{
    if (!h->prof) return;
    if (h->prof_used == h->prof_entries.size()) {
        ProfEntry e{};
        (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b);
        h->prof_entries.push_back(e);
    }
    ProfEntry& e = h->prof_entries[h->prof_used];
    e.name = name; e.flop = flop;
    (void)hipEventRecord(e.a, s);
}

void prof_end(mp_handle* h, hipStream_t s)
{
    if (!h->prof) return;
    (void)hipEventRecord(h->prof_entries[h->prof_used].b, s);
    ++h->prof_used;
}

// 0, or MP_EINVAL (with the handle's error text set) when the launch is beyond the kernels' 32-bit tile decode
int too_large(mp_handle* h, const char* name, int B, int H, int W)
{
    return fail(h, MP_EINVAL, std::string("mp_forward: layer ") + name + " has too many work items for one launch (B=" +
                                  std::to_string(B) + ", " + std::to_string(H) + "x" + std::to_string(W) +
                                  "): split the batch");
}

// launcher return codes: 0 launched; 1 more work items than the 32-bit tile decode addresses; 2 a layer shape the selected
// kernel is not instantiated for (a dispatch bug: run_conv only selects kernels whose *_supports() said yes)
int launch_failed(mp_handle* h, int code, const char* name, int B, int H, int W)
{
    if (code == 1) return too_large(h, name, B, H, W);
    return fail(h, MP_EINVAL, std::string("mp_forward: layer ") + name + ": the selected convolution kernel does not cover this "
                                  "layer shape (" + std::to_string(H) + "x" + std::to_string(W) + ")");
}

// which F(4x4,3x3) kernel run_conv() sends this 3x3 layer at H x W to: 0 none, 1 conv_wino43.hip (two waves per SIMD; reflection
// padding and frames that are multiples of the 4x4 tile; the only one that evaluates the first block inside the launch), 2
// conv_wino43b.hip (one wave per SIMD; any frame size, reflection or zero padding).  MP_DEBUG=wino43_gen: 0 (default) the first where
// it applies and the second otherwise, 1 / 2 only that one.
int wino43_kind(const mp_handle* h, const ConvLayer& L, int H, int W, bool fuse, int in_cstride = 0, int in_coff = 0,
                int out_cstride = 0, int out_coff = 0)
{
    if (!(L.taps == 9 && L.u43pack && h->wino && h->wino43 == 2)) return 0;
    ConvParams q{};
    q.pad_zero = h->cfg.reflection_pad ? 0 : 1; q.cin = L.cin; q.cout = L.cout; q.H = H; q.W = W;
    q.in_cstride = in_cstride; q.in_coff = in_coff; q.out_cstride = out_cstride; q.out_coff = out_coff;
    const bool g1 = h->wino43_gen != 2 && conv_wino43_supports(q), g2 = h->wino43_gen != 1 && conv_wino43b_supports(q);
    // fuse: the first block is evaluated by the layer's kernel -- only by the pooled 64 -> 64 layer, 64 real channels
    if (fuse) return h->fuse43 && g1 && L.pool && L.cin == 64 && L.cout == 64 && h->cfg.channel_version == 0 && L.u43pack_f1 ? 1 : 0;
    return g1 ? 1 : g2 ? 2 : 0;
}
bool uses_wino43(const mp_handle* h, const ConvLayer& L, int H, int W, bool fuse, int in_cstride = 0, int in_coff = 0,
                 int out_cstride = 0, int out_coff = 0)
{
    return wino43_kind(h, L, H, W, fuse, in_cstride, in_coff, out_cstride, out_coff) != 0;
}

int run_conv(mp_handle* h, const ConvLayer& L, const float* in, int in_cstride, int in_coff, float* out,
              int out_cstride, int out_coff, int B, int H, int W, const int* img_list, hipStream_t s,
              const FirstLayer* fuse = nullptr, const float* images = nullptr, int in_planar = 0, int out_planar = 0)
{
    ConvParams p{};
    p.in = in; p.out = out; p.wpack = L.wpack; p.bias = L.bias; p.scale = L.scale; p.shift = L.shift;
    p.img_list = img_list;
    p.B = B; p.H = H; p.W = W;
    p.in_cstride = in_cstride; p.in_coff = in_coff; p.cin = L.cin;
    p.out_cstride = out_cstride; p.out_coff = out_coff; p.cout = L.cout;
    p.nslices = L.nslices;
    p.pad_zero = h->cfg.reflection_pad ? 0 : 1;
    p.bn_first = h->cfg.bn_first;
    p.relu = L.relu ? 1 : 0;
    p.persist = h->persist;
    p.ncu = h->ncu; p.xcd_shift = h->xcd_shift;
    int mbw = 32;
    if (L.taps == 9) {
        mbw = pick_mbw(H, W);
        const int tw = mbw, th = 256 / mbw;
        p.tiles_x = (W + tw - 1) / tw; p.tiles_y = (H + th - 1) / th;
    } else {
        p.total_px = (long long)B * H * W;
    }
    const int f43 = wino43_kind(h, L, H, W, fuse != nullptr, in_cstride, in_coff, out_cstride, out_coff);
    prof_begin(h, fuse ? "enc.conv1+2" : L.name,
               2.0 * L.taps * L.cin * L.cout * (double)B * H * W + (fuse ? 2.0 * 9 * 64 * (double)B * H * W : 0.0), s);
    if (fuse) { p.img = images; p.w1 = fuse->w; p.b1 = fuse->bias; p.s1 = fuse->scale; p.t1 = fuse->shift; }
    int big;
    if (f43) {
        p.wpack = L.u43pack; p.in_planar = in_planar; p.out_planar = out_planar;
        if (fuse) {       // the fused launch: the first block's BatchNorm is folded away (build_encoder)
            p.wpack = L.u43pack_f1; p.bias = L.bias_f1; p.w1 = fuse->w_f1; p.b1 = fuse->bias_f1; p.s1 = nullptr; p.t1 = nullptr;
        }
        if (!fuse && h->fwd_batch <= 2 && h->splitk_max > 1) {
            // single-pair latency (the reference's shipped batchsize: 1): a launch with fewer items than half the CUs (conv7 /
            // conv8 of one 480x640 pair: 40 items of 32 units on 256 CUs) cuts the input channels into 2, 4 or 8 ranges --
            // (cin / 4) / ranges units each, even and >= 4 -- as long as the items still fit the machine once.  Only for
            // forwards of one or two images: the ranges are summed in another order than one accumulator chain would, and a
            // batched forward must not change its bits with the batch size (tests: HA grouping, sharded == single-rank)
            const long long items = conv_wino43_items(p);
            const int units = L.cin / 4;
            int ks = 0;
            while ((items << (ks + 1)) <= h->ncu && (2 << ks) <= h->splitk_max && (units >> (ks + 1)) >= 4 &&
                   ((units >> (ks + 1)) & 1) == 0 && ((units >> (ks + 1)) << (ks + 1)) == units) ++ks;
            if (ks > 0 && items <= 1024) {
                int rc = ensure(h, h->split_ws, (size_t)(items << ks) * (2 * 16 * 512 * 8));
                if (rc) return rc;
                p.ks_shift = ks; p.split_scratch = static_cast<float*>(h->split_ws.p);
            }
        }
        if (f43 == 1 && !fuse && !L.pool && p.ks_shift == 0 && !in_planar && L.cin % 16 == 0 && L.cin <= 256 && 256 % (L.cin / 2) == 0 &&
            L.cin >= 16 && h->vin_min_slices > 0 &&
            L.nslices >= h->vin_min_slices) {
            // many output slices over one input (heads: 512 couts = 8 slices): transform the input ONCE (conv_wino43.hip VIN)
            // The pre-transformed input is an OPTIONAL workspace (2.25 x the layer's input, linear in B): without it the kernel
            // transforms per slice, bit-identically -- so an allocation failure here is not a failure of the forward
            const int rc = ensure(h, h->vin_ws, (size_t)conv_wino43_vglobal_floats(p) * 4);
            if (rc == MP_OK) p.vglobal = static_cast<float*>(h->vin_ws.p);
            else { (void)hipGetLastError(); h->err.clear(); }
        }
        big = f43 == 2 ? launch_conv_wino43b(p, L.pool, s) : launch_conv_wino43(p, L.pool, s, fuse != nullptr);
    } else {
        big = launch_conv_mfma(p, L.taps, mbw, L.pool, fuse != nullptr, s);
    }
    prof_end(h, s);
    return big ? launch_failed(h, big, L.name, B, H, W) : MP_OK;
}

int run_conv_h(mp_handle* h, const ConvLayer& L, const _Float16* in, int in_cstride, int in_coff, _Float16* out,
                int out_cstride, int out_coff, int B, int H, int W, const int* img_list, hipStream_t s,
                const FirstLayer* fuse = nullptr, const float* images = nullptr)
{
    ConvParamsH p{};
    p.in = in; p.out = out; p.wpack = L.wpack_h; p.bias = L.bias_h; p.scale = L.scale; p.shift = L.shift;
    p.img_list = img_list;
    p.B = B; p.H = H; p.W = W;
    p.in_cstride = in_cstride; p.in_coff = in_coff; p.cin = L.cin;
    p.out_cstride = out_cstride; p.out_coff = out_coff; p.cout = L.cout;
    p.nslices = L.nslices;
    p.pad_zero = h->cfg.reflection_pad ? 0 : 1;
    p.bn_first = h->cfg.bn_first;
    p.dummy = static_cast<_Float16*>(h->dummy);
    p.ncu = h->ncu; p.xcd_shift = h->xcd_shift; p.res_groups = h->f16_res_groups;
    int mbw = 32;
    if (L.taps == 9) {
        mbw = pick_mbw(H, W);
        const int tw = mbw, th = 256 / mbw;
        p.tiles_x = (W + tw - 1) / tw; p.tiles_y = (H + th - 1) / th;
    } else {
        p.total_px = (long long)B * H * W;
    }
    if (fuse) { p.img = images; p.w1 = fuse->w_h; p.b1 = fuse->bias_h; p.s1 = fuse->scale; p.t1 = fuse->shift; }
    prof_begin(h, fuse ? "enc.conv1+2" : L.name,
               2.0 * L.taps * L.cin * L.cout * (double)B * H * W + (fuse ? 2.0 * 9 * 64 * (double)B * H * W : 0.0), s);
    int big = 0;
    if (h->f16_res && conv_f16_res_supports(p, L.taps)) {
        big = launch_conv_f16_res(p, mbw, L.pool, s);
    } else if (h->f16_res && !fuse && L.taps == 9 && L.cin == 64 && L.nslices > 1 && L.cout == 64 * L.nslices) {
        // 64 input channels, several 64-channel output slices (enc.conv5): a slice's packed weights are 72 KiB, so the
        // LDS-resident-weights kernel runs once per slice (the input is read once per slice: cheaper than streaming the weights)
        for (int sl = 0; sl < L.nslices && !big; ++sl) {
            ConvParamsH q = p;
            q.wpack = p.wpack + (size_t)sl * 36 * 2 * 64 * 8;
            q.bias = p.bias + 64 * sl; q.scale = p.scale + 64 * sl; q.shift = p.shift + 64 * sl;
            q.out_coff = out_coff + 64 * sl; q.cout = 64; q.nslices = 1;
            big = conv_f16_res_supports(q, L.taps) ? launch_conv_f16_res(q, mbw, L.pool, s) : 2;
        }
    } else {
        big = launch_conv_f16(p, L.taps, mbw, L.pool, s);
    }
    prof_end(h, s);
    return big ? launch_failed(h, big, L.name, B, H, W) : MP_OK;
}

// mixed_precision forward: fp16 activations end to end, fp32 softmax / descriptor normalisation
int forward_f16(mp_handle* h, const float* images, int B, int H, int W, int nsets, const int* counts,
                const int* const* lptr, float* prob, float* logits, float* desc, hipStream_t s)
{
    const int Hc = H / 8, Wc = W / 8;
    const long long npx = (long long)B * Hc * Wc;
    const int D = h->cfg.descriptor_size;
    const int hc = h->head_channels;                                 // 256 (channel_version 0) or descriptor_size
    const int headc = h->cfg.descriptor_head ? 2 * hc : hc;
    const int encc = h->heads3.cin;                                  // encoder output channels incl. padding: 128 (64 for channel_version 2)
    // same carve-up as the fp32 path (sizes in elements), element type fp16
    const size_t nP = (size_t)B * H * W * 64, nQ = (size_t)B * H * W * 16;
    const size_t nL = (size_t)npx * 128, nD = (size_t)npx * 128, nR = (size_t)npx * 256;
    int rc;
    if (!h->dummy) MP_HIP(hipMalloc(&h->dummy, 4096));
    _Float16* P = static_cast<_Float16*>(h->ws.p);
    _Float16* Q = P + nP;
    _Float16* Lg = Q + nQ;
    _Float16* X = Lg + nL;
    _Float16* R = X + nD;        // raw (un-normalised) descriptors
    (void)nR;
    for (int e = 0; e < nsets; ++e) {
        const int nb = counts[e];
        if (nb == 0) continue;
        const Encoder& E = h->enc[e];
        // the first block inside the conv2 launch (conv_f16_res.hip F1): reflection padding, the LDS-resident-weights kernel
        const bool fuse1 = h->f16_res && h->f16_fuse1 && h->cfg.reflection_pad && !E.first_pool && E.conv[0].pool &&
                           E.conv[0].cin == 64 && E.conv[0].cout == 64 && E.conv[0].nslices == 1;
        if (!fuse1) {
            Conv1ParamsH c1{};
            c1.in = images; c1.out = P; c1.w = E.first.w_h; c1.bias = E.first.bias_h; c1.scale = E.first.scale;
            c1.shift = E.first.shift; c1.img_list = lptr[e]; c1.B = nb; c1.H = H; c1.W = W;
            c1.pad_zero = h->cfg.reflection_pad ? 0 : 1; c1.bn_first = h->cfg.bn_first;
            c1.pool = E.first_pool ? 1 : 0;                          // double_convolution: false -- MaxPool2d follows the block directly
            prof_begin(h, "enc.conv1", 2.0 * 9 * 64 * (double)nb * H * W, s);
            launch_conv_first_f16(c1, s);
            prof_end(h, s);
        }
        int hh = E.first_pool ? H / 2 : H, ww = E.first_pool ? W / 2 : W;
        _Float16* src = P;
        _Float16* dst = Q;
        for (int i = 0; i < E.nconv; ++i) {
            const ConvLayer& L = E.conv[i];
            if ((rc = run_conv_h(h, L, src, L.cin, 0, i == E.nconv - 1 ? X : dst, L.cout, 0, nb, hh, ww, lptr[e], s,
                                 (i == 0 && fuse1) ? &E.first : nullptr, images))) return rc;
            if (L.pool) { hh /= 2; ww /= 2; }
            _Float16* t = src; src = dst; dst = t;
        }
    }
    if ((rc = run_conv_h(h, h->heads3, X, encc, 0, P, headc, 0, B, Hc, Wc, nullptr, s))) return rc;
    if (h->head_fuse) {
        // both 1x1 convolutions + BN + softmax / shuffle + normalisation in ONE launch that reads P once (head_tail_f16.hip)
        HeadTailParamsH t{};
        t.x = P; t.xstride = headc; t.K = hc;
        t.wdet = h->det1.wpack_h; t.bdet = h->det1.bias_h; t.sdet = h->det1.scale; t.tdet = h->det1.shift;
        t.wdesc = h->desc1.wpack_h; t.bdesc = h->desc1.bias_h; t.sdesc = h->desc1.scale; t.tdesc = h->desc1.shift;
        t.D = D; t.npx = npx; t.B = B; t.Hc = Hc; t.Wc = Wc;
        t.prob = prob; t.logits_nchw = logits; t.desc = h->cfg.descriptor_head ? desc : nullptr;
        t.softmax_mode = h->cfg.softmax_mode; t.normalize = h->cfg.normalize_descriptors ? 1 : 0; t.ncu = h->ncu;
        if (t.wdet && (!t.desc || t.wdesc) && (prob || logits || t.desc)) {
            prof_begin(h, "heads.tail", 2.0 * hc * (65.0 + (t.desc ? D : 0)) * (double)npx, s);
            const int miss = launch_head_tail_f16(t, s);
            prof_end(h, s);
            if (!miss) { MP_HIP(hipGetLastError()); return MP_OK; }
            if (h->prof && h->prof_used) --h->prof_used;      // not covered: the separate launches below are profiled instead
        }
    }
    if ((rc = run_conv_h(h, h->det1, P, headc, 0, Lg, 128, 0, B, Hc, Wc, nullptr, s))) return rc;
    if (prob || logits) {
        prof_begin(h, "det.softmax_shuffle", 0.0, s);
        launch_det_post_f16(Lg, 128, B, Hc, Wc, prob, logits, h->cfg.softmax_mode, s);
        prof_end(h, s);
    }
    if (desc) {
        if ((rc = run_conv_h(h, h->desc1, P, headc, hc, R, D, 0, B, Hc, Wc, nullptr, s))) return rc;
        prof_begin(h, "desc.l2norm", 0.0, s);
        launch_desc_l2norm_f16(R, desc, npx, D, h->cfg.normalize_descriptors ? 1 : 0, s);
        prof_end(h, s);
    }
    MP_HIP(hipGetLastError());
    return MP_OK;
}

bool footprint(float size, double iou, NmsFootprint& fp)
{
    // torchvision nms CPU kernel arithmetic (fp32) for two size x size boxes offset by (dy,dx):
    //   inter = max(0, size-|dy|) * max(0, size-|dx|); ovr = inter / (area + area - inter) > iou
    // -- the last comparison in DOUBLE: nms_kernel_impl(dets, scores, double iou_threshold) promotes the fp32 ovr (include/multipoint_hip.h: mp_box_nms)
    int R = (int)std::ceil(size) - 1;
    if (R < 0) R = 0;
    if (R > MP_NMS_MAX_R) return false;
    fp.R = R;
    const float half = size * 0.5f;
    for (int dy = -R; dy <= R; ++dy) {
        unsigned m = 0;
        for (int dx = -R; dx <= R; ++dx) {
            // boxes [y-half, x-half, y+half, x+half] at a generic in-image position
            const float y1a = 100.f - half, x1a = 100.f - half, y2a = 100.f + half, x2a = 100.f + half;
            const float y1b = (100.f + dy) - half, x1b = (100.f + dx) - half;
            const float y2b = (100.f + dy) + half, x2b = (100.f + dx) + half;
            const float area_a = (y2a - y1a) * (x2a - x1a), area_b = (y2b - y1b) * (x2b - x1b);
            float w = std::fmin(y2a, y2b) - std::fmax(y1a, y1b); if (w < 0.f) w = 0.f;
            float hh = std::fmin(x2a, x2b) - std::fmax(x1a, x1b); if (hh < 0.f) hh = 0.f;
            const float inter = w * hh;
            const float ovr = inter / (area_a + area_b - inter);
            if ((double)ovr > iou) m |= 1u << (dx + R);
        }
        fp.rowmask[dy + R] = m;
    }
    return true;
}

int nms_common(mp_handle* h, const float* prob, const unsigned char* mask, int B, int H, int Wc,
               float size, float min_prob, double iou, int topk, int K, int* kp_yx, float* kp_score,
               int* kp_count, float* prob_nms, int max_rounds, hipStream_t s)
{
    if (B <= 0 || H <= 0 || Wc <= 0)
        return fail(h, MP_EINVAL, "box_nms: need B,H,W > 0");
    NmsFootprint fp{};
    if (!footprint(size, iou, fp))
        return fail(h, MP_EINVAL, "box_nms: box size > " + std::to_string(MP_NMS_MAX_R + 1) + " unsupported");
    // the work map's rows are the caller's rounded up to a multiple of 4 floats (the kernels move 16-byte groups); the padding
    // columns are never candidates, and row-major order -- the tie-break -- is the same in both geometries.  Wc % 4 != 0 (any H x W
    // is a legal argument of utils.box_nms, utils.py:90-91): round 0 reads the map with the generic kernel's scalar loads
    const int W = (Wc + 3) & ~3;
    const long long n = (long long)B * H * W;
    const int ntiles = B * ((W + 31) / 32) * ((H + 31) / 32);
    // workspace: work map | list_idx | list_score
    const size_t bytes = (size_t)n * 4 * 3;
    int rc;
    if ((rc = ensure(h, h->ws2, bytes))) return rc;
    if ((rc = ensure(h, h->nms_state, (size_t)(64 + 2 * ntiles) * 4))) return rc;
    if ((rc = ensure(h, h->kp_scratch, keypoint_scratch_ints(B, H, W) * 4))) return rc;
    float* work = static_cast<float*>(h->ws2.p);
    int* list_idx = reinterpret_cast<int*>(work + n);
    float* list_score = work + 2 * n;
    int* remaining = static_cast<int*>(h->nms_state.p);
    // footprint tie guard: per-image counters the rounds add to and launch_select_keypoints reads and clears
    int* pairs = nullptr;
    if (h->tie_pairs_min > 0) {
        if (h->tie_pairs_cap < B) {
            if (h->tie_pairs) { MP_HIP(hipStreamSynchronize(s)); (void)hipFree(h->tie_pairs); h->tie_pairs = nullptr; h->tie_pairs_cap = 0; }
            const int cap = B < 256 ? 256 : B;
            MP_HIP(hipMalloc(reinterpret_cast<void**>(&h->tie_pairs), (size_t)cap * 4));
            MP_HIP(hipMemsetAsync(h->tie_pairs, 0, (size_t)cap * 4, s));
            h->tie_pairs_cap = cap;
        }
        pairs = h->tie_pairs;
    }
    // the candidate listing (prob * mask > min_prob) is fused into round 0, which reads the probability map itself
    // Rounds: a fixed number without any host read (max_rounds > 0, at most 64), or groups of 8 with one 4-byte read
    // of the undecided count after each group until it is zero (max_rounds == 0).  A round settles every chain of
    // dependent decisions inside a 32 x 32 tile, so the count of rounds is the longest chain measured in tiles: a
    // handful for detector maps, W / 32 for a monotone ramp across the frame -- hence the generous cap.
    int round = 0;
    const int per = max_rounds > 0 ? (max_rounds < 64 ? max_rounds : 64) : 8;
    const int cap = max_rounds > 0 ? per : 4096;
    for (;;) {
        for (int r = 0; r < per && round < cap; ++r, ++round) {
            if (round == 0) launch_nms_round0(prob, mask, min_prob, work, B, H, W, fp, remaining, s, Wc, h->tie_eps, pairs);
            else launch_nms_round(work, B, H, W, fp, remaining, round, s, h->tie_eps, pairs);
        }
        if (max_rounds > 0 || round >= cap) break;
        launch_nms_accumulate(remaining, B, H, W, round - 1, nullptr, s);          // the tiles' undecided counts -> the round's slot
        MP_HIP(hipMemcpyAsync(h->pinned, remaining + ((round - 1) & 63), 4, hipMemcpyDeviceToHost, s));
        MP_HIP(hipStreamSynchronize(s));
        if (h->pinned[0] == 0) break;
    }
    h->last_nms_rounds = round;
    if (!h->nms_total) {
        MP_HIP(hipMalloc(reinterpret_cast<void**>(&h->nms_total), 4));
        MP_HIP(hipMemsetAsync(h->nms_total, 0, 4, s));
    }
    launch_nms_accumulate(remaining, B, H, W, round - 1, h->nms_total, s);
    int* tie = nullptr;
    if ((topk > 0 && h->tie_min > 0) || pairs) {
        if (!h->tie_state) {
            MP_HIP(hipMalloc(reinterpret_cast<void**>(&h->tie_state), (1 + MP_TIE_MAX_IMAGES) * 4));
            MP_HIP(hipMemsetAsync(h->tie_state, 0, (1 + MP_TIE_MAX_IMAGES) * 4, s));
        }
        tie = h->tie_state;
        h->tie_last_B = B < MP_TIE_MAX_IMAGES ? B : MP_TIE_MAX_IMAGES;
    } else {
        h->tie_last_B = 0;
    }
    launch_select_keypoints(work, B, H, W, topk, K, list_idx, list_score, H * W, kp_yx, kp_score, kp_count,
                            prob_nms, static_cast<int*>(h->kp_scratch.p), s, h->tie_eps, topk > 0 ? h->tie_min : 0, tie, Wc,
                            pairs, h->tie_pairs_min);
    MP_HIP(hipGetLastError());
    if (max_rounds == 0 && round >= cap) {
        MP_HIP(hipMemcpyAsync(h->pinned, remaining + ((round - 1) & 63), 4, hipMemcpyDeviceToHost, s));
        MP_HIP(hipStreamSynchronize(s));
        if (h->pinned[0] != 0) return fail(h, MP_ESTATE, "box_nms did not converge within 4096 rounds");
    }
    return MP_OK;
}

}  // namespace

// =============================================================================================
namespace {
// XCC (= XCD) count of the KFD topology node whose PCI location matches `bus_id` ("dddd:bb:dd.f"); 0 if the topology is not
// readable (containers without /sys/class/kfd): the caller then falls back to compute units / 32
// MP_DEBUG: the ONE environment variable the library reads (in mp_create), a comma-separated list of developer switches
// `key` or `key=value` -- kernel selection for A/B runs and for the parity tests, which hold every kernel variant to the CPU reference path.
// They are not configuration: a model's algorithm is `model.conv_algorithm` / `model.batch_invariant` (mp_model_config).
//   no_winograd          the direct implicit-GEMM kernels for every 3x3 layer (what conv_algorithm 3 selects per model)
//   wino43=0             F(4x4,3x3) for no layer (round 6 retired the value 1 = 64-input-channel layers only: no routing selects it)
//   wino43_gen=0|1|2     0 conv_wino43.hip where it applies and conv_wino43b.hip elsewhere; 1 / 2: only that kernel
//   no_fuse              the fp32 first block as its own launch in front of the direct conv2 kernel
//   no_fuse43            ... in front of the F(4x4,3x3) conv2 kernel
//   no_head_fuse         separate 1x1 convolution / softmax / normalisation launches instead of the fused head tail (fp32 and fp16)
//   no_vin               the 3x3 head convolutions (>= 4 output slices) transform their input per slice inside the kernel instead of
//                        taking it pre-transformed from a pass of its own (conv_wino43.hip VIN; bit-identical either way)
//   no_planar            channel-quad-planar tensors never (default: behind conv1 and pooled producers; round 6 retired planar=2 = everywhere)
//   no_persist, persist_min_items=N   direct kernels: per-tile launches / persistent from N items per CU
//   splitk_max=1..8      most ranges the input channels of a small launch are cut into
//   f16_no_res, f16_no_fuse1, f16_res_groups=2   fp16 path: streaming kernel everywhere / first block as its own launch / two groups
//   ncu=N, nxcd=N        emulate a partitioned device (fewer persistent workgroups, same results)
// Returns true when `key` is present; *value receives the integer behind '=' (or `dflt` for a bare key).
bool debug_switch(const char* key, int* value = nullptr, int dflt = 1)
{
    const char* e = getenv("MP_DEBUG");
    if (!e) return false;
    const size_t kl = std::strlen(key);
    for (const char* q = e; *q;) {
        while (*q == ',' || *q == ' ') ++q;
        const char* end = q;
        while (*end && *end != ',') ++end;
        if ((size_t)(end - q) >= kl && std::strncmp(q, key, kl) == 0 && (q[kl] == '=' || q + kl == end || q[kl] == ' ')) {
            if (value) *value = q[kl] == '=' ? atoi(q + kl + 1) : dflt;
            return true;
        }
        q = end;
    }
    return false;
}

int kfd_num_xcc(const char* bus_id)
{
    unsigned dom = 0, bus = 0, dev = 0, fn = 0;
    if (sscanf(bus_id, "%x:%x:%x.%x", &dom, &bus, &dev, &fn) != 4) return 0;
    const unsigned long long want = ((unsigned long long)bus << 8) | (dev << 3) | fn;
    for (int node = 0; node < 64; ++node) {
        char path[128];
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", node);
        FILE* f = fopen(path, "r");
        if (!f) { if (node > 8) break; else continue; }
        char key[64]; unsigned long long val = 0, loc = ~0ull, domain = 0, xcc = 0, simd = 0;
        while (fscanf(f, "%63s %llu", key, &val) == 2) {
            if (!strcmp(key, "location_id")) loc = val;
            else if (!strcmp(key, "domain")) domain = val;
            else if (!strcmp(key, "num_xcc")) xcc = val;
            else if (!strcmp(key, "simd_count")) simd = val;
        }
        fclose(f);
        if (simd > 0 && loc == want && domain == dom) return (int)xcc;
    }
    return 0;
}
}  // namespace

extern "C" {

const char* mp_version(void) { return "multipoint_hip 0.1 (gfx950)"; }

const char* mp_last_error(const mp_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mp_device_shape(const mp_handle* h, int* compute_units, int* xcds, int* persistent_workgroups)
{
    if (!h) return MP_EINVAL;
    if (compute_units) *compute_units = h->ncu;
    if (xcds) *xcds = 1 << h->xcd_shift;
    if (persistent_workgroups) *persistent_workgroups = (int)persistent_grid(1ll << 40, h->ncu, h->xcd_shift);
    return MP_OK;
}

int mp_create(mp_handle** out, int device)
{
    mp_handle* h = nullptr;
    if (!out) return fail(h, MP_EINVAL, "mp_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    MP_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(h, MP_EINVAL, "mp_create: device " + std::to_string(device) + " not available (" +
                                      std::to_string(ndev) + " HIP devices visible)");
    MP_HIP(hipSetDevice(device));
    {
        // Rounds 1-4 read ~20 MP_* kernel-selection variables; round 5 folded them into MP_DEBUG=key[=value],... .  A script that
        // still sets an old name would silently A/B the default against itself: say so once.
        static const char* const legacy[] = {"MP_NO_WINOGRAD", "MP_WINO43", "MP_WINO43_GEN", "MP_NO_FUSE", "MP_NO_FUSE43", "MP_NO_HEAD_FUSE",
            "MP_NO_PLANAR", "MP_PLANAR", "MP_NO_PERSIST", "MP_PERSIST_MIN_ITEMS", "MP_SPLITK_MAX", "MP_F16_NO_RES", "MP_F16_NO_FUSE1",
            "MP_F16_RES_GROUPS", "MP_NCU", "MP_NXCD", "MP_POST_OVERLAP", "MP_NO_VIN", "MP_NO_POOL_FIRST"};
        static bool warned = false;
        for (const char* k : legacy)
            if (!warned && getenv(k)) {
                warned = true;
                fprintf(stderr, "libmultipoint_hip: the environment variable %s is no longer read (nor are the other MP_* kernel switches): "
                                "use MP_DEBUG=key[=value],... -- the keys are listed in csrc/api.hip (debug_switch)\n", k);
            }
    }
    hipDeviceProp_t prop;
    MP_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(h, MP_EINVAL, std::string("mp_create: kernels are built for gfx950 only, device is ") +
                                      prop.gcnArchName);
    // Machine shape: every persistent kernel launches one (fp16: two) workgroup(s) per compute unit and walks, per XCD, a
    // contiguous share of the work items (workgroup b is dispatched to XCD b mod nxcd; each XCD has its own L2).  Both numbers
    // come from the device: multiProcessorCount, and the XCC count of the KFD topology node at the device's PCI address
    // (a partitioned MI355X -- DPX / QPX / CPX -- reports 4 / 2 / 1 XCDs with 128 / 64 / 32 CUs).  MP_DEBUG=ncu / MP_DEBUG=nxcd override
    // (tests emulate smaller partitions on the full device: fewer workgroups, same results).
    int ncu = prop.multiProcessorCount, nxcd = 0;
    {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof bus, device) == hipSuccess) nxcd = kfd_num_xcc(bus);
        if (nxcd <= 0) nxcd = ncu >= 32 ? ncu / 32 : 1;           // gfx950: 32 active CUs per XCD
        int v = 0;
        if (debug_switch("ncu", &v) && v > 0 && v <= ncu) ncu = v;
        if (debug_switch("nxcd", &v) && v > 0) nxcd = v;
    }
    if (ncu < 1 || nxcd < 1 || (nxcd & (nxcd - 1)) != 0 || nxcd > ncu)
        return fail(h, MP_EINVAL, "mp_create: unsupported machine shape: " + std::to_string(ncu) + " compute units in " +
                                      std::to_string(nxcd) + " XCDs (the XCD count must be a power of two <= the CU count)");
    mp_handle* hh = new mp_handle();
    hh->device = device;
    hh->ncu = ncu;
    hh->xcd_shift = 0;
    while ((1 << hh->xcd_shift) < nxcd) ++hh->xcd_shift;
    {
        int v = 0;
        hh->fuse_first = !debug_switch("no_fuse");
        hh->wino = !debug_switch("no_winograd");
        hh->fuse43 = !debug_switch("no_fuse43");
        hh->head_fuse = !debug_switch("no_head_fuse");
        hh->f16_res = !debug_switch("f16_no_res");
        hh->f16_fuse1 = !debug_switch("f16_no_fuse1");
        if (debug_switch("f16_res_groups", &v) && v == 2) hh->f16_res_groups = 2;
        if (debug_switch("no_planar")) hh->planar = 0;
        if (debug_switch("wino43", &v) && v == 0) hh->wino43 = 0;
        if (debug_switch("wino43_gen", &v) && v >= 0 && v <= 2) hh->wino43_gen = v;
        if (debug_switch("persist_min_items", &v) && v > 0) hh->persist = v;
        if (debug_switch("no_persist")) hh->persist = 0;
        if (debug_switch("splitk_max", &v) && v >= 1 && v <= 8) hh->splitk_max = v;
        if (debug_switch("no_vin")) hh->vin_min_slices = 0;
    }
    hh->splitk_env = hh->splitk_max;
    hh->wino_env = hh->wino; hh->wino43_env = hh->wino43; hh->wino43_gen_env = hh->wino43_gen;
    if (hipHostMalloc(reinterpret_cast<void**>(&hh->pinned), 4096) != hipSuccess) {
        delete hh;
        return fail(h, MP_ENOMEM, "mp_create: hipHostMalloc failed");
    }
    *out = hh;
    return MP_OK;
}

void mp_destroy(mp_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    free_weights(h);
    if (h->ws.p) (void)hipFree(h->ws.p);
    if (h->ws2.p) (void)hipFree(h->ws2.p);
    if (h->ws3.p) (void)hipFree(h->ws3.p);
    if (h->ws4.p) (void)hipFree(h->ws4.p);
    if (h->split_ws.p) (void)hipFree(h->split_ws.p);
    if (h->nms_state.p) (void)hipFree(h->nms_state.p);
    if (h->kp_scratch.p) (void)hipFree(h->kp_scratch.p);
    if (h->nms_total) (void)hipFree(h->nms_total);
    if (h->tie_state) (void)hipFree(h->tie_state);
    if (h->tie_pairs) (void)hipFree(h->tie_pairs);
    if (h->dummy) (void)hipFree(h->dummy);
    if (h->pinned) (void)hipHostFree(h->pinned);
    for (auto& e : h->prof_entries) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    delete h;
}

int mp_load_weights(mp_handle* h, const mp_model_config* cfg, const mp_tensor* tensors, int n_tensors)
{
    if (!h) return MP_EINVAL;
    if (!cfg || (!tensors && n_tensors > 0)) return fail(h, MP_EINVAL, "mp_load_weights: NULL argument");
    if (cfg->channel_version < 0 || cfg->channel_version > 2)
        return fail(h, MP_EINVAL, "unsupported model config: channel_version must be 0, 1 or 2 (MultiPoint.py:38-53)");
    if (cfg->channel_version != 0 && cfg->key_layout == 1)
        return fail(h, MP_EINVAL, "unsupported model config: SuperPointMagicLeap has channel_version 0 shapes");
    if (!cfg->double_convolution && cfg->key_layout == 1)
        return fail(h, MP_EINVAL, "unsupported model config: SuperPointMagicLeap has two convolutions per stage");
    if (cfg->descriptor_head && cfg->descriptor_size != 64 && cfg->descriptor_size != 128 &&
        cfg->descriptor_size != 256)
        return fail(h, MP_EINVAL, "unsupported model config: descriptor_size must be 64, 128 or 256");
    if (cfg->conv_algorithm < 0 || cfg->conv_algorithm > 3)
        return fail(h, MP_EINVAL, "unsupported model config: conv_algorithm must be 0 (auto), 1 (winograd43), 2 (winograd43_general) or 3 (direct)");
    MP_HIP(hipSetDevice(h->device));
    free_weights(h);
    h->cfg = *cfg;
    // the convolution algorithm of the 3x3 layers is a MODEL setting (yaml model.conv_algorithm); the MP_DEBUG developer switches of
    // mp_create only apply to 'auto'
    h->wino = h->wino_env; h->wino43 = h->wino43_env; h->wino43_gen = h->wino43_gen_env;     // 0 auto: what mp_create chose
    if (cfg->conv_algorithm == 1) { h->wino = true; h->wino43 = 2; h->wino43_gen = 0; }
    else if (cfg->conv_algorithm == 2) { h->wino = true; h->wino43 = 2; h->wino43_gen = 2; }
    else if (cfg->conv_algorithm == 3) { h->wino = false; }
    h->splitk_max = cfg->batch_invariant ? 1 : h->splitk_env;
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].name || (!tensors[i].data && tensors[i].numel > 0))
            return fail(h, MP_EINVAL, "mp_load_weights: tensor " + std::to_string(i) + " has NULL field");
        tm.m[tensors[i].name] = &tensors[i];
    }
    int rc;
    if (cfg->key_layout == 1 && (cfg->multispectral || cfg->batchnorm || cfg->final_batchnorm))
        return fail(h, MP_EINVAL, "unsupported model config: SuperPointMagicLeap key layout has one encoder and no BatchNorm");
    if (!cfg->batchnorm && cfg->key_layout == 0)
        return fail(h, MP_EINVAL, "unsupported model config: MultiPoint key layout always has BatchNorm2d");
    if (cfg->multispectral) {
        if ((rc = build_encoder(h, tm, h->enc[0], "encoder_thermal"))) return rc;
        if ((rc = build_encoder(h, tm, h->enc[1], "encoder_optical"))) return rc;
    } else {
        if ((rc = build_encoder(h, tm, h->enc[0], "encoder"))) return rc;
    }
    // head key names: MultiPoint nn.Sequential (MultiPoint.py:62-88) or SuperPointMagicLeap (:25-29)
    const bool ml = cfg->key_layout == 1;
    const std::string det = "detector_head_convolutions", dsc = "descriptor_head_convolutions";
    const std::string bn3 = cfg->bn_first ? ".2" : ".3";
    const std::string det3 = ml ? "convPa" : det + ".1", dsc3 = ml ? "convDa" : dsc + ".1";
    const std::string det1k = ml ? "convPb" : det + ".4", dsc1k = ml ? "convDb" : dsc + ".4";
    const std::string det3bn = cfg->batchnorm ? det + bn3 : std::string(), dsc3bn = cfg->batchnorm ? dsc + bn3 : std::string();
    const std::string det1bn = cfg->final_batchnorm ? det + ".5" : std::string();
    const std::string dsc1bn = cfg->final_batchnorm ? dsc + ".5" : std::string();
    // both 3x3 head convs read the same encoder output: one launch with N = hc (+hc); hc = 256 for channel_version 0,
    // descriptor_size otherwise (MultiPoint.py:38-53)
    const int hc = cfg->channel_version == 0 ? 256 : cfg->descriptor_size;
    const int enc_out = h->enc[0].conv[h->enc[0].nconv - 1].cout;                 // 128 (64 for channel_version 2)
    const int enc_real = cfg->channel_version == 2 ? 64 : 128;
    h->head_channels = hc;
    if (cfg->descriptor_head)
        rc = build_conv(h, tm, h->heads3, "heads.conv3x3", {det3, dsc3}, {det3bn, dsc3bn}, {hc, hc}, enc_out, 9, false, true, enc_real);
    else
        rc = build_conv(h, tm, h->heads3, "heads.conv3x3", {det3}, {det3bn}, {hc}, enc_out, 9, false, true, enc_real);
    if (rc) return rc;
    if ((rc = build_conv(h, tm, h->det1, "det.conv1x1", {det1k}, {det1bn}, {65}, hc, 1, false, false))) return rc;
    if (cfg->descriptor_head &&
        (rc = build_conv(h, tm, h->desc1, "desc.conv1x1", {dsc1k}, {dsc1bn}, {cfg->descriptor_size}, hc, 1, false, false)))
        return rc;
    // strict=True semantics of load_state_dict: no unexpected keys
    for (auto& kv : tm.m)
        if (!tm.used.count(kv.first)) {
            if (kv.first.size() > 20 && kv.first.rfind(".num_batches_tracked") == kv.first.size() - 20) continue;
            free_weights(h);
            return fail(h, MP_EINVAL, "unexpected key in state_dict: " + kv.first);
        }
    h->loaded = true;
    return MP_OK;
}

static int forward_checked(mp_handle* h, const float* images, const unsigned char* is_optical, int B, int H, int W,
                           float* prob, float* logits, float* desc, void* stream)
{
    if (!h->loaded) return fail(h, MP_ESTATE, "mp_forward: no weights loaded (call mp_load_weights)");
    if (!images || B <= 0 || H <= 0 || W <= 0) return fail(h, MP_EINVAL, "mp_forward: bad image tensor");
    if ((H % 8) != 0 || (W % 8) != 0)
        return fail(h, MP_EINVAL, "mp_forward: H and W must be divisible by 8 (got " + std::to_string(H) +
                                      "x" + std::to_string(W) + ")");
    if (desc && !h->cfg.descriptor_head) return fail(h, MP_EINVAL, "mp_forward: model has no descriptor head");
    if (h->cfg.multispectral && !is_optical)
        return fail(h, MP_EINVAL, "mp_forward: multispectral model needs is_optical");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MP_HIP(hipSetDevice(h->device));
    const int Hc = H / 8, Wc = W / 8;
    const long long npx = (long long)B * Hc * Wc;
    const int D = h->cfg.descriptor_size;
    const int hc = h->head_channels;
    const int headc = h->cfg.descriptor_head ? 2 * hc : hc;
    // workspace: P (B*H*W*64) | Q (B*H*W*16) | X encoder output (npx*128) | logits (npx*80) | img lists
    const size_t nP = (size_t)B * H * W * 64, nQ = (size_t)B * H * W * 16;
    const size_t nL = (size_t)npx * 80, nD = (size_t)npx * 128;
    int rc;
    if ((rc = ensure(h, h->ws, (nP + nQ + nL + nD) * 4 + 2 * 1024 * 4 + 256))) return rc;
    float* P = static_cast<float*>(h->ws.p);
    float* Q = P + nP;
    float* Lg = Q + nQ;
    float* X = Lg + nL;      // encoder output: separate from the ping-pong buffers, because with two
                             // encoders the second pass would overwrite the first pass's result
    int* lists = reinterpret_cast<int*>(X + nD);
    if (h->prof_used > 4000) h->prof_used = 0;      // profile ring: entries accumulate until read

    // encoder(s): multispectral routes each image by is_optical (MultiPoint.py:107-122)
    int nsets = 1, counts[2] = {B, 0};
    const int* lptr[2] = {nullptr, nullptr};
    h->fwd_batch = B;
    if (h->cfg.multispectral) {
        if (B > 512) return fail(h, MP_EINVAL, "mp_forward: multispectral B > 512 unsupported");
        nsets = 2;
        std::vector<int> host(1024, 0);      // [0..512) thermal image ids, [512..1024) optical ids
        counts[0] = counts[1] = 0;
        for (int b = 0; b < B; ++b) {
            if (is_optical[b]) host[512 + counts[1]++] = b;
            else host[counts[0]++] = b;
        }
        // pageable source: the runtime stages it before returning, so `host` may die here
        MP_HIP(hipMemcpyAsync(lists, host.data(), 1024 * 4, hipMemcpyHostToDevice, s));
        lptr[0] = lists; lptr[1] = lists + 512;
    }
    if (h->cfg.mixed_precision) return forward_f16(h, images, B, H, W, nsets, counts, lptr, prob, logits, desc, s);
    for (int e = 0; e < nsets; ++e) {
        const int nb = counts[e];
        if (nb == 0) continue;
        const Encoder& E = h->enc[e];
        Conv1Params c1{};
        c1.in = images; c1.out = P; c1.w = E.first.w; c1.bias = E.first.bias; c1.scale = E.first.scale;
        c1.shift = E.first.shift; c1.img_list = lptr[e]; c1.B = nb; c1.H = H; c1.W = W;
        c1.pad_zero = h->cfg.reflection_pad ? 0 : 1; c1.bn_first = h->cfg.bn_first;
        c1.channels = E.first.channels;
        // the fused loader is a 64-channel direct-convolution kernel; with Winograd on, the standalone first block +
        // Winograd second convolution is faster than the fused direct kernel
        const bool fuse1 = h->fuse_first && h->cfg.channel_version == 0 && !E.first_pool &&
                           (!h->wino || uses_wino43(h, E.conv[0], H, W, true));
        // a tensor written by conv1 or an F(4x4,3x3) layer AND read by an F(4x4,3x3) layer is channel-quad planar
        // -- when the producer's stores are few: conv1, or a POOLED F(4x4,3x3) layer.  (An un-pooled layer stores 16 pixels per
        // lane and tile; planar, a store instruction then writes 16-byte pieces 64 bytes apart instead of 64-byte runs, which
        // costs the producer more than the consumer's patch DMAs gain: conv3 1.29 vs 1.17 ms.)
        bool f43[8] = {}, pl[9] = {};               // pl[i]: the input tensor of E.conv[i] is planar
        const int H1 = E.first_pool ? H / 2 : H, W1 = E.first_pool ? W / 2 : W;      // frame of the first block's output
        for (int i = 0, hh = H1, ww = W1; i < E.nconv; ++i) {
            f43[i] = uses_wino43(h, E.conv[i], hh, ww, i == 0 && fuse1);
            if (E.conv[i].pool) { hh /= 2; ww /= 2; }
        }
        pl[0] = h->planar && f43[0] && !E.first_pool;
        for (int i = 1; i < E.nconv; ++i) pl[i] = h->planar && f43[i - 1] && f43[i] && E.conv[i - 1].pool;
        c1.out_planar = pl[0] ? 1 : 0;
        c1.pool = E.first_pool ? 1 : 0;
        if (!fuse1) {
            prof_begin(h, "enc.conv1", 2.0 * 9 * 64 * (double)nb * H * W, s);
            launch_conv_first(c1, s);
            prof_end(h, s);
        }
        int hh = H1, ww = W1;
        float* src = P;
        float* dst = Q;
        for (int i = 0; i < E.nconv; ++i) {
            const ConvLayer& L = E.conv[i];
            if ((rc = run_conv(h, L, src, L.cin, 0, i == E.nconv - 1 ? X : dst, L.cout, 0, nb, hh, ww, lptr[e], s,
                     (i == 0 && fuse1) ? &E.first : nullptr, images, pl[i], pl[i + 1]))) return rc;
            if (L.pool) { hh /= 2; ww /= 2; }
            float* t = src; src = dst; dst = t;
        }
    }
    // heads
    if ((rc = run_conv(h, h->heads3, X, h->heads3.cin, 0, P, headc, 0, B, Hc, Wc, nullptr, s))) return rc;       // (a planar encoder output was measured: slower)
    if (h->head_fuse) {
        // both 1x1 convolutions + BN + softmax / shuffle + normalisation in ONE launch that reads P once (head_tail.hip)
        HeadTailParams t{};
        t.ncu = h->ncu;
        t.x = P; t.xstride = headc; t.K = hc;
        t.wdet = h->det1.wpack; t.bdet = h->det1.bias; t.sdet = h->det1.scale; t.tdet = h->det1.shift;
        t.wdesc = h->desc1.wpack; t.bdesc = h->desc1.bias; t.sdesc = h->desc1.scale; t.tdesc = h->desc1.shift;
        t.D = D; t.npx = npx; t.B = B; t.Hc = Hc; t.Wc = Wc;
        t.prob = prob; t.logits_nchw = logits; t.desc = desc;
        t.softmax_mode = h->cfg.softmax_mode; t.normalize = h->cfg.normalize_descriptors ? 1 : 0;
        prof_begin(h, "heads.tail", 2.0 * hc * (65.0 + (desc ? D : 0)) * (double)npx, s);
        const int miss = launch_head_tail(t, s);
        prof_end(h, s);
        if (!miss) {
            MP_HIP(hipGetLastError());
            return MP_OK;
        }
        if (h->prof) --h->prof_used;    // not covered: fall through to the separate kernels
        if (!h->head_fallback_noted) {  // ... visibly: once per handle on stderr, and the profile then lists "det.conv1x1" etc.
            h->head_fallback_noted = true;
            fprintf(stderr, "[multipoint_hip] note: fused head tail not instantiated for %d head channels / descriptor size %d: "
                            "using the separate 1x1 convolution, softmax and normalisation launches\n", hc, D);
        }
    }
    if ((rc = run_conv(h, h->det1, P, headc, 0, Lg, 80, 0, B, Hc, Wc, nullptr, s))) return rc;
    if (prob || logits) {
        prof_begin(h, "det.softmax_shuffle", 0.0, s);
        launch_det_post(Lg, 80, B, Hc, Wc, prob, logits, h->cfg.softmax_mode, s);
        prof_end(h, s);
    }
    if (desc) {
        if ((rc = run_conv(h, h->desc1, P, headc, hc, desc, D, 0, B, Hc, Wc, nullptr, s))) return rc;
        prof_begin(h, "desc.l2norm", 0.0, s);
        if (h->cfg.normalize_descriptors) launch_desc_l2norm(desc, desc, npx, D, 1, s);
        prof_end(h, s);
    }
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_forward(mp_handle* h, const float* images, const unsigned char* is_optical, int B, int H, int W,
               float* prob, float* logits, float* desc, void* stream)
{
    if (!h) return MP_EINVAL;
    return forward_checked(h, images, is_optical, B, H, W, prob, logits, desc, stream);
}

int mp_box_nms(mp_handle* h, const float* prob, const unsigned char* valid_mask, int B, int H, int W,
               float size, float min_prob, double iou, int keep_top_k, float* prob_nms, int max_rounds,
               void* stream)
{
    if (!h) return MP_EINVAL;
    if (!prob || !prob_nms) return fail(h, MP_EINVAL, "mp_box_nms: NULL tensor");
    MP_HIP(hipSetDevice(h->device));
    return nms_common(h, prob, valid_mask, B, H, W, size, min_prob, iou, keep_top_k, 0, nullptr, nullptr,
                      nullptr, prob_nms, max_rounds, static_cast<hipStream_t>(stream));
}

int mp_detect_keypoints(mp_handle* h, const float* prob, const unsigned char* valid_mask, int B, int H,
                        int W, float size, float min_prob, double iou, int keep_top_k, int K, int* kp_yx,
                        float* kp_score, int* kp_count, int max_rounds, void* stream)
{
    if (!h) return MP_EINVAL;
    if (!prob || !kp_yx || !kp_count || K <= 0) return fail(h, MP_EINVAL, "mp_detect_keypoints: bad argument");
    MP_HIP(hipSetDevice(h->device));
    return nms_common(h, prob, valid_mask, B, H, W, size, min_prob, iou, keep_top_k, K, kp_yx, kp_score,
                      kp_count, nullptr, max_rounds, static_cast<hipStream_t>(stream));
}

int mp_nms_unresolved(mp_handle* h, int* unresolved, void* stream)
{
    if (!h || !unresolved) return MP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    *unresolved = 0;
    if (!h->nms_total) return MP_OK;
    MP_HIP(hipMemcpyAsync(h->pinned, h->nms_total, 4, hipMemcpyDeviceToHost, s));
    MP_HIP(hipMemsetAsync(h->nms_total, 0, 4, s));
    MP_HIP(hipStreamSynchronize(s));
    *unresolved = h->pinned[0];
    return MP_OK;
}

int mp_topk_tie_guard(mp_handle* h, float eps, int min_each_side)
{
    if (!h) return MP_EINVAL;
    if (!(eps >= 0.f) || min_each_side < 0) return fail(h, MP_EINVAL, "mp_topk_tie_guard: eps >= 0 and min_each_side >= 0 (0: off)");
    h->tie_eps = eps; h->tie_min = min_each_side;
    return MP_OK;
}

int mp_nms_tie_guard(mp_handle* h, int min_pairs)
{
    if (!h) return MP_EINVAL;
    if (min_pairs < 0) return fail(h, MP_EINVAL, "mp_nms_tie_guard: min_pairs >= 0 (0: off)");
    h->tie_pairs_min = min_pairs;
    return MP_OK;
}

int mp_topk_ambiguous(mp_handle* h, int* flags, int B, int* total, void* stream)
{
    if (!h || !total || B < 0 || (B > 0 && !flags)) return MP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    *total = 0;
    for (int b = 0; b < B; ++b) flags[b] = 0;
    if (!h->tie_state) return MP_OK;
    MP_HIP(hipSetDevice(h->device));
    const int nb = B < h->tie_last_B ? B : h->tie_last_B;       // flags exist for the images of the latest call only
    MP_HIP(hipMemcpyAsync(h->pinned, h->tie_state, (size_t)(1 + nb) * 4, hipMemcpyDeviceToHost, s));
    MP_HIP(hipMemsetAsync(h->tie_state, 0, 4, s));
    MP_HIP(hipStreamSynchronize(s));
    *total = h->pinned[0];
    for (int b = 0; b < nb; ++b) flags[b] = h->pinned[1 + b];
    return MP_OK;
}

int mp_extract_keypoints(mp_handle* h, const float* map, const unsigned char* valid_mask, int B, int H, int W, float thr,
                         int K, int* kp_yx, float* kp_score, int* kp_count, void* stream)
{
    if (!h) return MP_EINVAL;
    if (!map || !kp_yx || !kp_count || K <= 0 || B <= 0 || H <= 0 || W <= 0)
        return fail(h, MP_EINVAL, "mp_extract_keypoints: bad argument");
    MP_HIP(hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->kp_scratch, keypoint_scratch_ints(B, H, W) * 4))) return rc;
    launch_extract_threshold(map, valid_mask, B, H, W, thr, K, kp_yx, kp_score, kp_count, static_cast<int*>(h->kp_scratch.p),
                             static_cast<hipStream_t>(stream));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_sample_descriptors(mp_handle* h, const float* desc, int B, int Hc, int Wc, int D, int H, int W,
                          const int* kp_yx, const int* kp_count, int K, float* out, void* stream)
{
    if (!h) return MP_EINVAL;
    if (!desc || !kp_yx || !kp_count || !out) return fail(h, MP_EINVAL, "mp_sample_descriptors: NULL tensor");
    if (D % 64 != 0 || D > 256 || D <= 0)
        return fail(h, MP_EINVAL, "mp_sample_descriptors: D must be 64, 128, 192 or 256");
    MP_HIP(hipSetDevice(h->device));
    launch_sample_desc(desc, B, Hc, Wc, D, H, W, kp_yx, kp_count, K, out, static_cast<hipStream_t>(stream));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_match_mutual_nn(mp_handle* h, const float* descA, const int* countA, const float* descB,
                       const int* countB, long long pair_stride, int count_stride, int P, int K, int D,
                       float threshold, int* match_idx, float* match_dist, int* match_count, void* stream)
{
    if (!h) return MP_EINVAL;
    if (!descA || !descB || !countA || !countB || !match_idx || !match_dist || !match_count)
        return fail(h, MP_EINVAL, "mp_match_mutual_nn: NULL tensor");
    if (D != 64 && D != 128 && D != 256) return fail(h, MP_EINVAL, "mp_match_mutual_nn: D must be 64, 128 or 256");
    if (P <= 0 || K <= 0) return fail(h, MP_EINVAL, "mp_match_mutual_nn: P and K must be positive");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MP_HIP(hipSetDevice(h->device));
    const size_t need = (size_t)P * K * 8 * 2 * MATCH_SHARES;      // packed (distance bits, index) arg-min arrays, one per column share
    int rc;
    if ((rc = ensure(h, h->ws3, need))) return rc;
    unsigned long long* rowbest = static_cast<unsigned long long*>(h->ws3.p);
    unsigned long long* colbest = rowbest + (size_t)P * K * MATCH_SHARES;
    launch_match_impl(descA, countA, descB, countB, pair_stride, count_stride, P, K, D, threshold, rowbest,
                      colbest, match_idx, match_dist, match_count, s);
    MP_HIP(hipGetLastError());
    return MP_OK;
}

static int match_extra_check(mp_handle* h, const char* fn, const void* a, const void* b, const void* c, const void* d,
                             int P, int K, int D)
{
    if (!a || !b || !c || !d) return fail(h, MP_EINVAL, std::string(fn) + ": NULL tensor");
    if (P <= 0 || P > 65535 || K <= 0) return fail(h, MP_EINVAL, std::string(fn) + ": need 0 < P <= 65535, K > 0");
    if (D <= 0 || D > 256) return fail(h, MP_EINVAL, std::string(fn) + ": D must be in [1, 256]");
    return MP_OK;
}

int mp_match_knn2(mp_handle* h, const float* descA, const int* countA, const float* descB, const int* countB,
                  long long pair_stride, int count_stride, int P, int K, int D, int* nn_idx, float* nn_dist,
                  void* stream)
{
    if (!h) return MP_EINVAL;
    int rc;
    if ((rc = match_extra_check(h, "mp_match_knn2", descA, descB, countA, countB, P, K, D))) return rc;
    if (!nn_idx || !nn_dist) return fail(h, MP_EINVAL, "mp_match_knn2: NULL output");
    MP_HIP(hipSetDevice(h->device));
    launch_match_knn2(descA, countA, descB, countB, pair_stride, count_stride, P, K, D, nn_idx, nn_dist,
                      static_cast<hipStream_t>(stream));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_match_threshold(mp_handle* h, const float* descA, const int* countA, const float* descB, const int* countB,
                       long long pair_stride, int count_stride, int P, int K, int D, float threshold, int capacity,
                       int* list_ij, float* list_dist, int* list_count, void* stream)
{
    if (!h) return MP_EINVAL;
    int rc;
    if ((rc = match_extra_check(h, "mp_match_threshold", descA, descB, countA, countB, P, K, D))) return rc;
    if (!list_ij || !list_dist || !list_count) return fail(h, MP_EINVAL, "mp_match_threshold: NULL output");
    if (capacity <= 0) return fail(h, MP_EINVAL, "mp_match_threshold: capacity must be positive");
    if (!(threshold >= 0.f)) return fail(h, MP_EINVAL, "mp_match_threshold: threshold must be non-negative");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MP_HIP(hipSetDevice(h->device));
    MP_HIP(hipMemsetAsync(list_count, 0, (size_t)P * sizeof(int), s));
    launch_match_threshold(descA, countA, descB, countB, pair_stride, count_stride, P, K, D, threshold, capacity, list_ij,
                           list_dist, list_count, s);
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_pair_metrics(mp_handle* h, const int* kp_yx, const int* kp_count, const int* match_idx, const double* homography,
                    int P, int K, int H, int W, float threshold_keypoints, int* metrics, unsigned char* tp,
                    void* stream)
{
    if (!h) return MP_EINVAL;
    if (!kp_yx || !kp_count || !match_idx || !homography || !metrics || !tp)
        return fail(h, MP_EINVAL, "mp_pair_metrics: NULL tensor");
    if (P <= 0 || K <= 0 || H <= 0 || W <= 0) return fail(h, MP_EINVAL, "mp_pair_metrics: P, K, H, W must be positive");
    if (!(threshold_keypoints >= 0.f)) return fail(h, MP_EINVAL, "mp_pair_metrics: threshold must be non-negative");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MP_HIP(hipSetDevice(h->device));
    // scratch: warped [2P][K][2] double | inv_idx [P][K] int
    const size_t nw = (size_t)2 * P * K * 2 * sizeof(double), ni = (size_t)P * K * sizeof(int);
    int rc;
    if ((rc = ensure(h, h->ws4, nw + ni))) return rc;
    double* warped = static_cast<double*>(h->ws4.p);
    int* inv_idx = reinterpret_cast<int*>(static_cast<char*>(h->ws4.p) + nw);
    MP_HIP(hipMemsetAsync(inv_idx, 0xff, ni, s));
    MP_HIP(hipMemsetAsync(tp, 0, (size_t)2 * P * K, s));
    MP_HIP(hipMemsetAsync(metrics, 0, (size_t)P * 8 * sizeof(int), s));
    launch_pair_metrics(kp_yx, kp_count, match_idx, homography, P, K, H, W, threshold_keypoints, warped, inv_idx, tp,
                        metrics, s);
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_repeatability(mp_handle* h, const int* kp_yx, const int* kp_count, const double* homography, int P, int K, int H,
                     int W, double distance_thresh, int* counts, void* stream)
{
    if (!h) return MP_EINVAL;
    if (!kp_yx || !kp_count || !homography || !counts) return fail(h, MP_EINVAL, "mp_repeatability: NULL tensor");
    if (P <= 0 || K <= 0 || H <= 0 || W <= 0) return fail(h, MP_EINVAL, "mp_repeatability: P, K, H, W must be positive");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MP_HIP(hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->ws4, (size_t)2 * P * K * 2 * sizeof(long long)))) return rc;
    MP_HIP(hipMemsetAsync(counts, 0, (size_t)P * 4 * sizeof(int), s));
    launch_repeatability(kp_yx, kp_count, homography, P, K, H, W, distance_thresh, static_cast<long long*>(h->ws4.p),
                         counts, s);
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_find_homography(mp_handle* h, const int* kp_yx, const int* kp_count, const int* match_idx, int P, int K,
                       double reproj_threshold, int max_iters, unsigned long long seed, double* homography,
                       unsigned char* inlier_mask, int* n_inliers, void* stream)
{
    if (!h) return MP_EINVAL;
    if (!kp_yx || !kp_count || !match_idx || !homography || !inlier_mask || !n_inliers)
        return fail(h, MP_EINVAL, "mp_find_homography: NULL tensor");
    if (P <= 0 || K <= 0 || K > 3200) return fail(h, MP_EINVAL, "mp_find_homography: need P > 0 and 0 < K <= 3200");
    if (max_iters <= 0 || max_iters > (1 << 20)) return fail(h, MP_EINVAL, "mp_find_homography: max_iters out of range");
    if (!(reproj_threshold > 0.0)) return fail(h, MP_EINVAL, "mp_find_homography: threshold must be positive");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MP_HIP(hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->ws4, (size_t)P * sizeof(unsigned long long)))) return rc;
    unsigned long long* best = static_cast<unsigned long long*>(h->ws4.p);
    MP_HIP(hipMemsetAsync(best, 0, (size_t)P * sizeof(unsigned long long), s));
    MP_HIP(hipMemsetAsync(inlier_mask, 0, (size_t)P * K, s));
    launch_ransac_homography(kp_yx, kp_count, match_idx, P, K, max_iters, reproj_threshold, seed, best, homography,
                             inlier_mask, n_inliers, s);
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_detector_metrics(mp_handle* h, const float* prob, const unsigned char* keypoint_map, int B, int H, int W,
                        float zero_threshold, float distance_thresh, unsigned long long* work, int* rec_index,
                        float* rec_prob, unsigned int* rec_bits, int* rec_count, int* n_gt, void* stream)
{
    if (!h) return MP_EINVAL;
    if (!prob || !keypoint_map || !work || !rec_index || !rec_prob || !rec_bits || !rec_count || !n_gt)
        return fail(h, MP_EINVAL, "mp_detector_metrics: NULL tensor");
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL)
        return fail(h, MP_EINVAL, "mp_detector_metrics: need 0 < B <= 65535, H, W > 0");
    if (!(distance_thresh >= 0.f) || !(distance_thresh < 3.f))
        return fail(h, MP_EINVAL, "mp_detector_metrics: distance_thresh must be in [0, 3) (5 x 5 window)");
    if (!(zero_threshold >= 0.f)) return fail(h, MP_EINVAL, "mp_detector_metrics: zero_threshold must be >= 0");
    MP_HIP(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    MP_HIP(hipMemsetAsync(work, 0, sizeof(unsigned long long) * (size_t)B * H * W, s));
    MP_HIP(hipMemsetAsync(rec_count, 0, sizeof(int) * (size_t)B, s));
    MP_HIP(hipMemsetAsync(n_gt, 0, sizeof(int) * (size_t)B, s));
    launch_detector_metrics(prob, keypoint_map, B, H, W, zero_threshold, distance_thresh, work, rec_index, rec_prob,
                            rec_bits, rec_count, n_gt, s);
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_warp_perspective(mp_handle* h, const float* src, int n_src, int H, int W, const double* dst_to_src, int n_out,
                        int Ho, int Wo, int mode, int padding, float* dst, void* stream)
{
    if (!h) return MP_EINVAL;
    if (!src || !dst_to_src || !dst) return fail(h, MP_EINVAL, "mp_warp_perspective: NULL tensor");
    if (n_src <= 0 || n_out <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || n_out > 65535)
        return fail(h, MP_EINVAL, "mp_warp_perspective: sizes must be positive (n_out <= 65535)");
    if ((mode != 0 && mode != 1) || (padding != 0 && padding != 1))
        return fail(h, MP_EINVAL, "mp_warp_perspective: mode must be 0 (bilinear) / 1 (nearest), padding 0 (zeros) / 1 (reflection)");
    MP_HIP(hipSetDevice(h->device));
    launch_warp_perspective(src, n_src, H, W, dst_to_src, n_out, Ho, Wo, mode, padding, dst, static_cast<hipStream_t>(stream));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_warp_perspective_cv(mp_handle* h, const float* src, int n, int H, int W, const double* hom_inv, int border,
                           float* dst, void* stream)
{
    if (!h) return MP_EINVAL;
    if (!src || !hom_inv || !dst) return fail(h, MP_EINVAL, "mp_warp_perspective_cv: NULL tensor");
    if (n <= 0 || n > 65535 || H <= 0 || W <= 0 || H > 32767 || W > 32767)
        return fail(h, MP_EINVAL, "mp_warp_perspective_cv: need 0 < n <= 65535 and 0 < H, W <= 32767");
    if (border != 0 && border != 1)
        return fail(h, MP_EINVAL, "mp_warp_perspective_cv: border must be 0 (BORDER_CONSTANT 0) or 1 (BORDER_REFLECT_101)");
    if (src == dst) return fail(h, MP_EINVAL, "mp_warp_perspective_cv: in-place warp is not supported");
    MP_HIP(hipSetDevice(h->device));
    launch_cv_warp_linear(src, n, H, W, hom_inv, border, dst, static_cast<hipStream_t>(stream));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_ha_valid_mask(mp_handle* h, const double* hom_inv, int G, int H, int W, int erosion_radius, int mask_border,
                     unsigned char* mask, void* stream)
{
    if (!h) return MP_EINVAL;
    if (!hom_inv || !mask) return fail(h, MP_EINVAL, "mp_ha_valid_mask: NULL tensor");
    if (G <= 0 || G > 65535 || H <= 0 || W <= 0) return fail(h, MP_EINVAL, "mp_ha_valid_mask: need 0 < G <= 65535, H, W > 0");
    if (erosion_radius < 0 || erosion_radius > 16) return fail(h, MP_EINVAL, "mp_ha_valid_mask: erosion_radius must be in [0, 16]");
    MP_HIP(hipSetDevice(h->device));
    launch_ha_valid_mask(hom_inv, G, H, W, erosion_radius, mask_border != 0, mask, static_cast<hipStream_t>(stream));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

static int ha_check(mp_handle* h, const char* fn, const float* pa, const float* pb, int B, int H, int W, int aggregation)
{
    if (!pa || (aggregation != 0 && !pb)) return fail(h, MP_EINVAL, std::string(fn) + ": NULL heat map");
    if (aggregation < 0 || aggregation > 2) return fail(h, MP_EINVAL, std::string(fn) + ": aggregation must be 0 (single), 1 (prod) or 2 (sum)");
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0) return fail(h, MP_EINVAL, std::string(fn) + ": need 0 < B <= 65535, H, W > 0");
    return MP_OK;
}

int mp_ha_begin(mp_handle* h, const float* prob_a, const float* prob_b, int B, int H, int W, int aggregation,
                float* prob, float* count, void* stream)
{
    if (!h) return MP_EINVAL;
    int rc;
    if ((rc = ha_check(h, "mp_ha_begin", prob_a, prob_b, B, H, W, aggregation))) return rc;
    if (!prob || !count) return fail(h, MP_EINVAL, "mp_ha_begin: NULL tensor");
    MP_HIP(hipSetDevice(h->device));
    launch_ha_begin(prob_a, prob_b, (long long)B * H * W, aggregation, prob, count, static_cast<hipStream_t>(stream));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_ha_accumulate(mp_handle* h, const float* prob_a, const float* prob_b, const unsigned char* mask,
                     const double* hom, int G, int B, int H, int W, int aggregation, float* prob, float* count,
                     void* stream)
{
    if (!h) return MP_EINVAL;
    int rc;
    if ((rc = ha_check(h, "mp_ha_accumulate", prob_a, prob_b, B, H, W, aggregation))) return rc;
    if (!mask || !hom || !prob || !count) return fail(h, MP_EINVAL, "mp_ha_accumulate: NULL tensor");
    if (G <= 0) return fail(h, MP_EINVAL, "mp_ha_accumulate: G must be positive");
    MP_HIP(hipSetDevice(h->device));
    launch_ha_accumulate(prob_a, prob_b, mask, hom, G, B, H, W, aggregation, prob, count, static_cast<hipStream_t>(stream));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_ha_finalize(mp_handle* h, const float* prob, const float* count, int B, int H, int W, int aggregation,
                   float min_count, float* out, void* stream)
{
    if (!h) return MP_EINVAL;
    int rc;
    if ((rc = ha_check(h, "mp_ha_finalize", prob, count, B, H, W, aggregation))) return rc;
    if (!out) return fail(h, MP_EINVAL, "mp_ha_finalize: NULL tensor");
    MP_HIP(hipSetDevice(h->device));
    launch_ha_finalize(prob, count, (long long)B * H * W, aggregation, min_count, out, static_cast<hipStream_t>(stream));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_gaussian_filter(mp_handle* h, const float* in, int B, int H, int W, int ksize, const float* weights, float* out,
                       void* stream)
{
    if (!h) return MP_EINVAL;
    if (!in || !weights || !out) return fail(h, MP_EINVAL, "mp_gaussian_filter: NULL tensor");
    if (in == out) return fail(h, MP_EINVAL, "mp_gaussian_filter: in-place filtering is not supported");
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0) return fail(h, MP_EINVAL, "mp_gaussian_filter: need 0 < B <= 65535, H, W > 0");
    if (ksize < 1 || ksize > 31 || (ksize & 1) == 0) return fail(h, MP_EINVAL, "mp_gaussian_filter: ksize must be odd and <= 31");
    if ((ksize - 1) / 2 >= H || (ksize - 1) / 2 >= W) return fail(h, MP_EINVAL, "mp_gaussian_filter: reflection padding needs (ksize-1)/2 < H, W");
    MP_HIP(hipSetDevice(h->device));
    launch_gaussian_filter(in, B, H, W, ksize, weights, out, static_cast<hipStream_t>(stream));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_profile_enable(mp_handle* h, int enable)
{
    if (!h) return MP_EINVAL;
    h->prof = enable != 0;
    h->prof_used = 0;
    return MP_OK;
}

int mp_profile_read(mp_handle* h, const char** names, float* ms, double* flop, int capacity, int* n)
{
    if (!h || !n) return MP_EINVAL;
    *n = 0;
    for (size_t i = 0; i < h->prof_used && (int)i < capacity; ++i) {
        ProfEntry& e = h->prof_entries[i];
        MP_HIP(hipEventSynchronize(e.b));
        float t = 0.f;
        MP_HIP(hipEventElapsedTime(&t, e.a, e.b));
        if (names) names[i] = e.name;
        if (ms) ms[i] = t;
        if (flop) flop[i] = e.flop;
        ++*n;
    }
    h->prof_used = 0;
    return MP_OK;
}

}  // extern "C"

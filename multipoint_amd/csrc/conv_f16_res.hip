// fp16 3x3 convolution with LDS-RESIDENT weights: the 64 -> 64 layers of the mixed_precision path (enc.conv2 / conv3 / conv4:
// 54 % of BASELINE configs[4]'s step), reference MultiPoint.py:99-103,143-148 under autocast -- same arithmetic and rounding
// points as conv_f16.hip (fp32 accumulate + fp16 bias -> fp16; ReLU; BatchNorm as an fp32 affine -> fp16; max-pool).
//
// Why a second kernel.  conv_f16.hip streams every wave's own copy of the weight fragments through the vector L1: 2 KiB per
// wave and step of 4 MFMAs (128 cycles) = 64 B/clk/CU with four SIMDs busy -- the L1's whole bandwidth -- and its matrix pipe
// sits at 35-46 % (round-2 verdict, weak 2).  A 64 -> 64 layer's weights are 9 x 64 x 64 fp16 = 72 KiB: they fit the 160 KiB LDS
// ONCE per CU.  So here ONE workgroup of 512 threads owns a CU, copies the packed weights into LDS once, and both operands of
// every MFMA come from LDS (ds_read_b128: 256 B/clk/CU on gfx950; the 4 KiB a wave reads per step are 50 % of that at full
// matrix rate).  The eight waves form TWO independent groups of four that behave like conv_f16.hip's two co-resident
// workgroups -- each walks its own work items, with its own 256-pixel activation tile in LDS (chunks of 32 input channels,
// 27 KiB) -- so that one group's epilogue and tile hand-over run under the other group's MFMAs.  A workgroup barrier would
// lock the groups in phase, so a group synchronises with a counter in LDS (one ds_add per wave + polling reads; LDS
// operations of a CU execute in order, so a wave's tile writes are visible before its increment).  Group 1 starts half an
// item late.
#include "mp_common.h"

#include <algorithm>
#include <type_traits>

#ifdef MP_TIMING
// developer instrumentation (tools/conv_timing_f16_res.py): per-workgroup cycle sums per phase of the fused conv1+conv2 launch,
// wave 0 of groups 0 and 1
__device__ unsigned long long g_timing_r[512 * 8];
extern "C" int mp_debug_read_timing_f16_res(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timing_r), sizeof(unsigned long long) * n);
}
#define MPR_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define MPR_ADD(slot, a, b) do { tsum[slot] += (b) - (a); } while (0)
#else
#define MPR_T(var) do { } while (0)
#define MPR_ADD(slot, a, b) do { } while (0)
#endif
#ifndef MPRX
#define MPRX 0     // developer elimination switches (timing only, results WRONG): 1 no dependence on the staging loads, 2 no epilogue,
#endif             // 4 no first-block production (F1), 8 no group barriers, 16 one MFMA step per chunk instead of 18, 32 un-pooled stores lane-linear
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CKR = 32;        // input channels per LDS chunk
constexpr int PSR = CKR + 8;   // LDS pixel stride in halfs: 80 B = 20 dwords -> 16 consecutive pixels start in 16 distinct 16-byte bank groups
constexpr int WRES = 36 * 2 * 64 * 8;      // halfs of the resident weights: [step = tap*4 + kgroup][nblock][lane][8] = 72 KiB

template <int MBW>
struct GeoR {
    static constexpr int MBH = 32 / MBW;
    static constexpr int TW = MBW;
    static constexpr int TH = 256 / MBW;
    static constexpr int LW = TW + 2;
    static constexpr int LH = TH + 2;
    static constexpr int NPIX = LW * LH;
    static constexpr int NV = NPIX * (CKR / 8);          // 16-byte vectors per chunk tile
    static constexpr int NITER = (NV + 255) / 256;       // staging vectors per thread of a group
    static constexpr int STEPS = 9 * (CKR / 16);         // k16-steps per chunk
};

__device__ __forceinline__ int reflect_clamp_r(int v, int n)
{
    v = v < 0 ? -v : v;
    v = v >= n ? 2 * (n - 1) - v : v;
    v = v < 0 ? 0 : v;
    return v >= n ? n - 1 : v;
}

// conv result pair (fp32 accumulators) -> activation as autocast produces it; identical to conv_f16.hip's act_h2
template <bool BNF>
__device__ __forceinline__ h2 act_r2(float a0, float a1, f32x2 bias, f32x2 scale, f32x2 shift)
{
    const f32x2 x = f32x2{a0, a1} + bias;
    h2 h = __builtin_convertvector(x, h2);
    const h2 zero = {0, 0};
    if (!BNF) h = __builtin_elementwise_max(h, zero);
    f32x2 y = __builtin_convertvector(h, f32x2) * scale + shift;
    asm volatile("" : "+v"(y));      // y exists as an fp32 pair (autocast: BatchNorm result in fp32, THEN fp16): no v_fma_mixlo_f16, which rounds once
    h2 o = __builtin_convertvector(y, h2);
    if (BNF) o = __builtin_elementwise_max(o, zero);
    return o;
}

// barrier of the four waves of a group: a counter in LDS.  The LDS executes the operations of a CU in the order they are
// issued, and a wave issues its own in program order: whatever a wave read from or wrote to LDS before its increment has been
// performed by the time another wave sees that increment.  No workgroup-scope fence: it would also drain vmcnt, i.e. wait for
// the staging loads and the epilogue's stores in flight.
typedef __attribute__((address_space(3))) unsigned lds_u32;
__device__ __forceinline__ void group_barrier(lds_u32* ctr, unsigned& target, int lane)
{
    if (MPRX & 8) return;
    target += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
        const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (v >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// F1: the layer's input is the first encoder block (Cin = 1 -> 64: Conv2d + ReLU + BatchNorm of p.img with p.w1 / b1 / s1 / t1,
// conv_f16.hip's conv_first_f16_kernel) and is never materialised in HBM: a group builds each 32-channel chunk of its activation
// tile straight into LDS, on the matrix pipe -- D[channel][pixel] = W1[channel][tap] X[tap][pixel], K = 9 taps padded to 16, one
// MFMA per 32 pixels and chunk; a lane gathers its pixel's taps from a (TH+4) x (TW+4) fp16 image patch in LDS whose rows and
// columns are staged already reflected (the block's own ReflectionPad2d), at a window origin reflected by THIS layer's padding.
// The launch then reads 2.6 MB of image per 16 frames instead of a 2.7 GB tensor that a separate HBM-write-bound launch had
// to produce.  Reflection padding only (zero-padding models keep the two launches).
template <int MBW, bool POOL, bool BNF, int NG, bool F1>
__global__ __launch_bounds__(256 * NG, NG) void conv_f16_res_kernel(const ConvParamsH p)
{
    using G = GeoR<MBW>;
    constexpr bool SWAP = !POOL;      // weights as the MFMA A operand -> lane = pixel, register quad = 4 channels
    __shared__ __attribute__((aligned(16))) _Float16 wl[WRES];
    __shared__ __attribute__((aligned(16))) _Float16 tiles[NG][G::NPIX * PSR];
    __shared__ __attribute__((aligned(16))) float prm[3 * 64];
    __shared__ unsigned gctr[NG];
    // un-pooled layers: a wave's output block (32 pixels x 64 channels, 4 KiB) passes through LDS so that a lane stores 16 bytes and eight
    // lanes a pixel's whole 128-byte line (lane = pixel in the accumulators: 8-byte pieces 128 bytes apart, 32 partial lines per store
    // instruction -- measured 20 % of enc.conv3's launch); 16-byte granules XOR-swizzled by the pixel's low bits instead of a padded stride.
    // The staging block lives in the group's own TILE: behind the last chunk's barrier nobody reads the tile any more, and the next item's
    // chunk 0 waits in the staging registers until the epilogue is through (lds_write() behind it).  A wave stages in the stripes of the
    // tile that its OWN threads write in lds_write() -- pixels 16 w + 64 j .. + 15 (1280 bytes each), eight 128-byte rows per stripe --
    // so no other wave's tile write can land in a block that is still being read, and no barrier separates epilogue and tile write.
    static_assert(G::NITER >= 5 && 16 * PSR >= 8 * 64, "a wave owns four full stripes of 16 tile pixels");
    // F1: image patches (two per group: the current item's and the next one's) + a zero tail that the padding taps read
    constexpr int IW = G::LW + 2, IH = G::LH + 2, NIP = IW * IH;
    constexpr int NIPB = ((NIP + 2 * IW + 3 + 7) / 8) * 8;          // halfs per patch buffer incl. the zero tail
    constexpr int NIPR = (NIP + 255) / 256;                          // patch pixels per thread
    __shared__ __attribute__((aligned(16))) _Float16 ipatch[F1 ? NG * 2 * NIPB : 8];
    __shared__ __attribute__((aligned(16))) float prm1[F1 ? 3 * 64 : 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int gt = tid & 255;                                     // thread of the group
    const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);   // wave of the group
    const int half = lane >> 5;
    const int li = lane & 31;
    _Float16* const lds = tiles[grp];

    // ---- once per workgroup: the layer's packed weights and epilogue parameters into LDS ----
    {
        const h8* const src = reinterpret_cast<const h8*>(p.wpack);
        for (int f = tid; f < WRES / 8; f += 256 * NG) reinterpret_cast<h8*>(wl)[f] = src[f];
        if (tid < 64) { prm[tid] = p.bias[tid]; prm[64 + tid] = p.scale[tid]; prm[128 + tid] = p.shift[tid]; }
        if (tid < NG) gctr[tid] = 0u;
        if constexpr (F1) {
            // the tails stay as written here: zeros, and a 1.0 where the lanes that hold taps 8..15 read "tap 9" (the bias)
            for (int f = tid; f < NG * 2 * NIPB; f += 256 * NG) ipatch[f] = (f % NIPB == NIP + 1) ? (_Float16)1.f : (_Float16)0.f;
            if (tid < 64) { prm1[64 + tid] = p.s1[tid]; prm1[128 + tid] = p.t1[tid]; }
        }
    }
    __syncthreads();                                              // the only workgroup barrier
    unsigned bar_target = 0u;
    lds_u32* const ctr = (lds_u32*)&gctr[grp];

    // ---- work items of this group: the workgroup's XCD share, two virtual workgroups per real one ----
    const int nxcd = 1 << p.xcd_shift;
    const int per_xcd = (p.nitems + nxcd - 1) >> p.xcd_shift;
    const int xcd = (int)blockIdx.x & (nxcd - 1);
    const int stride = NG * ((int)gridDim.x >> p.xcd_shift);
    const int item_end = min((xcd + 1) * per_xcd, p.nitems);
    int item = xcd * per_xcd + NG * ((int)blockIdx.x >> p.xcd_shift) + grp;
    if (item >= item_end) return;
    // group g starts g / NG of an item late (~13 k cycles per item): the groups then take turns between MFMA steps and
    // epilogue / tile hand-over
    for (int g = 0; g < grp; ++g) __builtin_amdgcn_s_sleep(NG == 2 ? 127 : 70);

    auto udiv = [](unsigned n, unsigned magic, unsigned d) -> unsigned { return d == 1 ? n : __umulhi(n, magic); };
    struct Where { int img, y0, x0; const _Float16* in_base; };
    auto decode = [&](int tile) __attribute__((always_inline)) -> Where {
        Where w{};
        const int trow = (int)udiv((unsigned)tile, p.magic_tx, (unsigned)p.tiles_x);
        const int tx = tile - trow * p.tiles_x;
        const int bi = (int)udiv((unsigned)trow, p.magic_ty, (unsigned)p.tiles_y);
        const int ty = trow - bi * p.tiles_y;
        w.img = p.img_list ? p.img_list[bi] : bi;
        w.y0 = ty * G::TH; w.x0 = tx * G::TW;
        w.in_base = p.in + (long long)w.img * p.H * p.W * p.in_cstride + p.in_coff;
        return w;
    };
    // per-thread staging offsets in halfs (channel granule c8 of patch pixel lp); interior items share one item-invariant set
    int goff[G::NITER];
    bool goff_rel = false;
    bool cur_pad = false;
    auto offsets = [&](const Where& w) __attribute__((always_inline)) -> const _Float16* {
        const bool interior = (w.y0 >= 1) && (w.y0 + G::TH < p.H) && (w.x0 >= 1) && (w.x0 + G::TW < p.W);
        if (interior) {
            if (!goff_rel) {
#pragma unroll
                for (int j = 0; j < G::NITER; ++j) {
                    const int f = gt + j * 256;
                    const int lp = f >> 2;
                    int off = 0;
                    if (f < G::NV) {
                        const int ly = lp / G::LW, lx = lp - ly * G::LW;
                        off = (ly * p.W + lx) * p.in_cstride + (f & 3) * 8;
                    }
                    goff[j] = off;
                }
                goff_rel = true;
            }
            return w.in_base + (long long)((w.y0 - 1) * p.W + (w.x0 - 1)) * p.in_cstride;
        }
        goff_rel = false;
#pragma unroll
        for (int j = 0; j < G::NITER; ++j) {
            const int f = gt + j * 256;
            const int lp = f >> 2, c8 = f & 3;
            int off = -1;
            if (f < G::NV) {
                const int ly = lp / G::LW, lx = lp - ly * G::LW;
                int gy = w.y0 + ly - 1, gx = w.x0 + lx - 1;
                bool zero = false;
                if (p.pad_zero) {
                    zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
                    gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
                } else {
                    gy = reflect_clamp_r(gy, p.H); gx = reflect_clamp_r(gx, p.W);
                }
                if (!zero) off = (gy * p.W + gx) * p.in_cstride + c8 * 8;
            }
            goff[j] = off;
        }
        return w.in_base;
    };


    // ---------------- F1: first encoder block evaluated into the tile ----------------
    _Float16* const ipg = ipatch + (F1 ? grp * 2 * NIPB : 0);     // this group's two patch buffers
    float ipx[NIPR];                                              // the NEXT item's patch pixels, loaded one item ahead
    h8 w1f[2];                                                    // A operands: W1[channel 32c + li][tap 8 half + e], 0 beyond tap 8
    constexpr int NMB1 = (G::NPIX + 31) / 32;                     // M-blocks of 32 tile pixels
    constexpr int NJ = (NMB1 + 3) / 4;                            // ... per wave
    int va1[NJ], vb1[NJ], wa1[NJ];                                // gather bases (halfs into a patch buffer), tile write offset (halfs; -1: phantom pixels)
    bool f1_rel = false;
    auto patch_load = [&](const Where& w) __attribute__((always_inline)) {
        const float* const im = p.img + (long long)w.img * p.H * p.W;
#pragma unroll
        for (int i = 0; i < NIPR; ++i) {
            const int f = min(gt + i * 256, NIP - 1);
            const int r = f / IW, c = f - r * IW;
            ipx[i] = im[reflect_clamp_r(w.y0 - 2 + r, p.H) * p.W + reflect_clamp_r(w.x0 - 2 + c, p.W)];
        }
    };
    auto patch_write = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIPR; ++i) {
            const int f = gt + i * 256;
            if (f < NIP) ipg[par * NIPB + f] = (_Float16)ipx[i];         // autocast rounds the convolution's input to fp16
        }
    };
    // window origins of this lane's pixels for item w: staged row r holds image row reflect(y0 - 2 + r), so the 3x3 window of the
    // tile pixel at frame position (gy, gx) -- reflected into the frame by this layer's padding -- starts at staged (gy' - y0 + 1,
    // gx' - x0 + 1); interior items share one item-invariant set
    auto f1_offsets = [&](const Where& w) __attribute__((always_inline)) {
        const bool interior = (w.y0 >= 1) && (w.y0 + G::TH < p.H) && (w.x0 >= 1) && (w.x0 + G::TW < p.W);
        if (interior && f1_rel) return;
        f1_rel = interior;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int pix = (wave + 4 * j) * 32 + li;
            const int pc = min(pix, G::NPIX - 1);
            const int ly = pc / G::LW, lx = pc - ly * G::LW;
            int oy = ly, ox = lx;
            if (!interior) {
                oy = reflect_clamp_r(w.y0 + ly - 1, p.H) - w.y0 + 1;
                ox = reflect_clamp_r(w.x0 + lx - 1, p.W) - w.x0 + 1;
                oy = min(max(oy, 0), IH - 3); ox = min(max(ox, 0), IW - 3);      // (only pixels of phantom outputs are clamped)
            }
            const int base = oy * IW + ox;
            va1[j] = half ? base + 2 * IW + 2 : base;              // element 0: tap 0 (half 0) or tap 8 (half 1)
            vb1[j] = half ? NIP : base;                            // elements 1..7: taps 1..7, or the zero tail
            wa1[j] = pix < G::NPIX ? pc * PSR + 4 * half : -1;
        }
    };
    // One pass per item: gather the taps of this wave's pixels once, evaluate BOTH 32-channel chunks (two MFMAs per 32 pixels;
    // the bias rides in the GEMM as tap 9 against a 1.0 in the patch tail), apply ReLU / BatchNorm, write chunk 0 into the tile
    // and keep chunk 1 packed in registers (8 per M-block) until chunk 0 has been multiplied (tile_chunk1).
    h4 keep1[NJ][4];
    auto act_nb = [](float a0, float a1, f32x2 scale, f32x2 shift) __attribute__((always_inline)) -> h2 {
        h2 h = __builtin_convertvector(f32x2{a0, a1}, h2);        // conv output (bias included) -> fp16
        const h2 zero = {0, 0};
        if (!BNF) h = __builtin_elementwise_max(h, zero);
        const f32x2 y = __builtin_convertvector(h, f32x2) * scale + shift;
        h2 o = __builtin_convertvector(y, h2);
        if (BNF) o = __builtin_elementwise_max(o, zero);
        return o;
    };
    auto build_tile = [&](const int par) __attribute__((always_inline)) {
        if (MPRX & 4) return;
        const _Float16* const ip = ipg + par * NIPB;
        const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        h8 x[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            x[j][0] = ip[va1[j]];
            x[j][1] = ip[vb1[j] + 1]; x[j][2] = ip[vb1[j] + 2];
            x[j][3] = ip[vb1[j] + IW]; x[j][4] = ip[vb1[j] + IW + 1]; x[j][5] = ip[vb1[j] + IW + 2];
            x[j][6] = ip[vb1[j] + 2 * IW]; x[j][7] = ip[vb1[j] + 2 * IW + 1];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 s4[4], t4[4];                                     // BN scale / shift of this lane's 16 channels of the chunk
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int cl = 32 * c + rg * 8 + half * 4;
                s4[rg] = *reinterpret_cast<const f32x4*>(&prm1[64 + cl]);
                t4[rg] = *reinterpret_cast<const f32x4*>(&prm1[128 + cl]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if ((wave + 4 * j) * 32 >= G::NPIX) continue;       // (wave-uniform)
                const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[c], x[j], z16, 0, 0, 0);
                // lane = pixel, register r = channel 32c + (r&3) + 8*(r>>2) + 4*half: quads of 4 consecutive channels
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const h2 lo = act_nb(d[rg * 4], d[rg * 4 + 1], f32x2{s4[rg][0], s4[rg][1]}, f32x2{t4[rg][0], t4[rg][1]});
                    const h2 hi = act_nb(d[rg * 4 + 2], d[rg * 4 + 3], f32x2{s4[rg][2], s4[rg][3]}, f32x2{t4[rg][2], t4[rg][3]});
                    const h4 v = h4{lo[0], lo[1], hi[0], hi[1]};
                    if (c == 0) { if (wa1[j] >= 0) *reinterpret_cast<h4*>(&lds[wa1[j] + rg * 8]) = v; }
                    else keep1[j][rg] = v;
                }
            }
        }
    };
    auto tile_chunk1 = [&]() __attribute__((always_inline)) {
        if (MPRX & 4) return;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if ((wave + 4 * j) * 32 >= G::NPIX) continue;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                if (wa1[j] >= 0) *reinterpret_cast<h4*>(&lds[wa1[j] + rg * 8]) = keep1[j][rg];
        }
    };
    if constexpr (F1) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * half + e;
                w1f[c][e] = k < 9 ? (_Float16)p.w1[k * 64 + 32 * c + li] : k == 9 ? (_Float16)p.b1[32 * c + li] : (_Float16)0.f;
            }
    }

    const int a_base = (((2 * wave) * G::MBH + li / MBW) * G::LW + (li % MBW)) * PSR + half * 8;
    constexpr int A_MB = G::MBH * G::LW * PSR;
    constexpr int RA = (NG == 2) ? 3 : 2;                // operand rings: fragments are fetched RA - 1 steps ahead (168 registers with three groups)
    h8 af[RA][2], bf[RA][2], stg[G::NITER];
    constexpr int S0 = 2;                                // first step that issues a staging load
    auto a_off = [](int s) -> int {
        const int tap = s >> 1, gg = s & 1;
        return ((tap / 3) * G::LW + tap % 3) * PSR + gg * 16;
    };
    auto mma = [](const h8& a, const h8& b, const f32x16& cc) -> f32x16 {
        return SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, cc, 0, 0, 0)
                    : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, cc, 0, 0, 0);
    };

    Where cur = decode(item);
    const _Float16* src = nullptr;
    auto lds_write = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < G::NITER; ++j) {
            const int f = gt + j * 256;
            if (f < G::NV) {
                h8 v = stg[j];
#if defined(MPRX) && (MPRX & 1)
                v = h8{1, 0, 0, 0, 0, 0, 0, 0};      // developer timing experiment (results wrong): no dependence on the staging loads
#endif
                if (cur_pad && goff[j] < 0) v = h8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<h8*>(&lds[(f >> 2) * PSR + (f & 3) * 8]) = v;
            }
        }
    };
    int par = 0;                                               // F1: patch buffer of the current item
    if constexpr (F1) {
        patch_load(cur);
        patch_write(0);
        f1_offsets(cur);
        if (item + stride < item_end) patch_load(decode(item + stride));
        group_barrier(ctr, bar_target, lane);                  // patch visible to the group
        build_tile(0);
    } else {
        src = offsets(cur);
        cur_pad = !goff_rel;
#pragma unroll
        for (int j = 0; j < G::NITER; ++j)
            stg[j] = *reinterpret_cast<const h8*>(src + (goff[j] >= 0 ? goff[j] : 0));
        lds_write();
    }
    group_barrier(ctr, bar_target, lane);

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef MP_TIMING
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        MPR_T(t_item);
        f32x16 acc[2][2];
        const int item_next = item + stride;
        const bool has_next = item_next < item_end;
        Where nxt = cur;
        const _Float16* const cur_src = src;
        bool nxt_pad = cur_pad;

        // one chunk of 32 input channels: 18 steps of 4 MFMAs; C = chunk (0, 1)
        auto chunk_body = [&](auto c_tag) __attribute__((always_inline)) {
            constexpr int C = decltype(c_tag)::value;
            constexpr bool LAST = (C == 1);
            const _Float16* in_next = nullptr;
            if constexpr (F1) {
                if (LAST && has_next) nxt = decode(item_next);
            } else if (!LAST) {
                in_next = cur_src + CKR;
            } else {
                if (has_next) {
                    nxt = decode(item_next);
                    src = offsets(nxt);
                    nxt_pad = !goff_rel;
                }
                in_next = src;                                  // !has_next: dummy re-read of the current tile
            }
            // weight fragments of step s of this chunk: resident block (tap*4 + 2*C + g) -- the packed order of a 64-channel chunk
            auto w_off = [](int s) -> int { return (((s >> 1) * 4 + 2 * C + (s & 1)) * 2) * 512; };
            MPR_T(t_s0);
            if (C == 0) MPR_ADD(0, t_item, t_s0);                  // item start -> first step
#pragma unroll
            for (int s = 0; s < RA - 1; ++s) {
                af[s][0] = *reinterpret_cast<const h8*>(&lds[a_base + a_off(s)]);
                af[s][1] = *reinterpret_cast<const h8*>(&lds[a_base + A_MB + a_off(s)]);
                bf[s][0] = *reinterpret_cast<const h8*>(&wl[w_off(s) + lane * 8]);
                bf[s][1] = *reinterpret_cast<const h8*>(&wl[w_off(s) + 512 + lane * 8]);
            }
#pragma unroll
            for (int s = 0; s < ((MPRX & 16) ? 1 : G::STEPS); ++s) {
                constexpr bool Z = (C == 0);
                acc[0][0] = mma(af[s % RA][0], bf[s % RA][0], (Z && s == 0) ? zero16 : acc[0][0]);
                __builtin_amdgcn_sched_barrier(0);
                if (s + RA - 1 < G::STEPS) {
                    const int sn = s + RA - 1;
                    bf[sn % RA][0] = *reinterpret_cast<const h8*>(&wl[w_off(sn) + lane * 8]);
                    bf[sn % RA][1] = *reinterpret_cast<const h8*>(&wl[w_off(sn) + 512 + lane * 8]);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][1] = mma(af[s % RA][0], bf[s % RA][1], (Z && s == 0) ? zero16 : acc[0][1]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (F1) {
                    // the next item's image patch: registers -> LDS (the other patch buffer: last read two barriers ago), then
                    // the patch of the item after it -> registers
                    if (LAST && s == 2 && has_next) patch_write(par ^ 1);
                    if (LAST && s == 4 && item_next + stride < item_end) patch_load(decode(item_next + stride));
                } else {
                    const int j = s - S0;
                    if (s >= S0 && j < G::NITER)      // unconditional load: keeps the compiler's vmcnt counting exact
                        stg[j] = *reinterpret_cast<const h8*>(in_next + (goff[j] >= 0 ? goff[j] : 0));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[1][0] = mma(af[s % RA][1], bf[s % RA][0], (Z && s == 0) ? zero16 : acc[1][0]);
                __builtin_amdgcn_sched_barrier(0);
                acc[1][1] = mma(af[s % RA][1], bf[s % RA][1], (Z && s == 0) ? zero16 : acc[1][1]);
                __builtin_amdgcn_sched_barrier(0);
                if (s + RA - 1 < G::STEPS) {
                    const int sn = s + RA - 1;
                    af[sn % RA][0] = *reinterpret_cast<const h8*>(&lds[a_base + a_off(sn)]);
                    af[sn % RA][1] = *reinterpret_cast<const h8*>(&lds[a_base + A_MB + a_off(sn)]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            MPR_T(t_s1);
            MPR_ADD(1, t_s0, t_s1);                                // the MFMA steps of a chunk
            group_barrier(ctr, bar_target, lane);              // this chunk's tile fully consumed by the group
            MPR_T(t_b0);
            MPR_ADD(2, t_s1, t_b0);                                // barrier behind the steps
            if constexpr (F1) {
                if (!LAST) {
                    tile_chunk1();
                } else if (has_next) {
                    f1_offsets(nxt);
                    build_tile(par ^ 1);
                }
            } else {
                if (LAST) cur_pad = nxt_pad;
                // (pooled layers: the staged registers are free again before the epilogue; un-pooled ones: the epilogue stages in the
                //  tile first, see `ostage` above)
                if (!LAST || (has_next && (POOL || (MPRX & 2)))) lds_write();
            }
            MPR_T(t_w);
            MPR_ADD(3, t_b0, t_w);                                 // tile production / LDS write
            if (!LAST) group_barrier(ctr, bar_target, lane);
            MPR_T(t_b1);
            MPR_ADD(4, t_w, t_b1);                                 // barrier in front of chunk 1
        };
        chunk_body(std::integral_constant<int, 0>{});
        chunk_body(std::integral_constant<int, 1>{});

        // ---------------- epilogue of item `cur` (conv_f16.hip's, one 64-channel slice) ----------------
        MPR_T(t_e0);
        const int img = cur.img, y0 = cur.y0, x0 = cur.x0;
        if (MPRX & 2) {
            float sink = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sink += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
            if (sink == 123.456f) p.out[tid] = (_Float16)sink;
        } else if constexpr (POOL) {
            // lane = channel (li), register r = pixel (r&3) + 8*(r>>2) + 4*half of the M-block; registers r, r+1 are
            // horizontally adjacent pixels -> one packed pair
            f32x2 bia[2], scl[2], sft[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const float b = prm[nb * 32 + li], sc = prm[64 + nb * 32 + li], sh = prm[128 + nb * 32 + li];
                bia[nb] = f32x2{b, b}; scl[nb] = f32x2{sc, sc}; sft[nb] = f32x2{sh, sh};
            }
            const int Ho = p.H >> 1, Wo = p.W >> 1;
            const int cs = p.out_cstride;
            // Pool BEFORE the activation: bias add, fp16 rounding and ReLU are non-decreasing and the BatchNorm affine is monotonic in the direction of its
            // scale's sign, so the maximum of a window's four activations IS the activation of the maximum (scale < 0: the minimum) of its
            // four accumulators, bit for bit -- one activation per pooled value instead of four (channels li and 32 + li share a packed pair)
            const f32x2 biap = {bia[0][0], bia[1][0]}, sclp = {scl[0][0], scl[1][0]}, sftp = {sft[0][0], sft[1][0]};
            const bool neg0 = sclp[0] < 0.f, neg1 = sclp[1] < 0.f;
            auto pooled_first = [&](const float (&q)[2][4]) __attribute__((always_inline)) -> h2 {
                const float x0 = fmaxf(fmaxf(q[0][0], q[0][1]), fmaxf(q[0][2], q[0][3])), n0 = fminf(fminf(q[0][0], q[0][1]), fminf(q[0][2], q[0][3]));
                const float x1 = fmaxf(fmaxf(q[1][0], q[1][1]), fmaxf(q[1][2], q[1][3])), n1 = fminf(fminf(q[1][0], q[1][1]), fminf(q[1][2], q[1][3]));
                return act_r2<BNF>(neg0 ? n0 : x0, neg1 ? n1 : x1, biap, sclp, sftp);
            };
            const bool full = (y0 + G::TH <= p.H) && (x0 + G::TW <= p.W);
            const int lane_off = 2 * half * cs + li;
            _Float16* const obase = p.out + ((long long)img * Ho * Wo) * cs + p.out_coff;
            constexpr int RDOWN = (MBW == 32) ? 0 : (MBW == 16) ? 8 : 4;
            constexpr int NMB = (MBW == 32) ? 1 : 2;
            auto store_all = [&](auto full_tag) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        if (RDOWN != 0 && (r & RDOWN) != 0) continue;
                        const int iu = (r & 3) + 8 * (r >> 2);
                        const int oy = (MBW == 32) ? (y0 + 2 * wave) >> 1 : (y0 + (2 * wave + mb) * G::MBH + iu / MBW) >> 1;
                        const int oxu = (x0 + iu % MBW) >> 1;
                        _Float16* const rowp = obase + ((long long)oy * Wo + oxu) * cs;
                        h2 vp;
                        {
                            float q[2][4];
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb) {
                                if constexpr (MBW == 32) {
                                    q[nb][0] = acc[0][nb][r]; q[nb][1] = acc[0][nb][r + 1]; q[nb][2] = acc[1][nb][r]; q[nb][3] = acc[1][nb][r + 1];
                                } else {
                                    q[nb][0] = acc[mb][nb][r]; q[nb][1] = acc[mb][nb][r + 1];
                                    q[nb][2] = acc[mb][nb][r + RDOWN]; q[nb][3] = acc[mb][nb][r + RDOWN + 1];
                                }
                            }
                            vp = pooled_first(q);
                        }
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const _Float16 v = vp[nb];
                            if constexpr (FULL) {
                                rowp[nb * 32 + lane_off] = v;
                            } else {
                                const bool ok = (oy < Ho) & (oxu + 2 * half < Wo);
                                _Float16* dst = ok ? rowp + nb * 32 + lane_off : p.dummy + lane;
                                *dst = v;
                            }
                        }
                    }
            };
            if (full) store_all(std::true_type{}); else store_all(std::false_type{});
        } else {
            // non-pooled: lane = pixel li, register r = channel (r&3) + 8*(r>>2) + 4*half of the N-block.  Per M-block (32 pixels): every lane
            // writes its eight 8-byte channel quads into the wave's staging block [pixel][granule ^ (pixel & 7)][8 halfs], then lane l reads
            // granule l & 7 of pixels l >> 3, + 8, + 16, + 24 and stores 16 bytes: eight lanes = one pixel's 64 channels = one 128-byte line
            const int cs = p.out_cstride;
            _Float16* const obase = p.out + (((long long)img * p.H + y0) * p.W + x0) * cs + p.out_coff;
            const bool full = (y0 + G::TH <= p.H) && (x0 + G::TW <= p.W);
            _Float16* const stg = lds + (16 * wave) * PSR;                          // + stripe (row >> 3) * 64 * PSR + (row & 7) * 64
            int lq = lane;
            asm volatile("" : "+v"(lq));                  // the addresses below are item-invariant: keep hipcc from holding them in registers through the MFMA loop
            const int wrow = ((lq & 31) >> 3) * (64 * PSR) + (lq & 7) * 64 + (lq >> 5) * 4, wsw = lq & 7;   // halfs: this lane's pixel row, + the half's 8 bytes in a granule
            const int rg_l = lq & 7, rp0 = lq >> 3;                                // read side: granule, first pixel
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int cl = nb * 32 + rg * 8 + half * 4;
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&prm[cl]);
                        const f32x4 s4 = *reinterpret_cast<const f32x4*>(&prm[64 + cl]);
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&prm[128 + cl]);
                        const h2 lo = act_r2<BNF>(acc[mb][nb][rg * 4], acc[mb][nb][rg * 4 + 1], f32x2{b4[0], b4[1]},
                                                  f32x2{s4[0], s4[1]}, f32x2{t4[0], t4[1]});
                        const h2 hi = act_r2<BNF>(acc[mb][nb][rg * 4 + 2], acc[mb][nb][rg * 4 + 3], f32x2{b4[2], b4[3]},
                                                  f32x2{s4[2], s4[3]}, f32x2{t4[2], t4[3]});
                        *reinterpret_cast<h4*>(stg + wrow + (((nb * 4 + rg) ^ wsw) << 3)) = h4{lo[0], lo[1], hi[0], hi[1]};
                    }
                asm volatile("" ::: "memory");                                    // (same wave: the LDS executes its operations in order)
                _Float16* const mp = obase + (long long)((2 * wave + mb) * G::MBH) * p.W * cs;
                const int gy0 = y0 + (2 * wave + mb) * G::MBH;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int px = rp0 + 8 * k;                                    // pixel of the M-block (rp0 < 8: px / MBW and the k-part of px % MBW are uniform)
                    const h8 v = *reinterpret_cast<const h8*>(stg + k * (64 * PSR) + rp0 * 64 + ((rg_l ^ rp0) << 3));   // row px = rp0 + 8 k: stripe k
                    _Float16* dst = mp + (((8 * k) / MBW) * p.W + (8 * k) % MBW) * cs + (rp0 * cs + rg_l * 8);
                    if (!full) {
                        const bool okp = (gy0 + px / MBW < p.H) & (x0 + px % MBW < p.W);
                        dst = okp ? dst : p.dummy + lane * 8;
                    }
                    *reinterpret_cast<h8*>(dst) = v;
                }
                asm volatile("" ::: "memory");
            }
        }
        if constexpr (!POOL) { if (has_next && !(MPRX & 2)) lds_write(); }   // the next item's chunk 0 into the tile the epilogue staged in
        MPR_T(t_e1);
        MPR_ADD(5, t_e0, t_e1);                                    // epilogue
#ifdef MP_TIMING
        tsum[7] += 1;
        if (F1 && !has_next && wave == 0 && lane == 0 && grp < 2 && blockIdx.x < 256)
            for (int i = 0; i < 8; ++i) g_timing_r[(blockIdx.x * 2 + grp) * 8 + i] = tsum[i];
#endif
        if (!has_next) return;
        group_barrier(ctr, bar_target, lane);                  // next item's tile complete
        MPR_T(t_b2);
        MPR_ADD(6, t_e1, t_b2);                                    // barrier before the next item
        item = item_next;
        cur = nxt;
        par ^= 1;
    }
}

template <int MBW, bool POOL, int NG, bool F1 = false>
int launch_res(const ConvParamsH& p, hipStream_t s)
{
    const long long nitems = (long long)p.B * p.tiles_x * p.tiles_y;
    if (nitems <= 0) return 0;
    ConvParamsH q = p;
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); };
    q.magic_tx = magic(p.tiles_x); q.magic_ty = magic(p.tiles_y);
    const long long dmax = std::max(p.tiles_x, p.tiles_y);
    if (nitems * dmax >= 0x100000000ll) return 1;      // beyond the 32-bit tile decode: reported as MP_EINVAL
    q.nitems = (int)nitems;
    // persistent workgroups, ONE per CU, each running NG groups (= virtual workgroups) over its XCD's share
    const unsigned grid = persistent_grid((nitems + NG - 1) / NG, p.ncu, p.xcd_shift);
    const ConvParamsH& pp = q;
    if (p.bn_first) hipLaunchKernelGGL((conv_f16_res_kernel<MBW, POOL, true, NG, F1>), dim3(grid), dim3(256 * NG), 0, s, pp);
    else hipLaunchKernelGGL((conv_f16_res_kernel<MBW, POOL, false, NG, F1>), dim3(grid), dim3(256 * NG), 0, s, pp);
    return 0;
}

}  // namespace

// the 3x3 layers whose packed weights fit the LDS once: 64 input and 64 output channels (one slice, one 64-channel chunk)
bool conv_f16_res_supports(const ConvParamsH& p, int taps)
{
    return taps == 9 && p.cin == 64 && p.cout == 64 && p.nslices == 1 && p.in_cstride % 8 == 0 && p.in_coff % 8 == 0;
}

int launch_conv_f16_res(const ConvParamsH& p, int mbw, bool pool, hipStream_t s)
{
    // three groups per CU for the pooled layers, two for the un-pooled ones (round 3, 8-byte stores: enc.conv3 0.45 vs 0.48 ms with two; round 5,
    // whole-line stores staged in the tile: 0.379 vs 0.372 ms alone and 4.02 vs 4.00 ms per step in the pipeline -- equal, and two groups leave
    // 31 KiB of LDS to the side stream's workgroups); MP_DEBUG=f16_res_groups=2 (read per handle in mp_create): two everywhere
    const int ng = (p.res_groups == 2 || !pool) ? 2 : 3;
    if (p.img) {           // first encoder block fused in: the pooled 64 -> 64 layer (enc.conv2), reflection padding
        if (!pool || p.pad_zero) return 2;
        if (mbw == 32) return launch_res<32, true, 2, true>(p, s);
        if (mbw == 16) return launch_res<16, true, 2, true>(p, s);
        return launch_res<8, true, 2, true>(p, s);
    }
    if (ng == 2) {
        if (mbw == 32) return pool ? launch_res<32, true, 2>(p, s) : launch_res<32, false, 2>(p, s);
        if (mbw == 16) return pool ? launch_res<16, true, 2>(p, s) : launch_res<16, false, 2>(p, s);
        return pool ? launch_res<8, true, 2>(p, s) : launch_res<8, false, 2>(p, s);
    }
    if (mbw == 32) return launch_res<32, true, 3>(p, s);          // (ng == 3 is a pooled layer)
    if (mbw == 16) return launch_res<16, true, 3>(p, s);
    return launch_res<8, true, 3>(p, s);
}

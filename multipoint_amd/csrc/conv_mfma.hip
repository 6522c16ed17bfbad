// 3x3 / 1x1 convolution as an implicit GEMM on the exact-fp32 matrix instruction of gfx950
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise a k-ordered fmaf chain).
//
// Replaces, for the MultiPoint encoder and heads (reference multipoint/models/MultiPoint.py:143-148,
// :62-66, :78-82, :168-185), the ATen sequence ReflectionPad2d(1) -> Conv2d(k=3) -> ReLU ->
// BatchNorm2d(eval) [-> MaxPool2d(2,2)] with ONE kernel: the padded halo tile is staged in LDS once
// and re-used by all 9 taps, bias/ReLU/BN-affine/2x2-max run in the epilogue on the accumulators.
//
// GEMM view:  M = output pixels, N = output channels, K = taps * Cin.
//   A[m][k]  = in[pixel m shifted by tap][cin]      (LDS halo tile, read as ds_read_b128)
//   B[k][n]  = packed weights, streamed from L2 straight into VGPRs in MFMA-fragment order
//              (1 KiB fully coalesced global_load_dwordx4 per 32-wide N-block per 8 k)
//   D        = 32x32 fp32 tiles, 16 VGPRs each.  The weight fragment is passed as the MFMA *A* operand and
//              the pixel fragment as *B*, so lane = pixel and each register quad = 4 consecutive channels
//              (16-byte stores in the epilogue)
// Workgroup = 256 threads = 4 waves, tile = 256 pixels x 64 channels; wave = 2x2 MFMA tiles
// (64 accumulator VGPRs).  K is walked in chunks of 32 input channels: per chunk the LDS tile is
// (tile+halo) x 32 ch = ~48 KiB, so three workgroups share a CU and their load/compute phases
// overlap (the fp32 MFMA takes 64 cycles per issue, everything else hides behind it).
#include "mp_common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#ifdef MP_TIMING
// developer instrumentation: per-wave s_memtime stamps of the first 8192 workgroups (slot 15: HW_ID | XCC_ID << 32)
__device__ unsigned long long g_timing[8192 * 4 * 16];
__device__ int g_timing_h = 480;          // only launches whose input height matches are stamped
#define MP_STAMP(i) do { if ((tid & 63) == 0 && blockIdx.x < 8192 && p.H == g_timing_h) g_timing[(blockIdx.x * 4 + (tid >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int mp_debug_select_height(int h) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_timing_h), &h, sizeof(int)); }
extern "C" int mp_debug_read_timing(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timing), sizeof(unsigned long long) * n);
}
#else
#define MP_STAMP(i) do { } while (0)
#endif
#ifdef MP_TIMING
// persistent kernel: wave-uniform per-phase cycle sums, written once per workgroup to g_timing[blockIdx*8 + i]
#define MPP_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define MPP_ADD(slot, a, b) do { tsum[slot] += (b) - (a); } while (0)
#else
#define MPP_T(var) do { } while (0)
#define MPP_ADD(slot, a, b) do { } while (0)
#endif
#if defined(MP_TIMING) && MP_TIMING == 2      // prologue-focused stamps (reuse slots 4..6 of the first chunk)
#define MP_STAMP_P(i) MP_STAMP(i)
#else
#define MP_STAMP_P(i) do { } while (0)
#endif

namespace {

constexpr int CK = 32;        // input channels per LDS chunk
constexpr int PS = CK + 4;    // LDS pixel stride in floats (144 B: 16 consecutive pixels hit
                              // 16 distinct 16-byte bank slots -> conflict-free ds_read_b128)

template <int TAPS, int MBW>
struct Geo {
    static constexpr int MBH = 32 / MBW;
    static constexpr int TW = MBW;
    static constexpr int TH = 256 / MBW;
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int LW = TW + 2 * HALO;
    static constexpr int LH = TH + 2 * HALO;
    static constexpr int NPIX = LW * LH;
    static constexpr int NF4 = NPIX * (CK / 4);          // float4 per chunk tile
    static constexpr int NITER = (NF4 + 255) / 256;      // staging float4 per thread
    static constexpr int STEPS = TAPS * (CK / 8);        // k8-steps per chunk
};

__device__ __forceinline__ int reflect_clamp(int v, int n)
{
    v = v < 0 ? -v : v;                     // ReflectionPad2d(1): -1 -> 1
    v = v >= n ? 2 * (n - 1) - v : v;       //                      n -> n-2
    v = v < 0 ? 0 : v;
    return v >= n ? n - 1 : v;              // (only reachable for pixels outside the image tile)
}

// FUSE1: the input of this convolution is the first encoder block (ReflectionPad -> Conv2d(1,64,3) -> ReLU
// -> BN, reference encoder modules 0-3) of the image: instead of reading a 64-channel tensor from HBM
// (78.6 MB written + re-read per 480x640 image) the (tile+halo) x 32-channel chunk is COMPUTED from a
// (tile + 2-pixel ring) image patch held in LDS, on the VALU, in the shadow of the MFMAs.
// ReLU as an integer max (finite inputs): one v_max_i32, no NaN-canonicalising v_max_f32 in front of it.
// Every VALU instruction outside the MFMA loop matters: fp32 MFMA and VALU share one pipe and a wave that
// is not streaming MFMAs advances only one instruction per MFMA slot of its neighbour (tools/conv_timing.py).
__device__ __forceinline__ float relu_f(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
__device__ __forceinline__ float max4_f(float a, float b, float c, float d)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
}

template <int TAPS, int MBW, bool POOL, bool FUSE1, bool BNF>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvParams p)
{
    using G = Geo<TAPS, MBW>;
    static_assert(!FUSE1 || TAPS == 9, "first-layer fusion is a 3x3 feature");
    constexpr bool RELU = (TAPS == 9);      // the 1x1 head convolutions have no ReLU (p.relu == 0)
    constexpr int ITW = G::TW + 4, ITH = G::TH + 4;          // image patch of the fused first layer
    constexpr int FUSE_FLOATS = FUSE1 ? (ITH * ITW + 9 * 64 + 3 * 64) : 0;
    // SWAP: weights as the MFMA A operand -> lane = pixel, register quad = 4 consecutive channels
    // (16-byte stores).  The pooled epilogue keeps lane = channel: its 2x2 window is then in-lane.
    constexpr bool SWAP = !POOL;
    __shared__ __attribute__((aligned(16))) float lds[G::NPIX * PS + FUSE_FLOATS];
    float* const w1s = lds + G::NPIX * PS;              // [9][64] first-layer weights (16-byte aligned)
    float* const p1s = w1s + 9 * 64;                    // bias | scale | shift, 64 each
    float* const its = p1s + 3 * 64;                    // image patch [ITH][ITW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int li = lane & 31;
    MP_STAMP(0);
#ifdef MP_TIMING
    if ((tid & 63) == 0 && blockIdx.x < 8192 && p.H == g_timing_h) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_timing[(blockIdx.x * 4 + (tid >> 6)) * 16 + 15] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
#endif

    // ---- workgroup -> (image, tile, channel slice); XCD-aware bijective remap so that
    //      consecutive logical ids (neighbouring tiles, same slice set) share one XCD's L2 ----
    int logical;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, nx = 1 << p.xcd_shift;
        const int q = nblk >> p.xcd_shift, r = nblk & (nx - 1), xcd = bid & (nx - 1), pos = bid >> p.xcd_shift;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    }
    // exact division by multiply-high with host-computed magic numbers: stays on the scalar unit (a
    // runtime integer division costs ~20 VALU instructions, and VALU shares the pipe with fp32 MFMA)
    auto udiv = [](unsigned n, unsigned magic, unsigned d) -> unsigned { return d == 1 ? n : __umulhi(n, magic); };
    int tile = (int)udiv((unsigned)logical, p.magic_slices, (unsigned)p.nslices);
    const int slice = logical - tile * p.nslices;

    int img = 0, y0 = 0, x0 = 0;
    long long px0 = 0;
    const float* in_base;      // wave-uniform base: image (3x3) or first pixel of the tile (1x1)
    if constexpr (TAPS == 9) {
        const int trow = (int)udiv((unsigned)tile, p.magic_tx, (unsigned)p.tiles_x);
        const int tx = tile - trow * p.tiles_x;
        const int bi = (int)udiv((unsigned)trow, p.magic_ty, (unsigned)p.tiles_y);
        const int ty = trow - bi * p.tiles_y;
        img = p.img_list ? p.img_list[bi] : bi;
        y0 = ty * G::TH; x0 = tx * G::TW;
        in_base = p.in + (long long)img * p.H * p.W * p.in_cstride + p.in_coff;
    } else {
        px0 = (long long)tile * 256;
        in_base = p.in + px0 * p.in_cstride + p.in_coff;
    }

    // ---- per-thread staging offsets (element offsets from in_base, -1 = store zeros) ----
    // interior tiles (every halo pixel inside the image: ~87 % of the tiles at 480x640) take a lean path:
    // no reflect / clamp / zero logic here and no zero-select at the LDS writes.
    MP_STAMP_P(8);       // block decode done (kernel arguments loaded)
#if defined(MP_TIMING) && MP_TIMING == 3
    {   // calibration: 256 dependent VALU adds between stamps 8 and 9 (how fast does this wave issue?)
        MP_STAMP(8);
        int xx = tid;
#pragma unroll
        for (int i = 0; i < 256; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(xx) : "v"(tid));
        MP_STAMP(9);
        if (xx == 0x7fffffff) p.out[0] = 0.f;
    }
#endif
    bool interior = false;
    if constexpr (TAPS == 9)
        interior = (y0 >= 1) && (y0 + G::TH < p.H) && (x0 >= 1) && (x0 + G::TW < p.W);
    int goff[G::NITER];
    if (interior) {
        const int base = ((y0 - 1) * p.W + (x0 - 1)) * p.in_cstride + (tid & 7) * 4;
#pragma unroll
        for (int j = 0; j < G::NITER; ++j) {
            const int lp = (tid + j * 256) >> 3;
            const int ly = lp / G::LW, lx = lp - ly * G::LW;
            goff[j] = (tid + j * 256 < G::NF4) ? base + (ly * p.W + lx) * (FUSE1 ? 0 : p.in_cstride) : 0;
            if constexpr (FUSE1) goff[j] = (ly + 1) * (G::TW + 4) + (lx + 1);       // window centre in the image patch
        }
    } else {
#pragma unroll
    for (int j = 0; j < G::NITER; ++j) {
        const int f = tid + j * 256;
        const int lp = f >> 3, c4 = f & 7;
        int off = -1;
        if (f < G::NF4) {
            if constexpr (TAPS == 9) {
                const int ly = lp / G::LW, lx = lp - ly * G::LW;
                int gy = y0 + ly - 1, gx = x0 + lx - 1;
                bool zero = false;
                if (p.pad_zero) {
                    zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
                    gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
                } else {
                    gy = reflect_clamp(gy, p.H); gx = reflect_clamp(gx, p.W);
                }
                if constexpr (FUSE1) {
                    // centre of the first-layer window inside the image patch (patch origin = tile - 2);
                    // pixels of partial tiles far outside the image are clamped (their outputs are not stored)
                    const int qy = min(max(gy, y0 - 1), y0 + G::TH), qx = min(max(gx, x0 - 1), x0 + G::TW);
                    if (!zero) off = (qy - y0 + 2) * ITW + (qx - x0 + 2);
                } else {
                    if (!zero) off = (gy * p.W + gx) * p.in_cstride + c4 * 4;
                }
            } else {
                if (px0 + lp < p.total_px) off = lp * p.in_cstride + c4 * 4;
            }
        }
        goff[j] = off;
    }
    }

    // ---- A-fragment LDS base of this lane (M-block 2*wave, tap (0,0), k-group 0) ----
    MP_STAMP_P(9);       // staging offsets done
    const int a_base = (((2 * wave) * G::MBH + li / MBW) * G::LW + (li % MBW)) * PS + half * 4;
    constexpr int A_MB = G::MBH * G::LW * PS;     // second M-block of the wave

    const int nchunks = p.cin / CK;
    // B fragments: [slice][chunk][step][nb][lane][4]
    const f32x4* wp = reinterpret_cast<const f32x4*>(p.wpack) +
                      ((long long)slice * nchunks) * (G::STEPS * 128) + lane;

    f32x16 acc[2][2];        // not zero-initialised: the first MFMA of the tile takes a literal-zero C operand

    // -------- operand pipelines --------------------------------------------------------------
    // B (weights): one linear stream over (chunk, step) straight from L2 into VGPRs, fetched two
    //   steps ahead and NEVER restarted inside a tile (a restart at every chunk boundary stalls
    //   all co-resident waves at once: measured 95 % -> 88 % of peak in a round-1 probe; docs/HISTORY.md A.5).
    //   The last two prefetches of a tile run past the slice (padding in the packed buffer).
    // A (activations): chunk c+1 is fetched global -> VGPR while chunk c is being multiplied
    //   (one 16-byte load per step, steps S0..), and moved VGPR -> LDS between two barriers.
    constexpr int RB = (G::STEPS % 3 == 0) ? 3 : 4;      // B ring size: must divide STEPS
    static_assert(G::STEPS % RB == 0 && G::STEPS % 2 == 0, "operand rings must stay aligned across chunks");
    f32x4 af[2][2], bf[RB][2], stg[G::NITER];
    bf[0][0] = wp[0];   bf[0][1] = wp[64];
    bf[1][0] = wp[128]; bf[1][1] = wp[128 + 64];

    // first-layer value of one staging slot: 4 channels (chunk*32 + (tid&7)*4 ..) of the pixel whose
    // window centre is its[off]; same k-ordered fmaf chain as conv_first_kernel.
    auto fused_slot = [&](int off, int chunk) -> f32x4 {
        const int ch = chunk * CK + (tid & 7) * 4;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float x = its[off + (kh - 1) * ITW + (kw - 1)];
                const f32x4 wv = *reinterpret_cast<const f32x4*>(&w1s[(kh * 3 + kw) * 64 + ch]);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = fmaf(x, wv[e], a[e]);
            }
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&p1s[ch]);
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(&p1s[64 + ch]);
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&p1s[128 + ch]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = a[e] + b4[e];
            if (BNF) v = relu_f(v * s4[e] + t4[e]);
            else v = relu_f(v) * s4[e] + t4[e];
            a[e] = v;
        }
        return a;
    };

    if constexpr (FUSE1) {
        const float* image = p.img + (long long)img * p.H * p.W;
        for (int f = tid; f < ITH * ITW; f += 256) {
            const int r = f / ITW, c = f - r * ITW;
            int gy = y0 - 2 + r, gx = x0 - 2 + c;
            float v;
            if (p.pad_zero) {
                const bool zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
                gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
                v = zero ? 0.f : image[gy * p.W + gx];
            } else {
                v = image[reflect_clamp(gy, p.H) * p.W + reflect_clamp(gx, p.W)];
            }
            its[f] = v;
        }
        for (int f = tid; f < 9 * 64; f += 256) w1s[f] = p.w1[f];
        if (tid < 64) { p1s[tid] = p.b1[tid]; p1s[64 + tid] = p.s1[tid]; p1s[128 + tid] = p.t1[tid]; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G::NITER; ++j) stg[j] = fused_slot(goff[j] >= 0 ? goff[j] : ITW + 1, 0);
    } else {
#pragma unroll
        for (int j = 0; j < G::NITER; ++j)
            stg[j] = *reinterpret_cast<const f32x4*>(in_base + (goff[j] >= 0 ? goff[j] : 0));
    }
    constexpr int S0 = (TAPS == 9) ? 6 : 0;               // first step that issues a staging load
    constexpr int PER_STEP = (TAPS == 9) ? 1 : 2;         // staging loads per step

    MP_STAMP(1);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mma = [](float a, float b, const f32x16& cc) -> f32x16 {
        return SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, cc, 0, 0, 0)
                    : __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, cc, 0, 0, 0);
    };
    auto chunk_body = [&](const int c, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        if (c > 0) MP_STAMP(5);                            // chunk 0 steps done
        if (c > 0) __syncthreads();                        // chunk c-1 fully consumed
        if (interior) {
#pragma unroll
            for (int j = 0; j < G::NITER; ++j) {
                const int f = tid + j * 256;
                if (f < G::NF4) *reinterpret_cast<f32x4*>(&lds[(f >> 3) * PS + (f & 7) * 4]) = stg[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < G::NITER; ++j) {
                const int f = tid + j * 256;
                if (f < G::NF4)      // padding slots (zero pad / beyond the image) are zeroed here, at the consumer
                    *reinterpret_cast<f32x4*>(&lds[(f >> 3) * PS + (f & 7) * 4]) =
                        (goff[j] >= 0) ? stg[j] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (c == 0) MP_STAMP(3);                           // first chunk landed and written to LDS
        __syncthreads();
        MP_STAMP(2 + 2 * (c & 1));                         // 2 / 4: chunk c steps start

        const f32x4* wc = wp + (long long)c * (G::STEPS * 128);
        const bool more = c + 1 < nchunks;
        const float* in_next = in_base + (more ? (c + 1) * CK : 0);     // dummy (re-reads chunk 0) on the last chunk
        af[0][0] = *reinterpret_cast<const f32x4*>(&lds[a_base]);
        af[0][1] = *reinterpret_cast<const f32x4*>(&lds[a_base + A_MB]);
#pragma unroll
        for (int s = 0; s < G::STEPS; ++s) {
            bf[(s + 2) % RB][0] = wc[(s + 2) * 128];
            bf[(s + 2) % RB][1] = wc[(s + 2) * 128 + 64];
            if (s + 1 < G::STEPS) {
                const int sn = s + 1;
                const int tap = sn >> 2, gg = sn & 3;
                const int kh = (TAPS == 9) ? tap / 3 : 0, kw = (TAPS == 9) ? tap % 3 : 0;
                const int aoff = (kh * G::LW + kw) * PS + gg * 8;
                af[sn & 1][0] = *reinterpret_cast<const f32x4*>(&lds[a_base + aoff]);
                af[sn & 1][1] = *reinterpret_cast<const f32x4*>(&lds[a_base + A_MB + aoff]);
            }
            // UNCONDITIONAL loads (dummy address when the slot is padding / there is no next chunk):
            // a load under a branch makes hipcc's s_waitcnt vmcnt(N) conservative, and every step
            // of the staging window then waits for the previous step's HBM load.
#pragma unroll
            for (int u = 0; u < PER_STEP; ++u) {
                const int j = (s - S0) * PER_STEP + u;
                if (s >= S0 && j < G::NITER) {
                    if constexpr (FUSE1) {
                        if (more) stg[j] = fused_slot(goff[j] >= 0 ? goff[j] : ITW + 1, c + 1);
                    } else {
                        stg[j] = *reinterpret_cast<const f32x4*>(in_next + (goff[j] >= 0 ? goff[j] : 0));
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = mma(af[s & 1][0][e], bf[s % RB][0][e], (FIRST && s == 0 && e == 0) ? zero16 : acc[0][0]);
                acc[0][1] = mma(af[s & 1][0][e], bf[s % RB][1][e], (FIRST && s == 0 && e == 0) ? zero16 : acc[0][1]);
                acc[1][0] = mma(af[s & 1][1][e], bf[s % RB][0][e], (FIRST && s == 0 && e == 0) ? zero16 : acc[1][0]);
                acc[1][1] = mma(af[s & 1][1][e], bf[s % RB][1][e], (FIRST && s == 0 && e == 0) ? zero16 : acc[1][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    chunk_body(0, std::true_type{});                     // peeled: its first MFMAs start from C = 0
    for (int c = 1; c < nchunks; ++c) chunk_body(c, std::false_type{});

    MP_STAMP(6);
    // ---------------- epilogue: bias -> (ReLU, BN) | (BN, ReLU) -> [2x2 max] -> store --------
    if constexpr (POOL) {
        // lane = channel (li), register r = pixel (r&3) + 8*(r>>2) + 4*half of the M-block
        float bia[2], scl[2], sft[2];
        int ch[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            ch[nb] = slice * 64 + nb * 32 + li;
            bia[nb] = p.bias[ch[nb]]; scl[nb] = p.scale[ch[nb]]; sft[nb] = p.shift[ch[nb]];
        }
        auto act = [&](float v, int nb) -> float {
            v += bia[nb];
            if (BNF) {
                v = v * scl[nb] + sft[nb];
                if (RELU) v = relu_f(v);
            } else {
                if (RELU) v = relu_f(v);
                v = v * scl[nb] + sft[nb];
            }
            return v;
        };

        {
            const int Ho = p.H >> 1, Wo = p.W >> 1;
            // partner registers of the 2x2 window: +1 column = r+1; +1 row depends on the M-block shape
            if constexpr (MBW == 32) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int oy = (y0 + 2 * wave) >> 1, ox = (x0 + i) >> 1;
                    if (oy < Ho && ox < Wo) {
                        const long long o = (((long long)img * Ho + oy) * Wo + ox) * p.out_cstride + p.out_coff;
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const float v = max4_f(act(acc[0][nb][r], nb), act(acc[0][nb][r + 1], nb),
                                                   act(acc[1][nb][r], nb), act(acc[1][nb][r + 1], nb));
                            if (ch[nb] < p.cout) p.out[o + ch[nb]] = v;
                        }
                    }
                }
            } else {
                constexpr int RDOWN = (MBW == 16) ? 8 : 4;      // register holding the pixel one row down
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        if ((r & RDOWN) != 0) continue;          // only top rows of each 2-row pair
                        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                        const int oy = (y0 + (2 * wave + mb) * G::MBH + i / MBW) >> 1;
                        const int ox = (x0 + i % MBW) >> 1;
                        if (oy < Ho && ox < Wo) {
                            const long long o = (((long long)img * Ho + oy) * Wo + ox) * p.out_cstride + p.out_coff;
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb) {
                                const float v = max4_f(act(acc[mb][nb][r], nb), act(acc[mb][nb][r + 1], nb),
                                                       act(acc[mb][nb][r + RDOWN], nb), act(acc[mb][nb][r + RDOWN + 1], nb));
                                if (ch[nb] < p.cout) p.out[o + ch[nb]] = v;
                            }
                        }
                    }
            }
        }
        MP_STAMP(7);
        return;
    }
    // non-pooled variants
    // D layout (weights as the MFMA A operand): lane = pixel (lane&31) of the M-block, register r =
    // channel (r&3) + 8*(r>>2) + 4*half of the N-block: every register quad is 4 consecutive channels
    // of one pixel -> one 16-byte store.
    auto act4 = [&](const f32x16& a, int rg, const f32x4& b4, const f32x4& s4, const f32x4& t4) -> f32x4 {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = a[rg * 4 + e] + b4[e];
            if (BNF) {
                x = x * s4[e] + t4[e];
                if (RELU) x = relu_f(x);
            } else {
                if (RELU) x = relu_f(x);
                x = x * s4[e] + t4[e];
            }
            v[e] = x;
        }
        return v;
    };
    auto store4 = [&](float* dst, int ch0, const f32x4& v) {
        if (ch0 + 3 < p.cout) {
            *reinterpret_cast<f32x4*>(dst + ch0) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ch0 + e < p.cout) dst[ch0 + e] = v[e];
        }
    };

#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int ch0 = slice * 64 + nb * 32 + rg * 8 + half * 4;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + ch0);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.scale + ch0);
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(p.shift + ch0);
            if constexpr (TAPS == 1) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const long long gp = px0 + (2 * wave + mb) * 32 + li;
                    if (gp < p.total_px)
                        store4(p.out + gp * p.out_cstride + p.out_coff, ch0, act4(acc[mb][nb], rg, b4, s4, t4));
                }
            } else if constexpr (!POOL) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const int oy = y0 + (2 * wave + mb) * G::MBH + li / MBW;
                    const int ox = x0 + li % MBW;
                    if (oy < p.H && ox < p.W)
                        store4(p.out + (((long long)img * p.H + oy) * p.W + ox) * p.out_cstride + p.out_coff, ch0,
                               act4(acc[mb][nb], rg, b4, s4, t4));
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// Persistent variant (all non-fused layers with enough work items): two workgroups per CU walk the (tile, slice)
// items of their XCD's contiguous eighth of the item space.  What it buys over one-workgroup-per-tile:
//   * the next item's first chunk is fetched during the current item's last chunk, so a tile's HBM/L2 latency and
//     its whole prologue disappear from the timeline (in the per-tile kernel a wave in its prologue crawls at one
//     vector instruction per MFMA of its SIMD neighbour: 33-58 k cycles per tile, tools/conv_timing.py);
//   * interior items share ONE set of item-invariant staging offsets -- only the scalar base pointer moves, i.e.
//     no per-item vector address arithmetic at all (every VALU instruction costs fp32-MFMA pipe cycles);
//   * the weight stream runs on from one item into the next, bias/scale/shift sit in LDS, and the staged registers
//     are written to LDS BEFORE the epilogue, so the epilogue runs with them dead (no extra register pressure).
// The inner step loop is the per-tile kernel's, unchanged.
template <int TAPS, int MBW, bool POOL, bool BNF>
__global__ __launch_bounds__(256, 1) void conv_mfma_persist_kernel(const ConvParams p)
{
    using G = Geo<TAPS, MBW>;
    constexpr bool RELU = (TAPS == 9);
    constexpr bool SWAP = !POOL;
    // ONE workgroup per CU (a second MFMA stream per SIMD only gets in the first one's way, see docs/HISTORY.md A.5), which
    // leaves room for a double-buffered LDS image: chunk c+1 is written into the other buffer while chunk c is being
    // multiplied (each staged vector 8 steps after its load was issued), so a chunk boundary is ONE barrier
    static_assert(TAPS == 9, "the persistent kernel handles the 3x3 layers");
    constexpr int BUF = G::NPIX * PS;
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    __shared__ __attribute__((aligned(16))) float prm[3 * 64];      // bias | scale | shift of the current slice

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int li = lane & 31;

    const XcdRange xr = xcd_range(p.nitems, p.xcd_shift);
    const int stride = xr.stride, item_end = xr.item_end;
    int item = xr.item;

    auto udiv = [](unsigned n, unsigned magic, unsigned d) -> unsigned { return d == 1 ? n : __umulhi(n, magic); };
    struct Where { int slice, img, y0, x0; long long px0; const float* in_base; };
    auto decode = [&](int it) __attribute__((always_inline)) -> Where {
        Where w{};
        const int tile = (int)udiv((unsigned)it, p.magic_slices, (unsigned)p.nslices);
        w.slice = it - tile * p.nslices;
        if constexpr (TAPS == 9) {
            const int trow = (int)udiv((unsigned)tile, p.magic_tx, (unsigned)p.tiles_x);
            const int tx = tile - trow * p.tiles_x;
            const int bi = (int)udiv((unsigned)trow, p.magic_ty, (unsigned)p.tiles_y);
            const int ty = trow - bi * p.tiles_y;
            w.img = p.img_list ? p.img_list[bi] : bi;
            w.y0 = ty * G::TH; w.x0 = tx * G::TW;
            w.in_base = p.in + (long long)w.img * p.H * p.W * p.in_cstride + p.in_coff;
        } else {
            w.px0 = (long long)tile * 256;
            w.in_base = p.in + w.px0 * p.in_cstride + p.in_coff;
        }
        return w;
    };
    int goff[G::NITER];
    bool goff_rel = false;          // goff holds the item-invariant relative offsets of interior items
    bool cur_pad = false;           // the LDS image being written has padding slots
    auto is_interior = [&](const Where& w) __attribute__((always_inline)) -> bool {
        if constexpr (TAPS == 9) return (w.y0 >= 1) && (w.y0 + G::TH < p.H) && (w.x0 >= 1) && (w.x0 + G::TW < p.W);
        else return w.px0 + 256 <= p.total_px;
    };
    auto offsets = [&](const Where& w) __attribute__((always_inline)) -> const float* {
        if (is_interior(w)) {
            if (!goff_rel) {
#pragma unroll
                for (int j = 0; j < G::NITER; ++j) {
                    const int f = tid + j * 256;
                    const int lp = f >> 3;
                    int off = 0;
                    if (f < G::NF4) {
                        if constexpr (TAPS == 9) {
                            const int ly = lp / G::LW, lx = lp - ly * G::LW;
                            off = (ly * p.W + lx) * p.in_cstride + (f & 7) * 4;
                        } else {
                            off = lp * p.in_cstride + (f & 7) * 4;
                        }
                    }
                    goff[j] = off;
                }
                goff_rel = true;
            }
            if constexpr (TAPS == 9) return w.in_base + (long long)((w.y0 - 1) * p.W + (w.x0 - 1)) * p.in_cstride;
            else return w.in_base;
        }
        goff_rel = false;
#pragma unroll
        for (int j = 0; j < G::NITER; ++j) {
            const int f = tid + j * 256;
            const int lp = f >> 3, c4 = f & 7;
            int off = -1;
            if (f < G::NF4) {
                if constexpr (TAPS == 9) {
                    const int ly = lp / G::LW, lx = lp - ly * G::LW;
                    int gy = w.y0 + ly - 1, gx = w.x0 + lx - 1;
                    bool zero = false;
                    if (p.pad_zero) {
                        zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
                        gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
                    } else {
                        gy = reflect_clamp(gy, p.H); gx = reflect_clamp(gx, p.W);
                    }
                    if (!zero) off = (gy * p.W + gx) * p.in_cstride + c4 * 4;
                } else {
                    if (w.px0 + lp < p.total_px) off = lp * p.in_cstride + c4 * 4;
                }
            }
            goff[j] = off;
        }
        return w.in_base;
    };

    const int a_base = (((2 * wave) * G::MBH + li / MBW) * G::LW + (li % MBW)) * PS + half * 4;
    constexpr int A_MB = G::MBH * G::LW * PS;
    const int nchunks = p.cin / CK;
    // Weight prefetch distance.  Vector memory loads return IN ORDER, so the wait for a weight fragment also waits
    // for every older load -- including the HBM staging loads of the next image issued in between.  One workgroup
    // per CU has the registers for a ring of 9 (8 steps = 8 k cycles ahead of use, more than the HBM latency);
    // measured against the per-tile kernel's 2-step distance it made no difference here (the staging loads land
    // in time), it is kept as slack.
    constexpr int RB = 9, PF = 8;
    static_assert(G::STEPS % RB == 0 && G::STEPS % 2 == 0, "operand rings must stay aligned across chunks");
    f32x4 af2[3][2], bf[RB][2], stg[G::NITER];
    constexpr int S0 = 6;            // first step that issues a staging load (one per step)
    constexpr int SD = 8;            // a staged vector is written to LDS this many steps after its load
    static_assert(S0 + SD + G::NITER <= G::STEPS, "staging must finish inside the chunk");
    int bufsel = 0;
    auto mma = [](float a, float b, const f32x16& cc) -> f32x16 {
        return SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, cc, 0, 0, 0)
                    : __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, cc, 0, 0, 0);
    };

    if (item >= item_end) return;
    Where cur = decode(item);
    const float* src = offsets(cur);
    cur_pad = !goff_rel;
#pragma unroll
    for (int j = 0; j < G::NITER; ++j)
        stg[j] = *reinterpret_cast<const f32x4*>(src + (goff[j] >= 0 ? goff[j] : 0));
    // staged vector j -> LDS image `buf` (padding slots of boundary items are zeroed here, at the consumer)
    auto lds_put = [&](int j, float* buf, bool pad) __attribute__((always_inline)) {
        const int f = tid + j * 256;
        if (f < G::NF4) {
            f32x4 v = stg[j];
            if (pad && goff[j] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&buf[(f >> 3) * PS + (f & 7) * 4]) = v;
        }
    };
    auto load_prm = [&](int slice) __attribute__((always_inline)) {
        if (tid < 64) {
            prm[tid] = p.bias[slice * 64 + tid]; prm[64 + tid] = p.scale[slice * 64 + tid]; prm[128 + tid] = p.shift[slice * 64 + tid];
        }
    };
#pragma unroll
    for (int j = 0; j < G::NITER; ++j) lds_put(j, lds, cur_pad);
    load_prm(cur.slice);
    const f32x4* wp = reinterpret_cast<const f32x4*>(p.wpack) + ((long long)cur.slice * nchunks) * (G::STEPS * 128) + lane;
#pragma unroll
    for (int s = 0; s < PF; ++s) { bf[s][0] = wp[s * 128]; bf[s][1] = wp[s * 128 + 64]; }
    __syncthreads();

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef MP_TIMING
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool t_on = (p.H == g_timing_h);
#endif
    for (;;) {
        MPP_T(t_item);
        f32x16 acc[2][2];
        const int item_next = item + stride;
        const bool has_next = item_next < item_end;
        Where nxt = cur;
        const f32x4* wnext = wp;
        const float* cur_src = src;
        bool nxt_pad = cur_pad;

        auto chunk_body = [&](const int c, auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool last = c + 1 == nchunks;
            const float* in_next;
            if (!last) {
                in_next = cur_src + (c + 1) * CK;
            } else {
                if (has_next) {
                    nxt = decode(item_next);
                    src = offsets(nxt);
                    nxt_pad = !goff_rel;
                    wnext = reinterpret_cast<const f32x4*>(p.wpack) + ((long long)nxt.slice * nchunks) * (G::STEPS * 128) + lane;
                }
                in_next = src;                                  // !has_next: dummy re-read of the current tile
            }
            const f32x4* wc = wp + (long long)c * (G::STEPS * 128);
            const f32x4* wt = last ? wnext : wc + G::STEPS * 128;      // where the weight prefetch continues
            const bool stage_pad = last ? nxt_pad : cur_pad;           // padding flags of the image being staged
            const float* const rbuf = lds + bufsel * BUF;               // image of this chunk
            float* const wbuf = lds + (bufsel ^ 1) * BUF;               // image being built for the next chunk / item
            MPP_T(t_s0);
            if (c == 0) MPP_ADD(0, t_item, t_s0);
            af2[0][0] = *reinterpret_cast<const f32x4*>(&rbuf[a_base]);
            af2[0][1] = *reinterpret_cast<const f32x4*>(&rbuf[a_base + A_MB]);
            af2[1][0] = *reinterpret_cast<const f32x4*>(&rbuf[a_base + 8]);          // step 1 = tap 0, channel group 1
            af2[1][1] = *reinterpret_cast<const f32x4*>(&rbuf[a_base + A_MB + 8]);
#pragma unroll
            for (int s = 0; s < G::STEPS; ++s) {
                // One wave per SIMD: nothing else fills the matrix pipe while this wave issues its memory instructions,
                // so they are spread over the step -- each group sits in the 64-cycle shadow of the MFMA before it.
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][0] = mma(af2[s % 3][0][e], bf[s % RB][0][e], (FIRST && s == 0 && e == 0) ? zero16 : acc[0][0]);
                    acc[0][1] = mma(af2[s % 3][0][e], bf[s % RB][1][e], (FIRST && s == 0 && e == 0) ? zero16 : acc[0][1]);
                    acc[1][0] = mma(af2[s % 3][1][e], bf[s % RB][0][e], (FIRST && s == 0 && e == 0) ? zero16 : acc[1][0]);
                    acc[1][1] = mma(af2[s % 3][1][e], bf[s % RB][1][e], (FIRST && s == 0 && e == 0) ? zero16 : acc[1][1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (e == 0) {                          // weights PF steps ahead (ring slot consumed in step s-1)
                        if (s + PF < G::STEPS) {
                            bf[(s + PF) % RB][0] = wc[(s + PF) * 128];
                            bf[(s + PF) % RB][1] = wc[(s + PF) * 128 + 64];
                        } else {
                            bf[(s + PF) % RB][0] = wt[(s + PF - G::STEPS) * 128];
                            bf[(s + PF) % RB][1] = wt[(s + PF - G::STEPS) * 128 + 64];
                        }
                    } else if (e == 1) {                   // activation fragments of step s+2 (slot consumed in step s ... see RA)
                        if (s + 2 < G::STEPS) {
                            const int sn = s + 2;
                            const int tap = sn >> 2, gg = sn & 3;
                            const int kh = tap / 3, kw = tap % 3;
                            const int aoff = (kh * G::LW + kw) * PS + gg * 8;
                            af2[sn % 3][0] = *reinterpret_cast<const f32x4*>(&rbuf[a_base + aoff]);
                            af2[sn % 3][1] = *reinterpret_cast<const f32x4*>(&rbuf[a_base + A_MB + aoff]);
                        }
                    } else if (e == 2) {
                        if (s >= S0 && s - S0 < G::NITER)          // unconditional load (exact vmcnt counting)
                            stg[s - S0] = *reinterpret_cast<const f32x4*>(in_next + (goff[s - S0] >= 0 ? goff[s - S0] : 0));
                    } else {
                        if (s >= S0 + SD && s - S0 - SD < G::NITER) lds_put(s - S0 - SD, wbuf, stage_pad);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            MPP_T(t_s1);
            MPP_ADD(1, t_s0, t_s1);
            bufsel ^= 1;
            if (last) cur_pad = nxt_pad;
            __syncthreads();                                   // next image complete, this one fully consumed
            MPP_T(t_b);
            MPP_ADD(2, t_s1, t_b);
        };
        chunk_body(0, std::true_type{});
        for (int c = 1; c < nchunks; ++c) chunk_body(c, std::false_type{});

        MPP_T(t_e0);
        // ---------------- epilogue of item `cur`: wave-uniform 64-bit base + one per-lane offset ----------------
        const int slice = cur.slice, img = cur.img, y0 = cur.y0, x0 = cur.x0;
        const long long px0 = cur.px0;
        const int cs = p.out_cstride;
        if constexpr (POOL) {
            // lane = channel (li), register r = pixel (r&3) + 8*(r>>2) + 4*half of the M-block
            float bia[2], scl[2], sft[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) { bia[nb] = prm[nb * 32 + li]; scl[nb] = prm[64 + nb * 32 + li]; sft[nb] = prm[128 + nb * 32 + li]; }
            auto act = [&](float v, int nb) __attribute__((always_inline)) -> float {
                v += bia[nb];
                if (BNF) {
                    v = v * scl[nb] + sft[nb];
                    if (RELU) v = relu_f(v);
                } else {
                    if (RELU) v = relu_f(v);
                    v = v * scl[nb] + sft[nb];
                }
                return v;
            };
            const int Ho = p.H >> 1, Wo = p.W >> 1;
            const bool full = (y0 + G::TH <= p.H) && (x0 + G::TW <= p.W) && (slice * 64 + 64 <= p.cout);
            const int lane_off = 2 * half * cs + li;
            float* const obase = p.out + ((long long)img * Ho * Wo) * cs + p.out_coff + slice * 64;
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
                        float* const rowp = obase + ((long long)oy * Wo + oxu) * cs;
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const float v = (MBW == 32)
                                ? max4_f(act(acc[0][nb][r], nb), act(acc[0][nb][r + 1], nb), act(acc[1][nb][r], nb), act(acc[1][nb][r + 1], nb))
                                : max4_f(act(acc[mb][nb][r], nb), act(acc[mb][nb][r + 1], nb), act(acc[mb][nb][r + RDOWN], nb),
                                         act(acc[mb][nb][r + RDOWN + 1], nb));
                            if constexpr (FULL) {
                                rowp[nb * 32 + lane_off] = v;
                            } else {
                                if ((oy < Ho) & (oxu + 2 * half < Wo) & (slice * 64 + nb * 32 + li < p.cout)) rowp[nb * 32 + lane_off] = v;
                            }
                        }
                    }
            };
            if (full) store_all(std::true_type{}); else store_all(std::false_type{});
        } else {
            int lane_off;
            float* obase;
            bool full;
            if constexpr (TAPS == 1) {
                lane_off = li * cs + half * 4;
                obase = p.out + px0 * cs + p.out_coff + slice * 64;
                full = (px0 + 256 <= p.total_px) && (slice * 64 + 64 <= p.cout);
            } else {
                lane_off = ((li / MBW) * p.W + li % MBW) * cs + half * 4;
                obase = p.out + (((long long)img * p.H + y0) * p.W + x0) * cs + p.out_coff + slice * 64;
                full = (y0 + G::TH <= p.H) && (x0 + G::TW <= p.W) && (slice * 64 + 64 <= p.cout);
            }
            auto store_all = [&](auto full_tag) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int cl = nb * 32 + rg * 8 + half * 4;
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&prm[cl]);
                        const f32x4 s4 = *reinterpret_cast<const f32x4*>(&prm[64 + cl]);
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&prm[128 + cl]);
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float x = acc[mb][nb][rg * 4 + e] + b4[e];
                                if (BNF) {
                                    x = x * s4[e] + t4[e];
                                    if (RELU) x = relu_f(x);
                                } else {
                                    if (RELU) x = relu_f(x);
                                    x = x * s4[e] + t4[e];
                                }
                                v[e] = x;
                            }
                            float* const mp = (TAPS == 1) ? obase + (long long)((2 * wave + mb) * 32) * cs
                                                          : obase + (long long)((2 * wave + mb) * G::MBH) * p.W * cs;
                            float* const dst = mp + nb * 32 + rg * 8 + lane_off;
                            if constexpr (FULL) {
                                *reinterpret_cast<f32x4*>(dst) = v;
                            } else {
                                bool okp;
                                if constexpr (TAPS == 1) okp = px0 + (2 * wave + mb) * 32 + li < p.total_px;
                                else okp = (y0 + (2 * wave + mb) * G::MBH + li / MBW < p.H) & (x0 + li % MBW < p.W);
                                const int ch0 = slice * 64 + cl;
                                if (okp) {
                                    if (ch0 + 3 < p.cout) {
                                        *reinterpret_cast<f32x4*>(dst) = v;
                                    } else {
#pragma unroll
                                        for (int e = 0; e < 4; ++e)
                                            if (ch0 + e < p.cout) dst[e] = v[e];
                                    }
                                }
                            }
                        }
                    }
            };
            if (full) store_all(std::true_type{}); else store_all(std::false_type{});
        }
        MPP_T(t_e1);
        MPP_ADD(4, t_e0, t_e1);
#ifdef MP_TIMING
        tsum[7] += 1;
        if (!has_next && tid == 0 && t_on)
            for (int i = 0; i < 8; ++i) g_timing[blockIdx.x * 8 + i] = tsum[i];
#endif
        if (!has_next) return;
        if (nxt.slice != cur.slice) {                          // (rare) every wave must be past its epilogue reads of prm
            __syncthreads();
            load_prm(nxt.slice);                               // visible to the next epilogue via the chunk barriers
        }
        item = item_next;
        cur = nxt;
        wp = wnext;
    }
}

template <int TAPS, int MBW, bool POOL, bool FUSE1>
int launch_t(const ConvParams& p, hipStream_t s)
{
    long long ntiles;
    if (TAPS == 9) ntiles = (long long)p.B * p.tiles_x * p.tiles_y;
    else ntiles = (p.total_px + 255) / 256;
    const long long nblk = ntiles * p.nslices;
    if (nblk <= 0) return 0;
    ConvParams q = p;
    // magic = floor(2^32 / d) + 1 gives floor(n / d) == umulhi(n, magic) for all n with n * d < 2^32
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); };
    q.magic_slices = magic(p.nslices); q.magic_tx = magic(p.tiles_x); q.magic_ty = magic(p.tiles_y);
    const long long dmax = std::max(std::max(p.nslices, p.tiles_x), p.tiles_y);
    if (nblk * dmax >= 0x100000000ll) return 1;        // beyond the 32-bit tile decode: reported as MP_EINVAL
    q.nitems = (int)nblk;
    const ConvParams& pp = q;
    if constexpr (!FUSE1) {
        if constexpr (TAPS == 9) {
            // persistent workgroups, ONE per CU, when every workgroup gets enough items (p.persist, default 8) for the
            // tail not to matter
            if (p.persist && nblk >= (long long)p.ncu * p.persist) {
                const unsigned grid = persistent_grid(nblk, p.ncu, p.xcd_shift);
                if (p.bn_first)
                    hipLaunchKernelGGL((conv_mfma_persist_kernel<TAPS, MBW, POOL, true>), dim3(grid), dim3(256), 0, s, pp);
                else
                    hipLaunchKernelGGL((conv_mfma_persist_kernel<TAPS, MBW, POOL, false>), dim3(grid), dim3(256), 0, s, pp);
                return 0;
            }
        }
    }
    if (p.bn_first)
        hipLaunchKernelGGL((conv_mfma_kernel<TAPS, MBW, POOL, FUSE1, true>), dim3((unsigned)nblk), dim3(256), 0, s, pp);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<TAPS, MBW, POOL, FUSE1, false>), dim3((unsigned)nblk), dim3(256), 0, s, pp);
    return 0;
}

}  // namespace

int launch_conv_mfma(const ConvParams& p, int taps, int mbw, bool pool, bool fuse1, hipStream_t s)
{
    if (taps == 1) return launch_t<1, 32, false, false>(p, s);
    if (fuse1) {            // always the pooled second encoder convolution
        if (mbw == 32) return launch_t<9, 32, true, true>(p, s);
        if (mbw == 16) return launch_t<9, 16, true, true>(p, s);
        return launch_t<9, 8, true, true>(p, s);
    }
    if (mbw == 32) return pool ? launch_t<9, 32, true, false>(p, s) : launch_t<9, 32, false, false>(p, s);
    if (mbw == 16) return pool ? launch_t<9, 16, true, false>(p, s) : launch_t<9, 16, false, false>(p, s);
    return pool ? launch_t<9, 8, true, false>(p, s) : launch_t<9, 8, false, false>(p, s);
}

// Fused tail of both heads, fp16 path (`mixed_precision`, BASELINE configs[4]): ONE kernel reads the 3x3 head convolution's fp16
// output once and produces
//   detector   Conv2d(hc, 65, 1) [+ BatchNorm2d(65)] -> Softmax2d -> drop the dustbin -> PixelShuffle(8)
//              (multipoint/models/MultiPoint.py:66-75,150-158), or the logits when force_return_logits is set,
//   descriptor Conv2d(hc, D, 1) [+ BatchNorm2d(D)] -> F.normalize(p=2, dim=1)            (MultiPoint.py:82-86,160-166)
// with the rounding points of autocast (MultiPoint.py:99-104; pinned by tests/golden/forward_f16.npz): the 1x1 convolutions
// accumulate in fp32 on v_mfma_f32_32x32x16_f16 over fp16 inputs and weights, add the fp16 bias and round to fp16; BatchNorm is an
// fp32 affine on that fp16 value, rounded to fp16 again; softmax and the L2 normalisation run in fp32 on those fp16 values.
// Through round 4 this was four launches (two 1x1 convolutions through conv_f16.hip -- the detector's 65 output channels padded
// to 128 there --, det_post, desc_l2norm: 0.19-0.20 ms of the 4.37 ms step, matrix pipe 9-12 % busy), each re-reading its
// predecessor's output from HBM.
//
// Same structure as head_tail.hip (the fp32 path's): per wave 32 pixels x (96 + D) output channels with the WEIGHT fragment as
// the MFMA's A operand, so D[cout][pixel]: a lane owns ONE pixel (lane & 31) and half of every 32-channel block (rows
// (r&3) + 8*(r>>2) + 4*(lane>>5)); softmax and L2 norm are in-register reductions plus one exchange with lane ^ 32.  Both
// operands reach the matrix pipe through LDS, filled by LDS-DMA: K is walked in chunks of 64 channels (the byte geometry of the
// fp32 kernel's 32-channel chunks), double-buffered; per chunk the four waves share ONE copy of the weight fragments
// (4 k-groups x (3 + D/32) KiB, the [slice64][chunk64][kgroup4][nblock2][lane][8 halfs] packing of conv_f16.hip with taps = 1)
// and each wave DMAs the 8 KiB of its own 32 pixels (a lane fetches 16 bytes = 8 channels of its own pixel's row, so the LDS
// image is already in fragment order).  The kernel is a stream: 20 MFMAs of 32 cycles per chunk against 13 DMAs per wave, so
// what bounds it is bytes in flight (one chunk ahead = 32 KiB per CU), not the matrix pipe.
#include "mp_common.h"

#include <type_traits>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16h(const _Float16* sbase, unsigned voff_bytes, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void dma_wait_h() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// conv output pair (fp32 accumulators) -> what autocast leaves: fp16(acc + bias), then the BatchNorm affine in fp32, rounded to
// fp16.  Written on a PAIR exactly like conv_f16.hip's act_h2 (without the ReLU: the 1x1 head convolutions have none,
// MultiPoint.py:66-72,82-86) so that hipcc emits v_pk_add_f32 / v_cvt_pk_f16_f32 / v_pk_fma_f32: the scalar form becomes
// v_fma_mixlo_f16 -- ONE rounding from the exact product-sum to fp16 where autocast rounds to fp32 first and to fp16 second,
// which flips output bits.
__device__ __forceinline__ f32x2 head_act2(float a0, float a1, f32x2 bias, f32x2 scale, f32x2 shift)
{
    const f32x2 x = f32x2{a0, a1} + bias;
    const h2 h = __builtin_convertvector(x, h2);
    f32x2 y = __builtin_convertvector(h, f32x2) * scale + shift;
    asm volatile("" : "+v"(y));      // y exists as an fp32 pair: no fused multiply-add-and-round-to-fp16 even where one half is unused
    const h2 o = __builtin_convertvector(y, h2);
    return __builtin_convertvector(o, f32x2);
}

// ND = D / 32 descriptor blocks (0: no descriptor head)
template <int ND>
__global__ __launch_bounds__(256) void head_tail_f16_kernel(const HeadTailParamsH p)
{
    constexpr int NT = 3 + ND;                                    // accumulator tiles: detector 0..2, descriptor 3..
    constexpr int NP = 96 + 32 * ND;
    constexpr int WCH = 4 * NT * 512;                             // halfs of one chunk's weights: [kgroup4][tile NT][lane][8]
    constexpr int XCH = 2 * 4 * 512;                              // halfs of one wave's X chunk: [det|desc][kgroup4][lane][8]
    constexpr int NBUF = ND <= 2 ? 3 : 2, PD = NBUF - 1;          // chunk buffers / chunks in flight (LDS: 3 x 52 KiB for D = 64)
    constexpr int NDMA = NT + (ND ? 8 : 4);                       // DMAs a wave issues per chunk
    __shared__ __attribute__((aligned(16))) _Float16 wl[NBUF * WCH];
    __shared__ __attribute__((aligned(16))) _Float16 xl[NBUF * 4 * XCH];
    __shared__ __attribute__((aligned(16))) float prm[3 * NP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, hf = lane >> 5;
    for (int i = tid; i < NP; i += 256) {
        const bool det = i < 96;
        const int c = det ? i : i - 96;
        prm[i] = det ? p.bdet[c] : p.bdesc[c];
        prm[NP + i] = det ? p.sdet[c] : p.sdesc[c];
        prm[2 * NP + i] = det ? p.tdet[c] : p.tdesc[c];
    }

    // persistent workgroups over tiles of 128 consecutive pixels, gridDim.x apart (head_tail.hip): the first chunk of the next
    // tile is DMA'd during the last chunk of the current one; waves / lanes beyond the end re-read the last pixel, store nothing
    const long long ntiles = (p.npx + 127) / 128;
    long long tile = blockIdx.x;
    long long px = 0;
    bool valid = false;
    unsigned xoff = 0, xoff_next = 0;
    const _Float16 *xbase = p.x, *xbase_next = p.x;
    auto place = [&](long long t, unsigned& off, const _Float16*& base) __attribute__((always_inline)) {
        const long long q = (t * 4 + wave) * 32 + pl;
        off = (unsigned)(((q < p.npx ? q : p.npx - 1) - t * 128) * p.xstride + hf * 8) * 2u;      // bytes from the tile's first pixel
        base = p.x + t * 128 * p.xstride;
    };
    place(tile, xoff, xbase);
    const int nchunks = p.K >> 6;
    const unsigned wl_lds = (unsigned)(size_t)wl, xl_lds = (unsigned)(size_t)xl + (unsigned)wave * (XCH * 2u);

    // DMA j (0 .. NT + 7) of this wave for chunk c into buffer buf: j < NT weight blocks (this wave's share of the 4 * NT:
    // k-group = wave, tile = j), then the 8 X blocks of its own pixels
    auto chunk_dma = [&](int c, int buf, int j, bool next_tile) __attribute__((always_inline)) {
        if (j < NT) {
            const int nb = j < 3 ? j : j - 3;
            const _Float16* w = j < 3 ? p.wdet : p.wdesc;
            const _Float16* src = w + (((((long long)(nb >> 1) * nchunks + c) * 4 + wave) * 2 + (nb & 1)) * 64) * 8;
            dma16h(src, (unsigned)lane * 16u, wl_lds + (unsigned)(buf * WCH + (wave * NT + j) * 512) * 2u);
        } else {
            const int q = j - NT, part = q >> 2, g = q & 3;
            if (ND == 0 && part == 1) return;
            dma16h((next_tile ? xbase_next : xbase) + part * p.K + c * 64 + g * 16, next_tile ? xoff_next : xoff,
                   xl_lds + (unsigned)(buf * 4 * XCH + (part * 4 + g) * 512) * 2u);
        }
    };

    // ring of NBUF chunk buffers, PD = NBUF - 1 chunks in flight: the kernel is a stream with 20 MFMAs (640 cycles) per chunk, so a
    // chunk's DMAs must be issued well over an HBM round trip (~3-4 k cycles under load) ahead -- with ONE chunk ahead (the
    // fp32 kernel's double buffer, whose chunk is 5 k cycles of MFMAs) every chunk waited for its own latency: 0.130 ms for 16
    // frames 1024x1280; three buffers fit the LDS up to D = 64 (3 x 52 KiB)
    int bufi = 0;                                                 // ring slot of the chunk being multiplied
#pragma unroll
    for (int d = 0; d < PD; ++d) {
#pragma unroll
        for (int j = 0; j < NT + 8; ++j) chunk_dma(d, d, j, false);
    }
    if (PD == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory"); else dma_wait_h();
    __syncthreads();                                              // chunk 0 and prm visible

    for (;;) {
    const bool has_next = tile + gridDim.x < ntiles;
    if (has_next) place(tile + gridDim.x, xoff_next, xbase_next);
    px = (tile * 4 + wave) * 32 + pl;
    valid = px < p.npx;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // the 4 * NT MFMAs of chunk c (ring slot bufi); the NT + 8 DMAs of the chunk PD ahead -- of this tile, or of the next one --
    // ride one per MFMA behind the first ones into the slot that was read last; fragments are fetched one k-group ahead
    for (int c = 0; c < nchunks; ++c) {
        const int ct = c + PD;                                    // the chunk to fetch
        const bool next_tile = ct >= nchunks;
        const bool prefetch = !next_tile || has_next;
        const int cf = next_tile ? ct - nchunks : ct;
        int bt = bufi + PD; bt = bt >= NBUF ? bt - NBUF : bt;
        const _Float16* const wb = wl + bufi * WCH + lane * 8;
        const _Float16* const xb = xl + wave * XCH + bufi * 4 * XCH + lane * 8;
        h8 wf[2][NT], xd[2], xs[2];
        auto frags = [&](int slot, int g) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[slot][t] = *reinterpret_cast<const h8*>(&wb[(g * NT + t) * 512]);
            xd[slot] = *reinterpret_cast<const h8*>(&xb[g * 512]);
            if (ND) xs[slot] = *reinterpret_cast<const h8*>(&xb[(4 + g) * 512]);
        };
        frags(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) frags((g + 1) & 1, g + 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[g & 1][t], t < 3 ? xd[g & 1] : xs[g & 1], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int m = g * NT + t;
                if (m < NT + 8) { if (prefetch) chunk_dma(cf, bt, m, next_tile); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // chunk c + 1 must have landed; with PD = 2 the NDMA operations this chunk issued may stay in flight (the vector memory
        // counter retires loads in order, so "at most NDMA outstanding" means everything older than them -- the previous chunk's
        // DMAs and the epilogue's stores, which count in vmcnt too -- has completed)
        if (PD == 2 && prefetch) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory"); else dma_wait_h();
        __syncthreads();                                          // next chunk landed, this one consumed
        bufi = bufi + 1 == NBUF ? 0 : bufi + 1;
    }

    // ---- detector: bias -> fp16 -> BN -> fp16, softmax over 65 channels in fp32 -> shuffle ----
    const int cell = (int)(px % ((long long)p.Hc * p.Wc));
    const int b = (int)(px / ((long long)p.Hc * p.Wc));
    const int hc = cell / p.Wc, wc = cell - hc * p.Wc;
    {
        float v[32];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int c = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf;              // registers r, r + 1 = channels c, c + 1
                const f32x2 o = head_act2(acc[t][r], acc[t][r + 1], f32x2{prm[c], prm[c + 1]}, f32x2{prm[NP + c], prm[NP + c + 1]},
                                          f32x2{prm[2 * NP + c], prm[2 * NP + c + 1]});
                v[16 * t + r] = o[0]; v[16 * t + r + 1] = o[1];
            }
        // the dustbin (channel 64) is row 0 of block 2: register 0 of the lower half-wave; both halves need it
        float d = head_act2(acc[2][0], acc[2][1], f32x2{prm[64], prm[65]}, f32x2{prm[NP + 64], prm[NP + 65]},
                            f32x2{prm[2 * NP + 64], prm[2 * NP + 65]})[0];
        d = __shfl(d, pl);                                            // from lane pl (hf = 0)
        if (p.logits_nchw && valid) {
            const long long plane = (long long)p.Hc * p.Wc;
            float* o = p.logits_nchw + (long long)b * 65 * plane + cell;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(long long)(32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf) * plane] = v[16 * t + r];
            if (hf == 0) o[64 * plane] = d;
        }
        if (p.prob) {
            // mode 0: nn.Softmax2d (max-subtracted); mode 1: SuperPointMagicLeap.generate_heatmap: exp(x) / (sum + 1e-5)
            float m = 0.f;
            if (p.softmax_mode == 0) {
                m = d;
#pragma unroll
                for (int i = 0; i < 32; ++i) m = fmaxf(m, v[i]);
                m = fmaxf(m, __shfl_xor(m, 32));
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) { v[i] = __expf(v[i] - m); s += v[i]; }
            s += __shfl_xor(s, 32);
            s += __expf(d - m);
            if (p.softmax_mode == 1) s += 0.00001f;
            const float rs = 1.0f / s;                            // one IEEE division per pixel (head_tail.hip)
            if (valid) {
                const int H = p.Hc * 8, W = p.Wc * 8;
                float* o = p.prob + ((long long)b * H + hc * 8) * W + wc * 8 + 4 * hf;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // channels 32t + 8q + 4hf + (0..3) = row dy = 4t + q of the 8x8 block, columns 4hf .. 4hf+3
                        const f32x4 o4 = {v[16 * t + 4 * q] * rs, v[16 * t + 4 * q + 1] * rs, v[16 * t + 4 * q + 2] * rs,
                                          v[16 * t + 4 * q + 3] * rs};
                        *reinterpret_cast<f32x4*>(o + (long long)(4 * t + q) * W) = o4;
                    }
            }
        }
    }
    // ---- descriptor: bias -> fp16 -> BN -> fp16, L2 normalisation in fp32 ----
    if constexpr (ND > 0) {
        if (p.desc) {
            float ss = 0.f;
            f32x4 dv[ND * 4];
#pragma unroll
            for (int t = 0; t < ND; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const int c = 96 + 32 * t + 8 * q + 4 * hf + e;
                        const f32x2 x = head_act2(acc[3 + t][4 * q + e], acc[3 + t][4 * q + e + 1], f32x2{prm[c], prm[c + 1]},
                                                  f32x2{prm[NP + c], prm[NP + c + 1]}, f32x2{prm[2 * NP + c], prm[2 * NP + c + 1]});
                        dv[4 * t + q][e] = x[0]; dv[4 * t + q][e + 1] = x[1];
                        ss += x[0] * x[0];
                        ss += x[1] * x[1];
                    }
            if (p.normalize) {
                ss += __shfl_xor(ss, 32);
                const float rd = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int i = 0; i < ND * 4; ++i) dv[i] = dv[i] * rd;
            }
            if (valid) {
                float* o = p.desc + px * (32 * ND) + 4 * hf;
#pragma unroll
                for (int t = 0; t < ND; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(o + 32 * t + 8 * q) = dv[4 * t + q];
            }
        }
    }
    if (!has_next) return;
    tile += gridDim.x;
    xoff = xoff_next; xbase = xbase_next;
    }
}

}  // namespace

// returns 0 when the fused kernel handled the launch, 1 when the shape is outside what it is instantiated for (the caller then
// runs the separate 1x1 convolution / softmax / normalisation launches)
int launch_head_tail_f16(const HeadTailParamsH& p, hipStream_t s)
{
    if (p.K % 128 != 0 || p.npx <= 0 || p.xstride % 8 != 0) return 1;
    const int D = p.desc ? p.D : 0;
    const long long tiles = (p.npx + 127) / 128;                  // one workgroup per CU walks them
    const dim3 grid((unsigned)(tiles < p.ncu ? tiles : p.ncu)), block(256);
    switch (D) {
    case 0: hipLaunchKernelGGL(head_tail_f16_kernel<0>, grid, block, 0, s, p); return 0;
    case 64: hipLaunchKernelGGL(head_tail_f16_kernel<2>, grid, block, 0, s, p); return 0;
    case 128: hipLaunchKernelGGL(head_tail_f16_kernel<4>, grid, block, 0, s, p); return 0;
    case 256: hipLaunchKernelGGL(head_tail_f16_kernel<8>, grid, block, 0, s, p); return 0;
    default: return 1;
    }
}

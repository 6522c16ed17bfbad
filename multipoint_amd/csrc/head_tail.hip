// Fused tail of both heads (fp32 path): ONE kernel reads the 3x3 head convolution's output once and produces
//   detector   Conv2d(hc, 65, 1) [+ BatchNorm2d(65)] -> Softmax2d -> drop the dustbin -> PixelShuffle(8)
//              (multipoint/models/MultiPoint.py:66-75,150-158), or the logits when force_return_logits is set,
//   descriptor Conv2d(hc, D, 1) [+ BatchNorm2d(D)] -> F.normalize(p=2, dim=1)            (MultiPoint.py:82-86,160-166)
// instead of four launches (two 1x1 convolutions through the generic implicit-GEMM kernel -- the detector's 65 output
// channels padded to 128 there --, softmax + shuffle, L2 norm), each re-reading its predecessor's output from HBM.
//
// GEMM view per wave: 32 pixels x (96 + D) output channels, K = hc, on v_mfma_f32_32x32x2_f32 with the WEIGHT fragment as
// the A operand: D[cout][pixel], so a lane owns ONE pixel (lane & 31) and half of every 32-channel block (rows
// (r&3) + 8*(r>>2) + 4*(lane>>5)).  The softmax and the L2 norm are then in-register reductions plus one exchange with
// lane ^ 32, every register quad is 4 consecutive channels (= 4 horizontally adjacent pixels of the shuffled heat map, or
// 16 bytes of the descriptor row), and nothing goes through LDS except the 3 x (96 + D) bias / scale / shift values.
// Operands reach the matrix pipe through LDS, filled by LDS-DMA (conv_wino.hip's lesson: a DMA is free next to MFMAs, a
// register load costs 10-45 cycles of matrix-pipe time): K is walked in chunks of 32 channels, double-buffered; per chunk
// the workgroup's four waves share ONE copy of the weight fragments (4 x (3 + D/32) KiB, packed by pack_conv_weights with
// taps = 1) and each wave DMAs the 8 KiB of its own 32 pixels (a lane fetches 16 bytes of its own pixel's row, so the LDS
// image is already in fragment order); one s_waitcnt vmcnt(0) + barrier per chunk.
#include "mp_common.h"

#include <type_traits>

namespace {

__device__ __forceinline__ float acc_rd(float a)
{
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a));
    return x;
}
// LDS-DMA, see conv_wino.hip: 64 lanes x 16 bytes from (uniform base + per-lane byte offset) to LDS [lds_byte + 16*lane, +16)
__device__ __forceinline__ void dma16(const float* sbase, unsigned voff_bytes, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ND = D / 32 descriptor blocks (0: no descriptor head)
template <int ND>
__global__ __launch_bounds__(256) void head_tail_kernel(const HeadTailParams p)
{
    constexpr int NT = 3 + ND;                                    // accumulator tiles: detector 0..2, descriptor 3..
    constexpr int NP = 96 + 32 * ND;
    constexpr int WCH = 4 * NT * 256;                             // floats of one chunk's weights: [kgroup4][tile NT][lane][4]
    constexpr int XCH = 2 * 4 * 256;                              // floats of one wave's X chunk: [det|desc][kgroup4][lane][4]
    __shared__ __attribute__((aligned(16))) float wl[2 * WCH];
    __shared__ __attribute__((aligned(16))) float xl[2 * 4 * XCH];
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

    // Persistent workgroups: a workgroup walks tiles of 128 consecutive pixels, gridDim.x apart; the first chunk of the next
    // tile is DMA'd during the last chunk of the current one, so no tile pays its HBM latency in the open.  Waves (and
    // lanes) beyond the end re-read the last pixel and store nothing -- every wave takes part in the weight DMAs and barriers.
    const long long ntiles = (p.npx + 127) / 128;
    long long tile = blockIdx.x;
    long long px = 0;
    bool valid = false;
    unsigned xoff = 0, xoff_next = 0;
    const float *xbase = p.x, *xbase_next = p.x;
    auto place = [&](long long t, unsigned& off, const float*& base) __attribute__((always_inline)) {
        const long long q = (t * 4 + wave) * 32 + pl;
        off = (unsigned)(((q < p.npx ? q : p.npx - 1) - t * 128) * p.xstride + hf * 4) * 4u;      // bytes from the tile's first pixel
        base = p.x + t * 128 * p.xstride;
    };
    place(tile, xoff, xbase);
    const int nchunks = p.K >> 5;
    const unsigned wl_lds = (unsigned)(size_t)wl, xl_lds = (unsigned)(size_t)xl + (unsigned)wave * (XCH * 4u);

    // DMA j (0 .. NT + 7) of this wave for chunk c into buffer buf: j < NT weight blocks (this wave's share of the 4 * NT:
    // k-group = wave, tile = j), then the 8 X blocks of its own pixels
    auto chunk_dma = [&](int c, int buf, int j, bool next_tile) __attribute__((always_inline)) {
        if (j < NT) {
            const int nb = j < 3 ? j : j - 3;
            const float* w = j < 3 ? p.wdet : p.wdesc;
            // pack_conv_weights layout [slice64][chunk32][kgroup4][nblock2][lane][4]
            const float* src = w + (((((long long)(nb >> 1) * nchunks + c) * 4 + wave) * 2 + (nb & 1)) * 64) * 4;
            dma16(src, (unsigned)lane * 16u, wl_lds + (unsigned)(buf * WCH + (wave * NT + j) * 256) * 4u);
        } else {
            const int q = j - NT, part = q >> 2, g = q & 3;
            if (ND == 0 && part == 1) return;
            dma16((next_tile ? xbase_next : xbase) + part * p.K + c * 32 + g * 8, next_tile ? xoff_next : xoff,
                  xl_lds + (unsigned)(buf * 4 * XCH + (part * 4 + g) * 256) * 4u);
        }
    };

#pragma unroll
    for (int j = 0; j < NT + 8; ++j) chunk_dma(0, 0, j, false);
    dma_wait();
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

    // the 16 * NT MFMAs of a chunk (buffer parity is a compile-time constant: chunks are unrolled in pairs, K is a multiple of
    // 64 for every model -- 32-channel remainders take the generic tail below); the NT + 8 DMAs of the next chunk ride one
    // per MFMA behind the first ones; fragments are fetched one k-group ahead
    auto chunk = [&](int c, auto buf_tag, bool prefetch, bool next_tile) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_tag)::value;
        const float* const wb = wl + buf * WCH + lane * 4;
        const float* const xb = xl + wave * XCH + buf * 4 * XCH + lane * 4;
        f32x4 wf[2][NT], xd[2], xs[2];
        auto frags = [&](int slot, int g) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[slot][t] = *reinterpret_cast<const f32x4*>(&wb[(g * NT + t) * 256]);
            xd[slot] = *reinterpret_cast<const f32x4*>(&xb[g * 256]);
            if (ND) xs[slot] = *reinterpret_cast<const f32x4*>(&xb[(4 + g) * 256]);
        };
        frags(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) frags((g + 1) & 1, g + 1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[g & 1][t][e], t < 3 ? xd[g & 1][e] : xs[g & 1][e], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const int m = (g * 4 + e) * NT + t;
                    if (prefetch && m < NT + 8) chunk_dma(next_tile ? 0 : c + 1, buf ^ 1, m, next_tile);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        dma_wait();
        __syncthreads();                                          // next chunk landed, this one consumed
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    // (the number of chunks is even for every model: K is a multiple of 64; launch_head_tail checks it)
    for (int c = 0; c < nchunks; c += 2) {
        chunk(c, B0{}, true, false);
        const bool lastc = c + 2 >= nchunks;
        chunk(c + 1, B1{}, !lastc || has_next, lastc);
    }

    // ---- detector: bias -> BN -> softmax over 65 channels -> shuffle ----
    const int cell = (int)(px % ((long long)p.Hc * p.Wc));
    const int b = (int)(px / ((long long)p.Hc * p.Wc));
    const int hc = cell / p.Wc, wc = cell - hc * p.Wc;
    {
        float v[32];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf;
                v[16 * t + r] = (acc_rd(acc[t][r]) + prm[c]) * prm[NP + c] + prm[2 * NP + c];
            }
        // the dustbin (channel 64) is row 0 of block 2: register 0 of the lower half-wave; both halves need it
        float d = (acc_rd(acc[2][0]) + prm[64]) * prm[NP + 64] + prm[2 * NP + 64];
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
            for (int i = 0; i < 32; ++i) { v[i] = __expf(v[i] - m); s += v[i]; }      // v_exp_f32: <= 2 ulp here (|x| <= ~30)
            s += __shfl_xor(s, 32);
            s += __expf(d - m);
            if (p.softmax_mode == 1) s += 0.00001f;
            // one IEEE division per pixel instead of 32 (a division is a ten-instruction sequence): v * (1/s) is within
            // one ulp of v / s
            const float rs = 1.0f / s;
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
    // ---- descriptor: bias -> BN -> L2 normalisation ----
    if constexpr (ND > 0) {
        if (p.desc) {
            float ss = 0.f;
            f32x4 dv[ND * 4];
#pragma unroll
            for (int t = 0; t < ND; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 32 * t + 8 * q + 4 * hf + e;
                        const float x = (acc_rd(acc[3 + t][4 * q + e]) + prm[96 + c]) * prm[NP + 96 + c] + prm[2 * NP + 96 + c];
                        dv[4 * t + q][e] = x;
                        ss += x * x;
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

// returns 0 when the fused kernel handled the launch, 1 when the shape is outside what it is instantiated for (the
// caller then runs the separate 1x1 convolution / softmax / normalisation kernels)
int launch_head_tail(const HeadTailParams& p, hipStream_t s)
{
    if (p.K % 64 != 0 || p.npx <= 0) return 1;
    const int D = p.desc ? p.D : 0;
    const long long tiles = (p.npx + 127) / 128;                  // one workgroup per CU walks them
    const dim3 grid((unsigned)(tiles < p.ncu ? tiles : p.ncu)), block(256);
    switch (D) {
    case 0: hipLaunchKernelGGL(head_tail_kernel<0>, grid, block, 0, s, p); return 0;
    case 64: hipLaunchKernelGGL(head_tail_kernel<2>, grid, block, 0, s, p); return 0;
    case 128: hipLaunchKernelGGL(head_tail_kernel<4>, grid, block, 0, s, p); return 0;
    case 256: hipLaunchKernelGGL(head_tail_kernel<8>, grid, block, 0, s, p); return 0;
    default: return 1;
    }
}

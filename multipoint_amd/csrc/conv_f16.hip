// fp16 convolution path (model config `mixed_precision: true`, reference MultiPoint.py:99-103: the forward
// runs under torch.cuda.amp.autocast, i.e. Conv2d on fp16 inputs/weights with fp32 accumulation and fp16
// outputs, BatchNorm evaluated in fp32 on fp16 activations, fp16 results).  BASELINE config 5.
//
// Same implicit GEMM as conv_mfma.hip with the byte geometry kept: an LDS pixel is 64 fp16 channels (128 B,
// stride 144 B -> conflict-free ds_read_b128), a step is (tap, 16-channel group) and one 16-byte operand per
// lane feeds ONE v_mfma_f32_32x32x16_f16 (32 cycles, 16x the fp32 instruction's rate), so the operand
// streams have to be fetched much further ahead: weights 5 steps (ring of 6), activations 2 steps (ring of 3).
// Rounding points (all round-to-nearest-even, as autocast produces them):
//   conv: fp32 accumulate (+ fp16 bias) -> fp16;  ReLU;  BN: fp32 affine on the fp16 value -> fp16;  max-pool.
#include "mp_common.h"

#include <algorithm>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int CKH = 64;        // input channels per LDS chunk
constexpr int PSH = CKH + 8;   // LDS pixel stride in halfs (144 B)

template <int TAPS, int MBW>
struct GeoH {
    static constexpr int MBH = 32 / MBW;
    static constexpr int TW = MBW;
    static constexpr int TH = 256 / MBW;
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int LW = TW + 2 * HALO;
    static constexpr int LH = TH + 2 * HALO;
    static constexpr int NPIX = LW * LH;
    static constexpr int NV = NPIX * (CKH / 8);          // 16-byte vectors per chunk tile
    static constexpr int NITER = (NV + 255) / 256;       // staging vectors per thread
    static constexpr int STEPS = TAPS * (CKH / 16);      // k16-steps per chunk
};

__device__ __forceinline__ int reflect_clamp_h(int v, int n)
{
    v = v < 0 ? -v : v;
    v = v >= n ? 2 * (n - 1) - v : v;
    v = v < 0 ? 0 : v;
    return v >= n ? n - 1 : v;
}

__device__ __forceinline__ float round_h(float v) { return (float)(_Float16)v; }

// conv result (fp32 accumulator) -> activation value as autocast produces it (still held in fp32)
template <bool RELU, bool BNF>
__device__ __forceinline__ float act_h(float acc, float bias, float scale, float shift)
{
    float v = round_h(acc + bias);
    if (BNF) {
        v = round_h(v * scale + shift);
        if (RELU) v = fmaxf(v, 0.f);
    } else {
        if (RELU) v = fmaxf(v, 0.f);
        v = round_h(v * scale + shift);
    }
    return v;
}

template <int TAPS, int MBW, bool POOL, bool BNF>
__global__ __launch_bounds__(256, 2) void conv_f16_kernel(const ConvParamsH p)
{
    using G = GeoH<TAPS, MBW>;
    constexpr bool RELU = (TAPS == 9);
    constexpr bool SWAP = !POOL;      // weights as the MFMA A operand -> lane = pixel, register quad = 4 channels
    __shared__ __attribute__((aligned(16))) _Float16 lds[G::NPIX * PSH];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int li = lane & 31;

    int logical;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, pos = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    }
    auto udiv = [](unsigned n, unsigned magic, unsigned d) -> unsigned { return d == 1 ? n : __umulhi(n, magic); };
    const int tile = (int)udiv((unsigned)logical, p.magic_slices, (unsigned)p.nslices);
    const int slice = logical - tile * p.nslices;

    int img = 0, y0 = 0, x0 = 0;
    long long px0 = 0;
    const _Float16* in_base;
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

    // per-thread staging offsets in halfs from in_base (-1: padding slot, zero-filled at the LDS write)
    int goff[G::NITER];
#pragma unroll
    for (int j = 0; j < G::NITER; ++j) {
        const int f = tid + j * 256;
        const int lp = f >> 3, c8 = f & 7;
        int off = -1;
        if (f < G::NV) {
            if constexpr (TAPS == 9) {
                const int ly = lp / G::LW, lx = lp - ly * G::LW;
                int gy = y0 + ly - 1, gx = x0 + lx - 1;
                bool zero = false;
                if (p.pad_zero) {
                    zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
                    gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
                } else {
                    gy = reflect_clamp_h(gy, p.H); gx = reflect_clamp_h(gx, p.W);
                }
                if (!zero) off = (gy * p.W + gx) * p.in_cstride + c8 * 8;
            } else {
                if (px0 + lp < p.total_px) off = lp * p.in_cstride + c8 * 8;
            }
        }
        goff[j] = off;
    }

    const int a_base = (((2 * wave) * G::MBH + li / MBW) * G::LW + (li % MBW)) * PSH + half * 8;
    constexpr int A_MB = G::MBH * G::LW * PSH;

    const int nchunks = p.cin / CKH;
    // B fragments: [slice][chunk][step][nb][lane][8 halfs]
    const h8* wp = reinterpret_cast<const h8*>(p.wpack) + ((long long)slice * nchunks) * (G::STEPS * 128) + lane;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    constexpr int RB = (G::STEPS % 6 == 0) ? 6 : 4;      // weight ring (divides STEPS: stays aligned across chunks)
    constexpr int PF = RB - 1;                           // steps of weight prefetch
    constexpr int RA = 3;                                // activation-fragment ring (restarted per chunk)
    static_assert(G::STEPS % RB == 0, "weight ring must stay aligned across chunks");
    h8 af[RA][2], bf[RB][2], stg[G::NITER];
#pragma unroll
    for (int s = 0; s < PF; ++s) { bf[s][0] = wp[s * 128]; bf[s][1] = wp[s * 128 + 64]; }
#pragma unroll
    for (int j = 0; j < G::NITER; ++j)
        stg[j] = *reinterpret_cast<const h8*>(in_base + (goff[j] >= 0 ? goff[j] : 0));

    constexpr int S0 = (TAPS == 9) ? 4 : 0;               // first step that issues a staging load
    constexpr int PER_STEP = (TAPS == 9) ? 1 : 2;
    auto a_off = [](int s) -> int {
        const int tap = s >> 2, gg = s & 3;
        const int kh = (TAPS == 9) ? tap / 3 : 0, kw = (TAPS == 9) ? tap % 3 : 0;
        return (kh * G::LW + kw) * PSH + gg * 16;
    };
    auto mma = [](const h8& a, const h8& b, const f32x16& cc) -> f32x16 {
        return SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, cc, 0, 0, 0)
                    : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, cc, 0, 0, 0);
    };

    for (int c = 0; c < nchunks; ++c) {
        if (c > 0) __syncthreads();                        // chunk c-1 fully consumed
#pragma unroll
        for (int j = 0; j < G::NITER; ++j) {
            const int f = tid + j * 256;
            if (f < G::NV) {
                h8 v = stg[j];
                if (goff[j] < 0) v = h8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<h8*>(&lds[(f >> 3) * PSH + (f & 7) * 8]) = v;
            }
        }
        __syncthreads();

        const h8* wc = wp + (long long)c * (G::STEPS * 128);
        const bool more = c + 1 < nchunks;
        const _Float16* in_next = in_base + (more ? (c + 1) * CKH : 0);   // dummy (re-reads chunk 0) on the last chunk
#pragma unroll
        for (int s = 0; s < RA - 1; ++s) {
            af[s][0] = *reinterpret_cast<const h8*>(&lds[a_base + a_off(s)]);
            af[s][1] = *reinterpret_cast<const h8*>(&lds[a_base + A_MB + a_off(s)]);
        }
#pragma unroll
        for (int s = 0; s < G::STEPS; ++s) {
            bf[(s + PF) % RB][0] = wc[(s + PF) * 128];
            bf[(s + PF) % RB][1] = wc[(s + PF) * 128 + 64];
            if (s + RA - 1 < G::STEPS) {
                const int sn = s + RA - 1;
                af[sn % RA][0] = *reinterpret_cast<const h8*>(&lds[a_base + a_off(sn)]);
                af[sn % RA][1] = *reinterpret_cast<const h8*>(&lds[a_base + A_MB + a_off(sn)]);
            }
#pragma unroll
            for (int u = 0; u < PER_STEP; ++u) {
                const int j = (s - S0) * PER_STEP + u;
                if (s >= S0 && j < G::NITER)      // unconditional load: keeps the compiler's vmcnt counting exact
                    stg[j] = *reinterpret_cast<const h8*>(in_next + (goff[j] >= 0 ? goff[j] : 0));
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = mma(af[s % RA][0], bf[s % RB][0], acc[0][0]);
            acc[0][1] = mma(af[s % RA][0], bf[s % RB][1], acc[0][1]);
            acc[1][0] = mma(af[s % RA][1], bf[s % RB][0], acc[1][0]);
            acc[1][1] = mma(af[s % RA][1], bf[s % RB][1], acc[1][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---------------- epilogue ----------------
    if constexpr (POOL) {
        // lane = channel (li), register r = pixel (r&3) + 8*(r>>2) + 4*half of the M-block
        float bia[2], scl[2], sft[2];
        int ch[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            ch[nb] = slice * 64 + nb * 32 + li;
            bia[nb] = p.bias[ch[nb]]; scl[nb] = p.scale[ch[nb]]; sft[nb] = p.shift[ch[nb]];
        }
        const int Ho = p.H >> 1, Wo = p.W >> 1;
        auto pooled = [&](float a, float b, float c, float d, int nb) -> _Float16 {
            const float v = fmaxf(fmaxf(act_h<RELU, BNF>(a, bia[nb], scl[nb], sft[nb]), act_h<RELU, BNF>(b, bia[nb], scl[nb], sft[nb])),
                                  fmaxf(act_h<RELU, BNF>(c, bia[nb], scl[nb], sft[nb]), act_h<RELU, BNF>(d, bia[nb], scl[nb], sft[nb])));
            return (_Float16)v;
        };
        if constexpr (MBW == 32) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int oy = (y0 + 2 * wave) >> 1, ox = (x0 + i) >> 1;
                if (oy < Ho && ox < Wo) {
                    const long long o = (((long long)img * Ho + oy) * Wo + ox) * p.out_cstride + p.out_coff;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        if (ch[nb] < p.cout)
                            p.out[o + ch[nb]] = pooled(acc[0][nb][r], acc[0][nb][r + 1], acc[1][nb][r], acc[1][nb][r + 1], nb);
                }
            }
        } else {
            constexpr int RDOWN = (MBW == 16) ? 8 : 4;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    if ((r & RDOWN) != 0) continue;
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int oy = (y0 + (2 * wave + mb) * G::MBH + i / MBW) >> 1;
                    const int ox = (x0 + i % MBW) >> 1;
                    if (oy < Ho && ox < Wo) {
                        const long long o = (((long long)img * Ho + oy) * Wo + ox) * p.out_cstride + p.out_coff;
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            if (ch[nb] < p.cout)
                                p.out[o + ch[nb]] = pooled(acc[mb][nb][r], acc[mb][nb][r + 1], acc[mb][nb][r + RDOWN],
                                                           acc[mb][nb][r + RDOWN + 1], nb);
                    }
                }
        }
        return;
    }
    // non-pooled: lane = pixel, register r = channel (r&3) + 8*(r>>2) + 4*half of the N-block -> 8-byte stores
    auto store4 = [&](_Float16* dst, int ch0, const h4& v) {
        if (ch0 + 3 < p.cout) {
            *reinterpret_cast<h4*>(dst + ch0) = v;
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
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                h4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (_Float16)act_h<RELU, BNF>(acc[mb][nb][rg * 4 + e], b4[e], s4[e], t4[e]);
                if constexpr (TAPS == 1) {
                    const long long gp = px0 + (2 * wave + mb) * 32 + li;
                    if (gp < p.total_px) store4(p.out + gp * p.out_cstride + p.out_coff, ch0, v);
                } else {
                    const int oy = y0 + (2 * wave + mb) * G::MBH + li / MBW;
                    const int ox = x0 + li % MBW;
                    if (oy < p.H && ox < p.W)
                        store4(p.out + (((long long)img * p.H + oy) * p.W + ox) * p.out_cstride + p.out_coff, ch0, v);
                }
            }
        }
}

template <int TAPS, int MBW, bool POOL>
void launch_h(const ConvParamsH& p, hipStream_t s)
{
    long long ntiles;
    if (TAPS == 9) ntiles = (long long)p.B * p.tiles_x * p.tiles_y;
    else ntiles = (p.total_px + 255) / 256;
    const long long nblk = ntiles * p.nslices;
    if (nblk <= 0) return;
    ConvParamsH q = p;
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); };
    q.magic_slices = magic(p.nslices); q.magic_tx = magic(p.tiles_x); q.magic_ty = magic(p.tiles_y);
    const long long dmax = std::max(std::max(p.nslices, p.tiles_x), p.tiles_y);
    if (nblk * dmax >= 0x100000000ll) return;
    const ConvParamsH& pp = q;
    if (p.bn_first)
        hipLaunchKernelGGL((conv_f16_kernel<TAPS, MBW, POOL, true>), dim3((unsigned)nblk), dim3(256), 0, s, pp);
    else
        hipLaunchKernelGGL((conv_f16_kernel<TAPS, MBW, POOL, false>), dim3((unsigned)nblk), dim3(256), 0, s, pp);
}

// ---- first encoder block, fp16 flavour: image (fp32 in HBM, rounded to fp16 as autocast casts the conv input)
//      -> 64 fp16 channels.  HBM-write bound: thread = (pixel, 8 channels), one 16-byte store.
constexpr int FTH = 8, FTW = 32, FLW = FTW + 2, FLH = FTH + 2;

__global__ __launch_bounds__(256) void conv_first_f16_kernel(const Conv1ParamsH p)
{
    __shared__ float tile[FLH * FLW];
    const int tid = threadIdx.x;
    const int tiles_x = (p.W + FTW - 1) / FTW, tiles_y = (p.H + FTH - 1) / FTH;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int bi = t / tiles_y;
    const int img = p.img_list ? p.img_list[bi] : bi;
    const int y0 = ty * FTH, x0 = tx * FTW;
    const float* in = p.in + (long long)img * p.H * p.W;
    for (int f = tid; f < FLH * FLW; f += 256) {
        const int ly = f / FLW, lx = f - ly * FLW;
        int gy = y0 + ly - 1, gx = x0 + lx - 1;
        float v;
        if (p.pad_zero) {
            const bool zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
            gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
            v = zero ? 0.f : in[gy * p.W + gx];
        } else {
            v = in[reflect_clamp_h(gy, p.H) * p.W + reflect_clamp_h(gx, p.W)];
        }
        tile[f] = round_h(v);
    }
    const int c8 = (tid & 7) * 8;
    float w[9][8], bia[8], scl[8], sft[8];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[k][e] = p.w[k * 64 + c8 + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bia[e] = p.bias[c8 + e]; scl[e] = p.scale[c8 + e]; sft[e] = p.shift[c8 + e]; }
    __syncthreads();
    _Float16* out = p.out + (long long)img * p.H * p.W * 64;
    const int psub = tid >> 3;                    // 32 pixels per pass
#pragma unroll 2
    for (int it = 0; it < (FTH * FTW) / 32; ++it) {
        const int pix = it * 32 + psub;
        const int py = pix / FTW, px = pix % FTW;
        float x[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = tile[(py + kh) * FLW + px + kw];
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) a = fmaf(x[k], w[k][e], a);
            o[e] = (_Float16)(p.bn_first ? act_h<true, true>(a, bia[e], scl[e], sft[e])
                                         : act_h<true, false>(a, bia[e], scl[e], sft[e]));
        }
        const int oy = y0 + py, ox = x0 + px;
        if (oy < p.H && ox < p.W) *reinterpret_cast<h8*>(out + ((long long)oy * p.W + ox) * 64 + c8) = o;
    }
}

}  // namespace

void launch_conv_f16(const ConvParamsH& p, int taps, int mbw, bool pool, hipStream_t s)
{
    if (taps == 1) { launch_h<1, 32, false>(p, s); return; }
    if (mbw == 32) { pool ? launch_h<9, 32, true>(p, s) : launch_h<9, 32, false>(p, s); }
    else if (mbw == 16) { pool ? launch_h<9, 16, true>(p, s) : launch_h<9, 16, false>(p, s); }
    else { pool ? launch_h<9, 8, true>(p, s) : launch_h<9, 8, false>(p, s); }
}

void launch_conv_first_f16(const Conv1ParamsH& p, hipStream_t s)
{
    const int tiles_x = (p.W + FTW - 1) / FTW, tiles_y = (p.H + FTH - 1) / FTH;
    const long long nblk = (long long)p.B * tiles_x * tiles_y;
    if (nblk <= 0) return;
    hipLaunchKernelGGL(conv_first_f16_kernel, dim3((unsigned)nblk), dim3(256), 0, s, p);
}

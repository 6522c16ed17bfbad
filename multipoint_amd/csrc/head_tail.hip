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
// Operands come straight from global memory: a lane's X fragment is 16 bytes of its own pixel's row (the 128-byte line is
// consumed over four k-groups out of L1), the weight fragments (packed by pack_conv_weights, taps = 1) stream from L2.
#include "mp_common.h"

namespace {

__device__ __forceinline__ float acc_rd(float a)
{
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a));
    return x;
}

// ND = D / 32 descriptor blocks (0: no descriptor head)
template <int ND>
__global__ __launch_bounds__(256) void head_tail_kernel(const HeadTailParams p)
{
    constexpr int NT = 3 + ND;                                    // accumulator tiles: detector 0..2, descriptor 3..
    __shared__ __attribute__((aligned(16))) float prm[3 * (96 + 32 * (ND ? ND : 1))];
    const int tid = threadIdx.x, lane = tid & 63;
    const int pl = lane & 31, hf = lane >> 5;
    constexpr int NP = 96 + 32 * ND;
    for (int i = tid; i < NP; i += 256) {
        const bool det = i < 96;
        const int c = det ? i : i - 96;
        prm[i] = det ? p.bdet[c] : p.bdesc[c];
        prm[NP + i] = det ? p.sdet[c] : p.sdesc[c];
        prm[2 * NP + i] = det ? p.tdet[c] : p.tdesc[c];
    }
    __syncthreads();

    const long long tile = (long long)blockIdx.x * 4 + (tid >> 6);
    const long long px0 = tile * 32;
    if (px0 >= p.npx) return;
    const long long px = px0 + pl;
    const bool valid = px < p.npx;
    const float* xrow = p.x + (valid ? px : p.npx - 1) * p.xstride + hf * 4;
    const int nchunks = p.K >> 5;
    // weight fragment of N-block nb (0..), k-group kg: pack_conv_weights layout [slice64][chunk32][kgroup4][nblock2][lane][4]
    auto wfrag = [&](const float* w, int nb, int kg) __attribute__((always_inline)) -> f32x4 {
        const long long off = (((((long long)(nb >> 1) * nchunks + (kg >> 2)) * 4 + (kg & 3)) * 2 + (nb & 1)) * 64 + lane) * 4;
        return *reinterpret_cast<const f32x4*>(w + off);
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int ngroups = p.K >> 3;
    // operand ring, RD k-groups deep: vector memory returns in order, so a weight fragment (L2) issued behind an X fragment
    // (first touch of a line: HBM) arrives no earlier than that -- everything is fetched RD - 1 groups ahead
    constexpr int RD = (ND <= 4) ? 3 : 2;
    f32x4 xd[RD], xs[RD], wf[RD][NT];
    // one operand load of k-group kg into ring slot buf: j = 0 detector X, 1 descriptor X, 2.. weight fragment j - 2
    auto fetch1 = [&](int buf, int kg, int j) __attribute__((always_inline)) {
        kg = kg < ngroups ? kg : ngroups - 1;
        if (j == 0) xd[buf] = *reinterpret_cast<const f32x4*>(xrow + kg * 8);
        else if (j == 1) { if (ND) xs[buf] = *reinterpret_cast<const f32x4*>(xrow + p.K + kg * 8); }
        else wf[buf][j - 2] = (j - 2) < 3 ? wfrag(p.wdet, j - 2, kg) : wfrag(p.wdesc, j - 5, kg);
    };
    auto fetch = [&](int buf, int kg) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NT + 2; ++j) fetch1(buf, kg, j);
    };
    // the 4 * NT MFMAs of ring slot buf; the NT + 2 loads of the group that refills slot nbuf ride ONE per MFMA pair
    // behind them: a burst of back-to-back vector loads costs ~45 cycles of matrix-pipe time each (tools/mfma_probe10.hip:
    // 10 cycles at one load per MFMA, 32-48 at two to four)
    auto step = [&](int buf, int nbuf, int nkg, bool prefetch) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[buf][t][e], t < 3 ? xd[buf][e] : xs[buf][e], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int m = e * NT + t;
                if (prefetch && (m & 1) == 0 && (m >> 1) < NT + 2) fetch1(nbuf, nkg, m >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
#pragma unroll
    for (int i = 0; i < RD - 1; ++i) fetch(i, i);
    // K is a multiple of 32 channels = 4 groups; the ring index must be a compile-time constant, so RD groups per trip
    int kg = 0;
    for (; kg + RD <= ngroups; kg += RD) {
#pragma unroll
        for (int i = 0; i < RD; ++i) step(i, (i + RD - 1) % RD, kg + i + RD - 1, true);
    }
#pragma unroll
    for (int i = 0; i < RD; ++i)
        if (kg + i < ngroups) step(i, (i + RD - 1) % RD, kg + i + RD - 1, i + 1 < RD);

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
}

}  // namespace

// returns 0 when the fused kernel handled the launch, 1 when the shape is outside what it is instantiated for (the
// caller then runs the separate 1x1 convolution / softmax / normalisation kernels)
int launch_head_tail(const HeadTailParams& p, hipStream_t s)
{
    if (p.K % 32 != 0 || p.npx <= 0) return 1;
    const int D = p.desc ? p.D : 0;
    const long long tiles = (p.npx + 31) / 32;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    switch (D) {
    case 0: hipLaunchKernelGGL(head_tail_kernel<0>, grid, block, 0, s, p); return 0;
    case 64: hipLaunchKernelGGL(head_tail_kernel<2>, grid, block, 0, s, p); return 0;
    case 128: hipLaunchKernelGGL(head_tail_kernel<4>, grid, block, 0, s, p); return 0;
    case 256: hipLaunchKernelGGL(head_tail_kernel<8>, grid, block, 0, s, p); return 0;
    default: return 1;
    }
}

// Detector / descriptor head epilogues.
//   det_post : nn.Softmax2d over the 65 detector channels, drop the dustbin, nn.PixelShuffle(8)
//              (reference multipoint/models/MultiPoint.py:74-75,153-158; PixelShuffle(8) ==
//              utils.depth_to_space, multipoint/utils/utils.py:64-69:
//              prob[n,0,8h+i,8w+j] = softmax(logits)[n,8i+j,h,w])
//   desc_l2  : torch.nn.functional.normalize(x, p=2, dim=1)  (MultiPoint.py:163-164), eps 1e-12
// Both are wave-level kernels: reductions are DPP/shuffle trees inside 8- or 16-lane groups.
#include "mp_common.h"

namespace {

// One wave = 8 horizontally adjacent coarse cells; lane l: cell l>>3, column j = l&7 of the 8x8
// block; registers i = 0..7 hold channel 8i+j.  For a fixed i the wave writes 64 contiguous floats.
template <typename T>
__global__ __launch_bounds__(256) void det_post_kernel(const T* __restrict__ logits, int lstride,
                                                      int B, int Hc, int Wc, float* __restrict__ prob,
                                                      float* __restrict__ logits_nchw, int mode)
{
    const int lane = threadIdx.x & 63;
    const int groups_x = (Wc + 7) >> 3;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long total = (long long)B * Hc * groups_x;
    if (wid >= total) return;
    const int gx = (int)(wid % groups_x);
    const int h = (int)((wid / groups_x) % Hc);
    const int b = (int)(wid / ((long long)groups_x * Hc));
    const int cell = lane >> 3, j = lane & 7;
    const int w = gx * 8 + cell;
    const bool valid = w < Wc;
    const long long cidx = ((long long)b * Hc + h) * Wc + (valid ? w : Wc - 1);
    const T* lp = logits + cidx * lstride;

    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)lp[8 * i + j];
    const float d = (float)lp[64];

    if (logits_nchw && valid) {
        const long long plane = (long long)Hc * Wc;
        float* o = logits_nchw + (long long)b * 65 * plane + (long long)h * Wc + w;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[(8 * i + j) * plane] = v[i];
        if (j == 0) o[64 * plane] = d;
    }
    if (!prob) return;

    // mode 0: nn.Softmax2d (max-subtracted); mode 1: SuperPointMagicLeap.generate_heatmap
    // (multipoint/models/SuperPointMagicLeap.py:72-73): exp(x) / (sum(exp(x)) + 1e-5), no max subtraction
    float m = 0.f;
    if (mode == 0) {
        m = d;
#pragma unroll
        for (int i = 0; i < 8; ++i) m = fmaxf(m, v[i]);
        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = expf(v[i] - m); s += v[i]; }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    s += expf(d - m);
    if (mode == 1) s += 0.00001f;

    const int H = Hc * 8, W = Wc * 8;
    if (valid) {
        float* o = prob + ((long long)b * H + h * 8) * W + w * 8 + j;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[(long long)i * W] = v[i] / s;
    }
}

// rows of D floats; LPP = D/4 lanes per pixel (16, 32 or 64), one float4 per lane.
template <typename T>
__global__ __launch_bounds__(256) void desc_l2norm_kernel(const T* __restrict__ raw,
                                                         float* __restrict__ out, long long npx,
                                                         int D, int normalize)
{
    const int lpp = D >> 2;
    const int ppw = 64 / lpp;                               // pixels per wave
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long px = wid * ppw + lane / lpp;
    const bool valid = px < npx;
    const int c = (lane % lpp) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (valid) {
        if constexpr (sizeof(T) == 4) {
            v = *reinterpret_cast<const f32x4*>(raw + px * D + c);
        } else {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 hv = *reinterpret_cast<const h4*>(raw + px * D + c);
            v = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        }
    }
    if (normalize) {
        float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        for (int off = 1; off < lpp; off <<= 1) s += __shfl_xor(s, off);
        const float denom = fmaxf(sqrtf(s), 1e-12f);
        v[0] /= denom; v[1] /= denom; v[2] /= denom; v[3] /= denom;
    }
    if (valid) *reinterpret_cast<f32x4*>(out + px * D + c) = v;
}

}  // namespace

void launch_det_post(const float* logits, int lstride, int B, int Hc, int Wc, float* prob,
                     float* logits_nchw, int mode, hipStream_t s)
{
    const long long waves = (long long)B * Hc * ((Wc + 7) / 8);
    if (waves <= 0) return;
    hipLaunchKernelGGL(det_post_kernel<float>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, logits,
                       lstride, B, Hc, Wc, prob, logits_nchw, mode);
}

void launch_det_post_f16(const _Float16* logits, int lstride, int B, int Hc, int Wc, float* prob,
                         float* logits_nchw, int mode, hipStream_t s)
{
    const long long waves = (long long)B * Hc * ((Wc + 7) / 8);
    if (waves <= 0) return;
    hipLaunchKernelGGL(det_post_kernel<_Float16>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, logits,
                       lstride, B, Hc, Wc, prob, logits_nchw, mode);
}

void launch_desc_l2norm(const float* raw, float* out, long long npx, int D, int normalize,
                        hipStream_t s)
{
    const int ppw = 64 / (D / 4);
    const long long waves = (npx + ppw - 1) / ppw;
    if (waves <= 0) return;
    hipLaunchKernelGGL(desc_l2norm_kernel<float>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, raw,
                       out, npx, D, normalize);
}

// fp16 head output -> fp32 descriptors (F.normalize computes the norm in fp32 under autocast and the
// division promotes to fp32)
void launch_desc_l2norm_f16(const _Float16* raw, float* out, long long npx, int D, int normalize,
                            hipStream_t s)
{
    const int ppw = 64 / (D / 4);
    const long long waves = (npx + ppw - 1) / ppw;
    if (waves <= 0) return;
    hipLaunchKernelGGL(desc_l2norm_kernel<_Float16>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, raw,
                       out, npx, D, normalize);
}

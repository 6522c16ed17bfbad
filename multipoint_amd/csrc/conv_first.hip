// First encoder block: ReflectionPad2d(1) -> Conv2d(1, 64, 3) -> ReLU -> BatchNorm2d(64)
// (reference multipoint/models/MultiPoint.py:143-148 with N_in = 1, encoder modules 0-3).
// K = 9 only: arithmetic intensity 4.4 FLOP/B, i.e. bound by the 64-channel NHWC output write
// (78.6 MB per 480x640 image).  Direct VALU convolution: 16 lanes share one pixel, each lane owns
// 4 consecutive output channels whose 36 weights + bias/scale/shift live in registers; a wave
// writes 4 adjacent pixels = 1 KiB contiguous per store instruction.
#include "mp_common.h"

namespace {

constexpr int TH = 8, TW = 32;              // output tile per workgroup
constexpr int LW = TW + 2, LH = TH + 2;

__device__ __forceinline__ int reflect_clamp1(int v, int n)
{
    v = v < 0 ? -v : v;
    v = v >= n ? 2 * (n - 1) - v : v;
    v = v < 0 ? 0 : v;
    return v >= n ? n - 1 : v;
}

// C1 = output channels (64, or 32 for channel_version 1 / 2): C1/4 lanes share a pixel
template <int C1>
__global__ __launch_bounds__(256) void conv_first_kernel(const Conv1Params p)
{
    __shared__ float tile[LH * LW];
    const int tid = threadIdx.x;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int bi = t / tiles_y;
    const int img = p.img_list ? p.img_list[bi] : bi;
    const int y0 = ty * TH, x0 = tx * TW;
    const float* in = p.in + (long long)img * p.H * p.W;

    for (int f = tid; f < LH * LW; f += 256) {
        const int ly = f / LW, lx = f - ly * LW;
        int gy = y0 + ly - 1, gx = x0 + lx - 1;
        float v;
        if (p.pad_zero) {
            const bool zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
            gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
            v = zero ? 0.f : in[gy * p.W + gx];
        } else {
            v = in[reflect_clamp1(gy, p.H) * p.W + reflect_clamp1(gx, p.W)];
        }
        tile[f] = v;
    }

    // this lane's 4 output channels
    constexpr int LPP = C1 / 4;                   // lanes per pixel
    const int c4 = (tid % LPP) * 4;
    float w[9][4], bia[4], scl[4], sft[4];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.w + k * C1 + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[k][e] = v[e];
    }
    {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + c4);
        const f32x4 s = *reinterpret_cast<const f32x4*>(p.scale + c4);
        const f32x4 h = *reinterpret_cast<const f32x4*>(p.shift + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bia[e] = b[e]; scl[e] = s[e]; sft[e] = h[e]; }
    }
    __syncthreads();

    float* out = p.out + (long long)img * p.H * p.W * C1;
    constexpr int PPP = 256 / LPP;                // pixels per pass
    const int psub = tid / LPP;
#pragma unroll 4
    for (int it = 0; it < (TH * TW) / PPP; ++it) {
        const int pix = it * PPP + psub;
        const int py = pix / TW, px = pix % TW;
        float x[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = tile[(py + kh) * LW + px + kw];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) a = fmaf(x[k], w[k][e], a);
            a += bia[e];
            if (p.bn_first) a = fmaxf(a * scl[e] + sft[e], 0.f);
            else a = fmaxf(a, 0.f) * scl[e] + sft[e];
            o[e] = a;
        }
        const int oy = y0 + py, ox = x0 + px;
        if (oy < p.H && ox < p.W)
            // streaming store: the 5 GB this layer writes per 64 images are read back once by the next layer, long after
            // they have left the caches (measured 0.91 vs 0.93 ms, and the next layer runs 0.5 % faster)
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(out + ((long long)oy * p.W + ox) * C1 + c4));
    }
}

// double_convolution: false (MultiPoint.py:147-148; generate_encoder :168-172): the block is followed directly by MaxPool2d(2,2).
// Same arithmetic per pixel (and the same multiply-add order) as conv_first_kernel; a lane evaluates the four pixels of one
// pooling window and stores their maximum: out [B][H/2][W/2][C1].
template <int C1>
__global__ __launch_bounds__(256) void conv_first_pool_kernel(const Conv1Params p)
{
    __shared__ float tile[LH * LW];
    const int tid = threadIdx.x;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int bi = t / tiles_y;
    const int img = p.img_list ? p.img_list[bi] : bi;
    const int y0 = ty * TH, x0 = tx * TW;
    const float* in = p.in + (long long)img * p.H * p.W;
    for (int f = tid; f < LH * LW; f += 256) {
        const int ly = f / LW, lx = f - ly * LW;
        int gy = y0 + ly - 1, gx = x0 + lx - 1;
        float v;
        if (p.pad_zero) {
            const bool zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
            gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
            v = zero ? 0.f : in[gy * p.W + gx];
        } else {
            v = in[reflect_clamp1(gy, p.H) * p.W + reflect_clamp1(gx, p.W)];
        }
        tile[f] = v;
    }
    constexpr int LPP = C1 / 4;
    const int c4 = (tid % LPP) * 4;
    float w[9][4], bia[4], scl[4], sft[4];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.w + k * C1 + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[k][e] = v[e];
    }
    {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + c4);
        const f32x4 s = *reinterpret_cast<const f32x4*>(p.scale + c4);
        const f32x4 h = *reinterpret_cast<const f32x4*>(p.shift + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bia[e] = b[e]; scl[e] = s[e]; sft[e] = h[e]; }
    }
    __syncthreads();

    const int Ho = p.H >> 1, Wo = p.W >> 1;
    float* out = p.out + (long long)img * Ho * Wo * C1;
    constexpr int PPP = 256 / LPP;                // pooled pixels per pass
    const int psub = tid / LPP;
    for (int it = 0; it < (TH * TW / 4) / PPP; ++it) {
        const int pix = it * PPP + psub;
        const int qy = pix / (TW / 2), qx = pix % (TW / 2);
        f32x4 o;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const int py = 2 * qy + (sub >> 1), px = 2 * qx + (sub & 1);
            float x[9];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = tile[(py + kh) * LW + px + kw];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) a = fmaf(x[k], w[k][e], a);
                a += bia[e];
                if (p.bn_first) a = fmaxf(a * scl[e] + sft[e], 0.f);
                else a = fmaxf(a, 0.f) * scl[e] + sft[e];
                o[e] = sub == 0 ? a : fmaxf(o[e], a);
            }
        }
        const int oy = (y0 >> 1) + qy, ox = (x0 >> 1) + qx;
        if (oy < Ho && ox < Wo)
            *reinterpret_cast<f32x4*>(out + ((long long)oy * Wo + ox) * C1 + c4) = o;
    }
}

// The same block writing the channel-quad-planar layout [B][C1/4][H][W][4] that conv_wino43.hip consumes: a thread owns ONE pixel
// and walks the C1/4 channel quads (weights and bias / scale / shift of a quad are wave-uniform: LDS broadcast reads), so a
// store instruction writes two 512-byte row segments of one plane.  The multiply-add order is the NHWC kernel's.
template <int C1>
__global__ __launch_bounds__(256) void conv_first_planar_kernel(const Conv1Params p)
{
    __shared__ float tile[LH * LW];
    __shared__ __attribute__((aligned(16))) float wl[9 * C1], pl[3 * C1];
    const int tid = threadIdx.x;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int bi = t / tiles_y;
    const int img = p.img_list ? p.img_list[bi] : bi;
    const int y0 = ty * TH, x0 = tx * TW;
    const float* in = p.in + (long long)img * p.H * p.W;
    for (int f = tid; f < LH * LW; f += 256) {
        const int ly = f / LW, lx = f - ly * LW;
        int gy = y0 + ly - 1, gx = x0 + lx - 1;
        float v;
        if (p.pad_zero) {
            const bool zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
            gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
            v = zero ? 0.f : in[gy * p.W + gx];
        } else {
            v = in[reflect_clamp1(gy, p.H) * p.W + reflect_clamp1(gx, p.W)];
        }
        tile[f] = v;
    }
    for (int f = tid; f < 9 * C1; f += 256) wl[f] = p.w[f];
    for (int f = tid; f < C1; f += 256) { pl[f] = p.bias[f]; pl[C1 + f] = p.scale[f]; pl[2 * C1 + f] = p.shift[f]; }
    __syncthreads();

    const int py = tid / TW, px = tid % TW;
    float x[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = tile[(py + kh) * LW + px + kw];
    const int oy = y0 + py, ox = x0 + px;
    const bool ok = oy < p.H && ox < p.W;
    const long long plane = (long long)p.H * p.W * 4;
    float* out = p.out + (long long)img * (C1 / 4) * plane + ((long long)oy * p.W + ox) * 4;
#pragma unroll 4
    for (int q = 0; q < C1 / 4; ++q) {
        f32x4 wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(&wl[k * C1 + 4 * q]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&pl[4 * q]);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(&pl[C1 + 4 * q]);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(&pl[2 * C1 + 4 * q]);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) a = fmaf(x[k], wv[k][e], a);
            a += b[e];
            if (p.bn_first) a = fmaxf(a * sc[e] + sh[e], 0.f);
            else a = fmaxf(a, 0.f) * sc[e] + sh[e];
            o[e] = a;
        }
        if (ok) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(out + q * plane));
    }
}

}  // namespace

void launch_conv_first(const Conv1Params& p, hipStream_t s)
{
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const long long nblk = (long long)p.B * tiles_x * tiles_y;
    if (nblk <= 0) return;
    if (p.pool) {
        if (p.channels == 32) hipLaunchKernelGGL(conv_first_pool_kernel<32>, dim3((unsigned)nblk), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv_first_pool_kernel<64>, dim3((unsigned)nblk), dim3(256), 0, s, p);
        return;
    }
    if (p.out_planar) {
        if (p.channels == 32) hipLaunchKernelGGL(conv_first_planar_kernel<32>, dim3((unsigned)nblk), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv_first_planar_kernel<64>, dim3((unsigned)nblk), dim3(256), 0, s, p);
        return;
    }
    if (p.channels == 32) hipLaunchKernelGGL(conv_first_kernel<32>, dim3((unsigned)nblk), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_first_kernel<64>, dim3((unsigned)nblk), dim3(256), 0, s, p);
}

// Homographic adaptation on the GPU (reference multipoint/utils/homographies.py:38-189, driven by
// export_keypoints.py:64-103): perspective warps of single-channel maps, the valid mask of a homography, and the
// aggregation of the warped-back heat maps.  All kernels are HBM-streaming: one thread per output pixel, rows of the
// output are contiguous over lanes (coalesced 256-byte stores), gathers go through the texture-less L2 path.
//
//  * warp_kernel            kornia warp_perspective == grid_sample(align_corners=True) in pixel coordinates
//                           (homographies.py:404-433); bilinear / nearest, zeros / reflection padding
//  * valid_mask_kernel      compute_valid_mask (homographies.py:361-389): cv2.warpPerspective(ones, INTER_NEAREST)
//                           + cv2.erode with a (2r+1)^2 box, optional zero ring (mask_border)
//  * accumulate_kernel      count += nearest(valid_mask, H);  prob += bilinear(prob_w, H) * that   (:111-113, :178-180)
//                           for a group of G homographies per launch, summed in the reference's order
//  * begin_kernel         prob = map of the un-warped images, count = 1   (:60-68, :149-150)
//  * finalize_kernel        prob / count, sqrt or * 0.5 for the two-spectra aggregations, min_count   (:115-127, :182-187)
//  * gaussian_filter_kernel ReflectionPad2d + k x k depthwise filter (:55-58, utils.py:124-160)
#include "mp_common.h"

namespace {

struct Hom { double m[9]; };

__device__ __forceinline__ Hom load_hom(const double* p)
{
    Hom h;
#pragma unroll
    for (int i = 0; i < 9; ++i) h.m[i] = p[i];
    return h;
}

// projective map of pixel (x, y); false when the point is at infinity
__device__ __forceinline__ bool project(const Hom& h, int x, int y, float& u, float& v)
{
    const double X = h.m[0] * x + h.m[1] * y + h.m[2];
    const double Y = h.m[3] * x + h.m[4] * y + h.m[5];
    const double Z = h.m[6] * x + h.m[7] * y + h.m[8];
    const double uu = X / Z, vv = Y / Z;
    u = (float)uu; v = (float)vv;
    return isfinite(uu) && isfinite(vv);
}

// ATen grid sampler, padding_mode='reflection', align_corners=True: reflect about 0 and size-1, then clip
__device__ __forceinline__ float reflect_coord(float c, int size)
{
    if (size <= 1) return 0.f;
    const float span = (float)(size - 1);
    c = fabsf(c);
    const float extra = fmodf(c, span);
    const int flips = (int)floorf(c / span);
    c = (flips & 1) ? span - extra : extra;
    return fminf(span, fmaxf(c, 0.f));
}

template <typename LD>
__device__ __forceinline__ float sample_bilinear(LD ld, float ix, float iy, int H, int W)
{
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float x1f = x0f + 1.f, y1f = y0f + 1.f;
    const float nw = (x1f - ix) * (y1f - iy), ne = (ix - x0f) * (y1f - iy);
    const float sw = (x1f - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
    // |coordinate| beyond the int range cannot be in bounds
    if (!(fabsf(ix) < 1e9f) || !(fabsf(iy) < 1e9f)) return 0.f;
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const bool xa = x0 >= 0 && x0 < W, xb = x1 >= 0 && x1 < W;
    const bool ya = y0 >= 0 && y0 < H, yb = y1 >= 0 && y1 < H;
    float out = 0.f;
    if (ya && xa) out += ld(y0, x0) * nw;
    if (ya && xb) out += ld(y0, x1) * ne;
    if (yb && xa) out += ld(y1, x0) * sw;
    if (yb && xb) out += ld(y1, x1) * se;
    return out;
}

// MODE 0 bilinear, 1 nearest; PAD 0 zeros, 1 reflection
template <int MODE, int PAD>
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ src, int n_src, int H, int W,
                                                   const double* __restrict__ hom, int n_out, int Ho, int Wo,
                                                   float* __restrict__ dst)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (x >= Wo || y >= Ho) return;
    const Hom h = load_hom(hom + (size_t)n * 9);
    const float* img = src + (size_t)(n % n_src) * H * W;
    float u, v;
    const bool ok = project(h, x, y, u, v);
    float out = 0.f;
    if (ok) {
        if (PAD == 1) { u = reflect_coord(u, W); v = reflect_coord(v, H); }
        if (MODE == 0) {
            out = sample_bilinear([&](int yy, int xx) { return img[(size_t)yy * W + xx]; }, u, v, H, W);
        } else {
            const float xr = nearbyintf(u), yr = nearbyintf(v);
            if (xr >= 0.f && xr < (float)W && yr >= 0.f && yr < (float)H) out = img[(size_t)(int)yr * W + (int)xr];
        }
    }
    dst[((size_t)n * Ho + y) * Wo + x] = out;
}

// ---------------------------------------------------------------------------------------------------------------
// cv2.warpPerspective(image, M, (W, H), flags=INTER_LINEAR, borderMode=BORDER_REFLECT_101 | BORDER_CONSTANT) as the
// dataset's homographic augmentation calls it (multipoint/datasets/augmentation/augmentation.py:33-36), restated from
// OpenCV's published algorithm (opencv-python==4.2.0.34, imgproc WarpPerspectiveInvoker + remapBilinear<float>):
//  * the destination is cut into blocks 64 wide (bw0 below); per row of a block X0 = M0*xb + M1*y + M2 (likewise Y0,
//    W0) in float64, per pixel W = 32 / (W0 + M6*x1), (X0 + M0*x1)*W rounded half-to-even to an integer in 1/32 px
//    units: source pixel = that >> 5 (saturated to int16), fraction = that & 31;
//  * the four taps are weighted by the float32 table (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy*fx with f = frac/32 and
//    summed left to right in float32 without contraction;
//  * out-of-frame taps: BORDER_CONSTANT -> 0, BORDER_REFLECT_101 -> reflected about the edge pixel centres.
// M is the INVERTED homography (dst -> src), as cv2 computes it before the loop.
__device__ __forceinline__ int cv_border_101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        p = p < 0 ? -p : 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// Every product and sum below is rounded separately, as OpenCV's scalar code does: contraction is switched off for the
// two functions (hipcc's __dmul_rn / __fadd_rn wrappers are inline operators that carry their own `contract` flag).
__device__ __forceinline__ int cv_fixed_coord(double num, double w)
{
#pragma clang fp contract(off)
    double f = num * w;
    f = fmax(-2147483648.0, fmin(2147483647.0, f));
    return (int)rint(f);
}

template <int BORDER>   // 0 BORDER_CONSTANT (value 0), 1 BORDER_REFLECT_101
__global__ __launch_bounds__(256) void cv_warp_linear_kernel(const float* __restrict__ src, int H, int W,
                                                             const double* __restrict__ hom_inv, int bw0,
                                                             float* __restrict__ dst)
{
#pragma clang fp contract(off)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (x >= W || y >= H) return;
    const Hom h = load_hom(hom_inv + (size_t)n * 9);
    const float* img = src + (size_t)n * H * W;
    const int xb = (x / bw0) * bw0;
    const double dxb = (double)xb, dx1 = (double)(x - xb), dy = (double)y;
    const double X0 = (h.m[0] * dxb + h.m[1] * dy) + h.m[2];
    const double Y0 = (h.m[3] * dxb + h.m[4] * dy) + h.m[5];
    const double W0 = (h.m[6] * dxb + h.m[7] * dy) + h.m[8];
    double w = W0 + h.m[6] * dx1;
    w = w != 0.0 ? 32.0 / w : 0.0;
    const int X = cv_fixed_coord(X0 + h.m[0] * dx1, w);
    const int Y = cv_fixed_coord(Y0 + h.m[3] * dx1, w);
    const int sx = max(-32768, min(32767, X >> 5)), sy = max(-32768, min(32767, Y >> 5));
    const float fx = (float)(X & 31) * (1.f / 32.f), fy = (float)(Y & 31) * (1.f / 32.f);
    const float ax = 1.f - fx, ay = 1.f - fy;
    const float w0 = ay * ax, w1 = ay * fx, w2 = fy * ax, w3 = fy * fx;
    float v0, v1, v2, v3;
    if (BORDER == 0) {
        const bool xa = (unsigned)sx < (unsigned)W, xc = (unsigned)(sx + 1) < (unsigned)W;
        const bool ya = (unsigned)sy < (unsigned)H, yc = (unsigned)(sy + 1) < (unsigned)H;
        v0 = (ya && xa) ? img[(size_t)sy * W + sx] : 0.f;
        v1 = (ya && xc) ? img[(size_t)sy * W + sx + 1] : 0.f;
        v2 = (yc && xa) ? img[(size_t)(sy + 1) * W + sx] : 0.f;
        v3 = (yc && xc) ? img[(size_t)(sy + 1) * W + sx + 1] : 0.f;
    } else {
        const int xa = cv_border_101(sx, W), xc = cv_border_101(sx + 1, W);
        const int ya = cv_border_101(sy, H), yc = cv_border_101(sy + 1, H);
        v0 = img[(size_t)ya * W + xa]; v1 = img[(size_t)ya * W + xc];
        v2 = img[(size_t)yc * W + xa]; v3 = img[(size_t)yc * W + xc];
    }
    const float p0 = v0 * w0, p1 = v1 * w1, p2 = v2 * w2, p3 = v3 * w3;
    dst[((size_t)n * H + y) * W + x] = ((p0 + p1) + p2) + p3;
}

// ---------------------------------------------------------------------------------------------------------------
// valid mask: tile 32 x 32 outputs, halo r (<= 16) staged in LDS as the raw cv2.warpPerspective(ones) mask
constexpr int VT = 32, VR_MAX = 16, VL = VT + 2 * VR_MAX;

__global__ __launch_bounds__(256) void valid_mask_kernel(const double* __restrict__ hom_inv, int H, int W, int r,
                                                         int mask_border, unsigned char* __restrict__ mask)
{
#pragma clang fp contract(off)    // OpenCV's scalar code rounds every product and sum separately
    __shared__ unsigned char raw[VL * VL];
    __shared__ unsigned char rowmin[VL * VT];
    const int g = blockIdx.z;
    const Hom h = load_hom(hom_inv + (size_t)g * 9);
    const int x0 = blockIdx.x * VT, y0 = blockIdx.y * VT;
    const int L = VT + 2 * r;
    for (int f = threadIdx.x; f < L * L; f += 256) {
        const int ly = f / L, lx = f - ly * L;
        const int x = x0 + lx - r, y = y0 + ly - r;
        unsigned char m;
        if (x < 0 || x >= W || y < 0 || y >= H) {
            // cv2.erode ignores what lies outside (border value +inf); the zero ring of mask_border erodes it
            m = mask_border ? 0 : 1;
        } else {
            // cv2.warpPerspective, INTER_NEAREST, BORDER_CONSTANT 0 on an all-ones source: 1 iff the rounded
            // (half to even, cvRound) source coordinate lies inside the frame
            const double Z = h.m[6] * x + h.m[7] * y + h.m[8];
            const double iz = Z != 0.0 ? 1.0 / Z : 0.0;
            const double fx = (h.m[0] * x + h.m[1] * y + h.m[2]) * iz;
            const double fy = (h.m[3] * x + h.m[4] * y + h.m[5]) * iz;
            const double rx = rint(fmax(-2147483648.0, fmin(2147483647.0, fx)));
            const double ry = rint(fmax(-2147483648.0, fmin(2147483647.0, fy)));
            m = (rx >= 0.0 && rx < (double)W && ry >= 0.0 && ry < (double)H) ? 1 : 0;
        }
        raw[ly * VL + lx] = m;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < L * VT; f += 256) {
        const int ly = f / VT, lx = f - ly * VT;
        unsigned char m = 1;
        for (int d = 0; d <= 2 * r; ++d) m &= raw[ly * VL + lx + d];
        rowmin[ly * VT + lx] = m;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < VT * VT; f += 256) {
        const int ly = f / VT, lx = f - ly * VT;
        const int x = x0 + lx, y = y0 + ly;
        if (x >= W || y >= H) continue;
        unsigned char m = 1;
        for (int d = 0; d <= 2 * r; ++d) m &= rowmin[(ly + d) * VT + lx];
        mask[((size_t)g * H + y) * W + x] = m;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// AGG 0: one map, 1: product of two maps, 2: their sum (aggregation of the two spectra, homographies.py:63-68,103-108)
template <int AGG>
__global__ __launch_bounds__(256) void accumulate_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                         const unsigned char* __restrict__ mask,
                                                         const double* __restrict__ hom, int G, int B, int H, int W,
                                                         float* __restrict__ prob, float* __restrict__ count)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (x >= W || y >= H) return;
    const size_t o = ((size_t)b * H + y) * W + x;
    float p = prob[o], c = count[o];
    for (int g = 0; g < G; ++g) {
        const Hom h = load_hom(hom + (size_t)g * 9);
        float u, v;
        if (!project(h, x, y, u, v)) continue;
        float cs = 0.f;
        const float xr = nearbyintf(u), yr = nearbyintf(v);
        if (xr >= 0.f && xr < (float)W && yr >= 0.f && yr < (float)H)
            cs = (float)mask[((size_t)g * H + (int)yr) * W + (int)xr];
        const size_t img = ((size_t)g * B + b) * H * W;
        const float s = sample_bilinear(
            [&](int yy, int xx) {
                const size_t i = img + (size_t)yy * W + xx;
                return AGG == 0 ? pa[i] : (AGG == 1 ? pa[i] * pb[i] : pa[i] + pb[i]);
            },
            u, v, H, W);
        c += cs;
        p += s * cs;
    }
    prob[o] = p;
    count[o] = c;
}

// prob = first map (combined with the second one for the two-spectra aggregations), count = 1  (:60-68, :149-150)
__global__ __launch_bounds__(256) void begin_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                    long long n, int aggregation, float* __restrict__ prob,
                                                    float* __restrict__ count)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = pa[i];
    prob[i] = aggregation == 0 ? a : (aggregation == 1 ? a * pb[i] : a + pb[i]);
    count[i] = 1.f;
}

__global__ __launch_bounds__(256) void finalize_kernel(const float* __restrict__ prob, const float* __restrict__ count,
                                                       long long n, int aggregation, float min_count,
                                                       float* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float c = count[i];
    float v = prob[i] / c;
    if (aggregation == 1) v = sqrtf(v);
    else if (aggregation == 2) v *= 0.5f;
    if (min_count > 0.f && c < min_count) v = 0.f;
    out[i] = v;
}

// AGG as above: the two maps are combined after filtering each (homographies.py:55-68)
__global__ __launch_bounds__(256) void gaussian_filter_kernel(const float* __restrict__ in, int H, int W, int k,
                                                              const float* __restrict__ wgt, float* __restrict__ out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (x >= W || y >= H) return;
    const float* img = in + (size_t)b * H * W;
    const int r = (k - 1) / 2;
    float acc = 0.f;
    for (int dy = 0; dy < k; ++dy) {
        int yy = y + dy - r;
        yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
        for (int dx = 0; dx < k; ++dx) {
            int xx = x + dx - r;
            xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
            acc += img[(size_t)yy * W + xx] * wgt[dy * k + dx];
        }
    }
    out[((size_t)b * H + y) * W + x] = acc;
}

dim3 px_grid(int W, int H, int n) { return dim3((W + 63) / 64, (H + 3) / 4, n); }

}  // namespace

void launch_warp_perspective(const float* src, int n_src, int H, int W, const double* hom, int n_out, int Ho, int Wo,
                             int mode, int padding, float* dst, hipStream_t s)
{
    const dim3 g = px_grid(Wo, Ho, n_out);
    if (mode == 0 && padding == 0) warp_kernel<0, 0><<<g, 256, 0, s>>>(src, n_src, H, W, hom, n_out, Ho, Wo, dst);
    else if (mode == 0) warp_kernel<0, 1><<<g, 256, 0, s>>>(src, n_src, H, W, hom, n_out, Ho, Wo, dst);
    else if (padding == 0) warp_kernel<1, 0><<<g, 256, 0, s>>>(src, n_src, H, W, hom, n_out, Ho, Wo, dst);
    else warp_kernel<1, 1><<<g, 256, 0, s>>>(src, n_src, H, W, hom, n_out, Ho, Wo, dst);
}

void launch_cv_warp_linear(const float* src, int n, int H, int W, const double* hom_inv, int border, float* dst,
                           hipStream_t s)
{
    // OpenCV's block width: bh0 = min(16, H); bw0 = min(1024 / bh0, W)
    const int bh0 = H < 16 ? H : 16;
    const int bw0 = (1024 / bh0) < W ? (1024 / bh0) : W;
    const dim3 g = px_grid(W, H, n);
    if (border == 0) cv_warp_linear_kernel<0><<<g, 256, 0, s>>>(src, H, W, hom_inv, bw0, dst);
    else cv_warp_linear_kernel<1><<<g, 256, 0, s>>>(src, H, W, hom_inv, bw0, dst);
}

void launch_ha_valid_mask(const double* hom_inv, int G, int H, int W, int r, int mask_border, unsigned char* mask,
                          hipStream_t s)
{
    valid_mask_kernel<<<dim3((W + VT - 1) / VT, (H + VT - 1) / VT, G), 256, 0, s>>>(hom_inv, H, W, r, mask_border, mask);
}

void launch_ha_accumulate(const float* pa, const float* pb, const unsigned char* mask, const double* hom, int G, int B,
                          int H, int W, int aggregation, float* prob, float* count, hipStream_t s)
{
    const dim3 g = px_grid(W, H, B);
    if (aggregation == 0) accumulate_kernel<0><<<g, 256, 0, s>>>(pa, pb, mask, hom, G, B, H, W, prob, count);
    else if (aggregation == 1) accumulate_kernel<1><<<g, 256, 0, s>>>(pa, pb, mask, hom, G, B, H, W, prob, count);
    else accumulate_kernel<2><<<g, 256, 0, s>>>(pa, pb, mask, hom, G, B, H, W, prob, count);
}

void launch_ha_begin(const float* pa, const float* pb, long long n, int aggregation, float* prob, float* count,
                     hipStream_t s)
{
    begin_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(pa, pb, n, aggregation, prob, count);
}

void launch_ha_finalize(const float* prob, const float* count, long long n, int aggregation, float min_count, float* out,
                        hipStream_t s)
{
    finalize_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(prob, count, n, aggregation, min_count, out);
}

void launch_gaussian_filter(const float* in, int B, int H, int W, int k, const float* wgt, float* out, hipStream_t s)
{
    gaussian_filter_kernel<<<px_grid(W, H, B), 256, 0, s>>>(in, H, W, k, wgt, out);
}

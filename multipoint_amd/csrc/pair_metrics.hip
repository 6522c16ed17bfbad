// GPU-resident pair metrics: the per-sample arithmetic of utils.compute_descriptor_metrics (reference
// multipoint/utils/evaluation.py:287-328) on the device-resident keypoint / match lists.
//   warped_optical  = warp_keypoints(kp_optical, H_gt)            (evaluation.py:287; homographies.py:331-346:
//                     cv2.perspectiveTransform on float64 (x,y) points, result back as (y,x) float64)
//   correct_optical[i][j] = || float32(warped_optical[i] - kp_thermal[j]) ||_2 <= threshold   (:291-292, torch.norm fp32)
//   n_gt            = #rows of `correct` with at least one true entry                          (:297-298)
//   tp[i]           = correct[i][match(i)] for every mutual-NN match                           (:301-311)
//   N               = #warped points inside the image (filter_points, homographies.py:358-372) (:315-316)
// The N x M matrix is never materialised: one thread per query keypoint scans the other image's list from LDS.
#include "mp_common.h"

namespace {

// cv2.perspectiveTransform (64-bit float): w = x*m6 + y*m7 + m8; w = w != 0 ? 1/w : 0; x' = (x*m0 + y*m1 + m2)*w ...
// keypoints are (y, x); m is row-major 3x3 acting on (x, y, 1).
__global__ __launch_bounds__(256) void warp_kernel(const int* __restrict__ kp_yx, const int* __restrict__ kp_count,
                                                   const double* __restrict__ hom, int K, int H, int W,
                                                   double* __restrict__ warped, int* __restrict__ metrics)
{
#pragma clang fp contract(off)    // cv2.perspectiveTransform rounds every product and sum separately
    const int b = blockIdx.y;                    // image slot: 2p = optical, 2p+1 = thermal
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n = min(kp_count[b], K);
    int inside = 0;
    if (k < n) {
        const double* m = hom + (size_t)b * 9;
        const double y = (double)kp_yx[((size_t)b * K + k) * 2], x = (double)kp_yx[((size_t)b * K + k) * 2 + 1];
        double w = (((x * m[6]) + (y * m[7])) + m[8]);
        w = (fabs(w) > 2.220446049250313e-16) ? 1.0 / w : 0.0;
        const double xo = ((((x * m[0]) + (y * m[1])) + m[2]) * w);
        const double yo = ((((x * m[3]) + (y * m[4])) + m[5]) * w);
        warped[((size_t)b * K + k) * 2] = yo;
        warped[((size_t)b * K + k) * 2 + 1] = xo;
        inside = (yo >= 0.0) & (xo >= 0.0) & (yo < (double)H) & (xo < (double)W);
    }
    const unsigned long long bal = __ballot(inside);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&metrics[(b >> 1) * 8 + 4 + (b & 1)], __popcll(bal));
}

// one thread per query keypoint of image slot b (direction b&1: 0 = optical queries vs thermal keypoints)
__global__ __launch_bounds__(256) void correct_kernel(const int* __restrict__ kp_yx, const int* __restrict__ kp_count,
                                                      const double* __restrict__ warped, const int* __restrict__ match_idx,
                                                      int K, float thr, int* __restrict__ inv_idx,
                                                      unsigned char* __restrict__ tp, int* __restrict__ metrics)
{
    __shared__ int oth[256 * 2];
    const int b = blockIdx.y, p = b >> 1, dir = b & 1, ob = b ^ 1;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n = min(kp_count[b], K), m = min(kp_count[ob], K);
    double wy = 0.0, wx = 0.0;
    int mate = -1;                                // index of this keypoint's mutual-NN partner in the other image
    if (k < n) {
        wy = warped[((size_t)b * K + k) * 2]; wx = warped[((size_t)b * K + k) * 2 + 1];
        mate = dir == 0 ? match_idx[(size_t)p * K + k] : inv_idx[(size_t)p * K + k];
    }
    int any = 0, hit = 0;
    for (int j0 = 0; j0 < m; j0 += 256) {
        __syncthreads();
        if (j0 + threadIdx.x < m) {
            oth[threadIdx.x * 2] = kp_yx[((size_t)ob * K + j0 + threadIdx.x) * 2];
            oth[threadIdx.x * 2 + 1] = kp_yx[((size_t)ob * K + j0 + threadIdx.x) * 2 + 1];
        }
        __syncthreads();
        if (k < n) {
            const int cnt = min(256, m - j0);
            for (int j = 0; j < cnt; ++j) {
                // float64 difference, rounded to float32, 2-norm in float32 (torch.norm(dist.float(), dim=-1))
                const float dy = (float)(wy - (double)oth[j * 2]), dx = (float)(wx - (double)oth[j * 2 + 1]);
                const float d = sqrtf(__fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dx, dx)));
                const int c = d <= thr;
                any |= c;
                hit |= c & (j0 + j == mate);
            }
        }
    }
    if (k < n && mate >= 0) tp[(size_t)b * K + k] = (unsigned char)hit;
    const unsigned long long ba = __ballot(any), bh = __ballot(hit && mate >= 0);
    if ((threadIdx.x & 63) == 0) {
        if (ba) atomicAdd(&metrics[p * 8 + 0 + dir], __popcll(ba));        // n_gt
        if (bh) atomicAdd(&metrics[p * 8 + 2 + dir], __popcll(bh));        // num_matched
    }
}

// inverse of the mutual-NN map: inv[p][j] = i where match_idx[p][i] == j
__global__ __launch_bounds__(256) void invert_kernel(const int* __restrict__ match_idx, const int* __restrict__ kp_count,
                                                     int K, int* __restrict__ inv_idx, int* __restrict__ metrics)
{
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = min(kp_count[2 * p], K), m = min(kp_count[2 * p + 1], K);
    int has = 0;
    if (i < n) {
        const int j = match_idx[(size_t)p * K + i];
        if (j >= 0 && j < m) { inv_idx[(size_t)p * K + j] = i; has = 1; }
    }
    const unsigned long long bal = __ballot(has);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&metrics[p * 8 + 6], __popcll(bal));   // number of matches
}

// ---- keypoint repeatability (reference multipoint/utils/evaluation.py:156-199) ---------------------------------------
// warp_keypoints with its default integer return type (homographies.py:331-346): the float64 result of
// cv2.perspectiveTransform is truncated toward zero (ndarray.astype(int)) after EACH of the two warps
// (frame -> world by inv(h_own), world -> other frame by h_other), then filter_points keeps in-image points.
__device__ __forceinline__ void warp_trunc(const double* m, long long& y, long long& x)
{
#pragma clang fp contract(off)
    const double xd = (double)x, yd = (double)y;
    double w = (((xd * m[6]) + (yd * m[7])) + m[8]);
    w = (fabs(w) > 2.220446049250313e-16) ? 1.0 / w : 0.0;
    const double xo = ((((xd * m[0]) + (yd * m[1])) + m[2]) * w);
    const double yo = ((((xd * m[3]) + (yd * m[4])) + m[5]) * w);
    x = (long long)xo; y = (long long)yo;
}

__global__ __launch_bounds__(256) void rep_warp_kernel(const int* __restrict__ kp_yx, const int* __restrict__ kp_count,
                                                       const double* __restrict__ hom, int K, int H, int W,
                                                       long long* __restrict__ warped, int* __restrict__ out)
{
#pragma clang fp contract(off)    // cv2.perspectiveTransform rounds every product and sum separately
    const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    const int n = min(kp_count[b], K);
    int inside = 0;
    if (k < n) {
        long long y = kp_yx[((size_t)b * K + k) * 2], x = kp_yx[((size_t)b * K + k) * 2 + 1];
        warp_trunc(hom + (size_t)b * 18, y, x);
        warp_trunc(hom + (size_t)b * 18 + 9, y, x);
        inside = (y >= 0) & (x >= 0) & (y < H) & (x < W);
        warped[((size_t)b * K + k) * 2] = inside ? y : -1;          // -1: filtered out
        warped[((size_t)b * K + k) * 2 + 1] = x;
    }
    const unsigned long long bal = __ballot(inside);
    // out[p]: count1, count2, N_thermal, N_optical
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&out[(b >> 1) * 4 + ((b & 1) ? 2 : 3)], __popcll(bal));
}

// one thread per warped point of slot b; the other image's keypoints are scanned from LDS
__global__ __launch_bounds__(256) void rep_count_kernel(const int* __restrict__ kp_yx, const int* __restrict__ kp_count,
                                                        const long long* __restrict__ warped, int K, double thr,
                                                        int* __restrict__ out)
{
    __shared__ int oth[256 * 2];
    const int b = blockIdx.y, ob = b ^ 1;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n = min(kp_count[b], K), m = min(kp_count[ob], K);
    long long wy = -1, wx = 0;
    if (k < n) { wy = warped[((size_t)b * K + k) * 2]; wx = warped[((size_t)b * K + k) * 2 + 1]; }
    const bool live = wy >= 0;
    int hit = 0;
    for (int j0 = 0; j0 < m; j0 += 256) {
        __syncthreads();
        if (j0 + threadIdx.x < m) {
            oth[threadIdx.x * 2] = kp_yx[((size_t)ob * K + j0 + threadIdx.x) * 2];
            oth[threadIdx.x * 2 + 1] = kp_yx[((size_t)ob * K + j0 + threadIdx.x) * 2 + 1];
        }
        __syncthreads();
        if (live) {
            const int cnt = min(256, m - j0);
            for (int j = 0; j < cnt; ++j) {
                const long long dy = wy - oth[j * 2], dx = wx - oth[j * 2 + 1];
                hit |= sqrt((double)(dy * dy + dx * dx)) <= thr;          // np.linalg.norm of an int array: float64
            }
        }
    }
    const unsigned long long bal = __ballot(hit);
    // warped THERMAL points near an optical keypoint -> count1; warped OPTICAL points near a thermal one -> count2
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&out[(b >> 1) * 4 + ((b & 1) ? 0 : 1)], __popcll(bal));
}

}  // namespace

void launch_repeatability(const int* kp_yx, const int* kp_count, const double* hom, int P, int K, int H, int W, double thr,
                          long long* warped, int* out, hipStream_t s)
{
    const dim3 g((K + 255) / 256, 2 * P);
    hipLaunchKernelGGL(rep_warp_kernel, g, dim3(256), 0, s, kp_yx, kp_count, hom, K, H, W, warped, out);
    hipLaunchKernelGGL(rep_count_kernel, g, dim3(256), 0, s, kp_yx, kp_count, warped, K, thr, out);
}

void launch_pair_metrics(const int* kp_yx, const int* kp_count, const int* match_idx, const double* hom, int P, int K,
                         int H, int W, float thr, double* warped, int* inv_idx, unsigned char* tp, int* metrics,
                         hipStream_t s)
{
    const dim3 g((K + 255) / 256, 2 * P), gp((K + 255) / 256, P);
    hipLaunchKernelGGL(warp_kernel, g, dim3(256), 0, s, kp_yx, kp_count, hom, K, H, W, warped, metrics);
    hipLaunchKernelGGL(invert_kernel, gp, dim3(256), 0, s, match_idx, kp_count, K, inv_idx, metrics);
    hipLaunchKernelGGL(correct_kernel, g, dim3(256), 0, s, kp_yx, kp_count, warped, match_idx, K, thr, inv_idx, tp, metrics);
}

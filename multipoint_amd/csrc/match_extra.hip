// The remaining matcher modes of utils.get_matches (reference multipoint/utils/matching.py:4-33, :74-99), next to the
// mutual-NN kernels of sample_match.hip:
//
//  * knn2_kernel       cv2.BFMatcher(cv2.NORM_L2).knnMatch(d1, d2, 2) and .match() without crossCheck: for every query
//                      row the two nearest train rows under || a - b ||_2 (float32 sum of squared differences, then
//                      sqrt), ties -> lower train index first (OpenCV inserts a candidate only if it is strictly
//                      closer).  get_matches applies Lowe's ratio test (0.9) to the two on the host list.
//  * threshold_kernel  ThresholdMatcher.match (:81-99): every (i, j) with sqrt(2 - 2 clip(a.b, -1, 1)) < threshold,
//                      appended to a fixed-capacity list (the caller orders it row-major like np.argwhere).
//
// Layout: a 256-thread workgroup owns 32 query rows of one pair; lane group g = tid / 8 is the query, sub = tid % 8
// walks the train rows sub, sub+8, ... of a 64-row tile staged in LDS (row stride D+1 floats: the 8 sub-lanes of a
// query read 8 different rows -> different banks; the 32 queries read the same train element -> broadcast).  The N x M
// matrix is never written.  These modes are off the benchmarked default (bfmatcher + crossCheck); the kernels are
// LDS-bandwidth bound, 2 LDS reads per FMA.
#include "mp_common.h"

namespace {

constexpr int QB = 32, TB = 64, SUB = 8;

struct Top2 { float d0, d1; int i0, i1; };

__device__ __forceinline__ void top2_insert(Top2& t, float d, int j)
{
    // candidates arrive in increasing j per lane; across lanes the merge below orders ties by index
    if (d < t.d0 || (d == t.d0 && j < t.i0)) { t.d1 = t.d0; t.i1 = t.i0; t.d0 = d; t.i0 = j; }
    else if (d < t.d1 || (d == t.d1 && j < t.i1)) { t.d1 = d; t.i1 = j; }
}

template <int MODE>   // 0: knn2 (squared L2), 1: threshold list (dot product)
__global__ __launch_bounds__(256) void rows_kernel(const float* __restrict__ dA, const int* __restrict__ nA,
                                                   const float* __restrict__ dB, const int* __restrict__ nB,
                                                   long long pair_stride, int count_stride, int K, int D, float thr,
                                                   int* __restrict__ out_idx, float* __restrict__ out_dist,
                                                   int capacity, int* __restrict__ list_ij,
                                                   float* __restrict__ list_dist, int* __restrict__ list_count)
{
    extern __shared__ float lds[];
    const int p = blockIdx.y;
    const int N = min(nA[(size_t)p * count_stride], K), M = min(nB[(size_t)p * count_stride], K);
    const int q0 = blockIdx.x * QB;
    if (q0 >= N) return;
    const int LD = D + 1;
    float* qs = lds;                 // [QB][LD]
    float* ts = lds + QB * LD;       // [TB][LD]
    const float* A = dA + (size_t)p * pair_stride;
    const float* B = dB + (size_t)p * pair_stride;
    for (int f = threadIdx.x; f < QB * D; f += 256) {
        const int r = f / D, c = f - r * D;
        qs[r * LD + c] = (q0 + r < N) ? A[(size_t)(q0 + r) * D + c] : 0.f;
    }
    const int g = threadIdx.x / SUB, sub = threadIdx.x % SUB;
    const int qi = q0 + g;
    Top2 best{INFINITY, INFINITY, -1, -1};
    for (int t0 = 0; t0 < M; t0 += TB) {
        __syncthreads();
        for (int f = threadIdx.x; f < TB * D; f += 256) {
            const int r = f / D, c = f - r * D;
            ts[r * LD + c] = (t0 + r < M) ? B[(size_t)(t0 + r) * D + c] : 0.f;
        }
        __syncthreads();
        for (int r = sub; r < TB; r += SUB) {
            const int j = t0 + r;
            if (j >= M || qi >= N) continue;
            const float* a = qs + g * LD;
            const float* b = ts + r * LD;
            float s = 0.f;
            if (MODE == 0) {
                for (int c = 0; c < D; ++c) { const float e = a[c] - b[c]; s += e * e; }
                top2_insert(best, s, j);
            } else {
                for (int c = 0; c < D; ++c) s += a[c] * b[c];
                const float d = sqrtf(2.f - 2.f * fminf(1.f, fmaxf(-1.f, s)));
                if (d < thr) {
                    const int slot = atomicAdd(&list_count[p], 1);
                    if (slot < capacity) {
                        list_ij[((size_t)p * capacity + slot) * 2] = qi;
                        list_ij[((size_t)p * capacity + slot) * 2 + 1] = j;
                        list_dist[(size_t)p * capacity + slot] = d;
                    }
                }
            }
        }
    }
    if (MODE == 0) {
        // merge the 8 sub-lane lists of a query (lanes g*8 .. g*8+7 of the same wave)
        for (int off = 1; off < SUB; off <<= 1) {
            Top2 o;
            o.d0 = __shfl_xor(best.d0, off); o.i0 = __shfl_xor(best.i0, off);
            o.d1 = __shfl_xor(best.d1, off); o.i1 = __shfl_xor(best.i1, off);
            if (o.i0 >= 0) top2_insert(best, o.d0, o.i0);
            if (o.i1 >= 0) top2_insert(best, o.d1, o.i1);
        }
        if (sub == 0 && qi < N) {
            const size_t o = ((size_t)p * K + qi) * 2;
            out_idx[o] = best.i0; out_idx[o + 1] = best.i1;
            out_dist[o] = best.i0 >= 0 ? sqrtf(best.d0) : 0.f;
            out_dist[o + 1] = best.i1 >= 0 ? sqrtf(best.d1) : 0.f;
        }
    }
}

}  // namespace

static size_t rows_lds_bytes(int D) { return (size_t)(QB + TB) * (D + 1) * sizeof(float); }

void launch_match_knn2(const float* dA, const int* nA, const float* dB, const int* nB, long long pair_stride,
                       int count_stride, int P, int K, int D, int* idx, float* dist, hipStream_t s)
{
    const dim3 g((K + QB - 1) / QB, P);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rows_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)rows_lds_bytes(D));
    rows_kernel<0><<<g, 256, rows_lds_bytes(D), s>>>(dA, nA, dB, nB, pair_stride, count_stride, K, D, 0.f, idx, dist, 0,
                                                     nullptr, nullptr, nullptr);
}

void launch_match_threshold(const float* dA, const int* nA, const float* dB, const int* nB, long long pair_stride,
                            int count_stride, int P, int K, int D, float thr, int capacity, int* list_ij,
                            float* list_dist, int* list_count, hipStream_t s)
{
    const dim3 g((K + QB - 1) / QB, P);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rows_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)rows_lds_bytes(D));
    rows_kernel<1><<<g, 256, rows_lds_bytes(D), s>>>(dA, nA, dB, nB, pair_stride, count_stride, K, D, thr, nullptr,
                                                     nullptr, capacity, list_ij, list_dist, list_count);
}

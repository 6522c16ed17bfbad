// Keypoint list extraction: deterministic ordered stream compaction + exact top-k.
//
// Replaces, from the reference:
//   * per-image top-k of box_nms           multipoint/utils/utils.py:109-116
//     ("first k kept indices in descending score order"; ties: lower row-major index first)
//   * scatter into zeros_like(prob)        multipoint/utils/utils.py:119-120
//   * torch.nonzero(prob > thr)            predict_align_image_pair.py:170-171, evaluation.py:262-263
//     -> (N,2) (y,x) rows in row-major order
// One 1024-thread workgroup per image: ballot/shuffle prefix scans give every kept pixel its
// row-major rank without atomics (so the output order is deterministic and equals nonzero()),
// and a 4-pass 8-bit radix select over the fp32 score bits finds the exact k-th score; ties at
// the threshold are admitted in row-major order.
#include "mp_common.h"

namespace {

constexpr int BT = 1024;                      // threads per workgroup (16 waves)

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(v, off);
        if (lane >= off) v += n;
    }
    return v;
}

// exclusive prefix of `c` over the workgroup in thread order; returns block total via *total
__device__ __forceinline__ int block_excl_scan(int c, int* s_wave, int* total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int inc = wave_incl_scan(c, lane);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BT / 64; ++i) {
        const int x = s_wave[i];
        if (i < w) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - c;
}

// MODE 0: kept pixels of an NMS work map (v < 0, score = -v) -> list
// MODE 1: v > thr -> direct keypoint output (torch.nonzero semantics)
template <int MODE>
__device__ int compact_image(const float* __restrict__ img, int n, float thr, int W, int cap,
                             int* __restrict__ o_idx, float* __restrict__ o_score,
                             int* __restrict__ o_yx, int* s_wave)
{
    int base = 0;
    for (int start = 0; start < n; start += BT * 4) {
        const int i = start + threadIdx.x * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < n) v = *reinterpret_cast<const f32x4*>(img + i);       // n % 4 == 0
        bool k[4];
        int c = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { k[e] = MODE == 0 ? (v[e] < 0.f) : (v[e] > thr); c += k[e]; }
        int tot;
        int pos = base + block_excl_scan(c, s_wave, &tot);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k[e]) {
                if (pos < cap) {
                    const int idx = i + e;
                    if (MODE == 0) { o_idx[pos] = idx; o_score[pos] = -v[e]; }
                    else {
                        o_yx[2 * pos] = idx / W; o_yx[2 * pos + 1] = idx % W;
                        if (o_score) o_score[pos] = v[e];
                    }
                }
                ++pos;
            }
        base += tot;
    }
    return base;
}

__global__ __launch_bounds__(BT) void select_keypoints_kernel(
    const float* __restrict__ work, int H, int W, int topk, int K, int* __restrict__ list_idx,
    float* __restrict__ list_score, int list_cap, int* __restrict__ kp_yx, float* __restrict__ kp_score,
    int* __restrict__ kp_count, float* __restrict__ prob_nms)
{
    __shared__ int s_wave[BT / 64];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_sel[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = H * W;
    const float* img = work + (long long)b * n;
    int* lidx = list_idx + (long long)b * list_cap;
    float* lsc = list_score + (long long)b * list_cap;

    if (prob_nms) {   // zeros_like(prob), utils.py:119
        f32x4* o = reinterpret_cast<f32x4*>(prob_nms + (long long)b * n);
        for (int i = tid; i < n / 4; i += BT) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    int nk = compact_image<0>(img, n, 0.f, W, list_cap, lidx, lsc, nullptr, s_wave);
    nk = min(nk, list_cap);
    __syncthreads();

    const int k = (topk > 0) ? topk : 0x7fffffff;      // list outputs are clipped to K below
    unsigned T = 0;          // admit score bits > T, plus the first `need` entries == T
    int need = 0;
    const bool select = nk > k;
    if (select) {
        unsigned prefix = 0, mask = 0;
        int kk = k;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < nk; i += BT) {
                const unsigned key = __float_as_uint(lsc[i]);
                if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned cum = 0;
                int d = 255;
                for (; d > 0; --d) {
                    if (cum + s_hist[d] >= (unsigned)kk) break;
                    cum += s_hist[d];
                }
                s_sel[0] = (unsigned)d;
                s_sel[1] = cum;
            }
            __syncthreads();
            prefix |= s_sel[0] << shift;
            mask |= 255u << shift;
            kk -= (int)s_sel[1];
            __syncthreads();
        }
        T = prefix;
        need = kk;
    }

    // ordered output (row-major, == torch.nonzero order of the NMS'ed map)
    int out_base = 0, tie_base = 0;
    for (int start = 0; start < nk; start += BT) {
        const int i = start + tid;
        unsigned key = 0;
        int idx = 0;
        float sc = 0.f;
        if (i < nk) { sc = lsc[i]; key = __float_as_uint(sc); idx = lidx[i]; }
        bool sel = i < nk;
        if (select) {
            const int eq = (i < nk) && (key == T);
            int tot_eq;
            const int tie_rank = tie_base + block_excl_scan(eq, s_wave, &tot_eq);
            tie_base += tot_eq;
            sel = (i < nk) && ((key > T) || (eq && tie_rank < need));
        }
        int tot;
        const int pos = out_base + block_excl_scan(sel ? 1 : 0, s_wave, &tot);
        out_base += tot;
        if (sel && pos < K) {
            if (kp_yx) {
                kp_yx[((long long)b * K + pos) * 2] = idx / W;
                kp_yx[((long long)b * K + pos) * 2 + 1] = idx % W;
            }
            if (kp_score) kp_score[(long long)b * K + pos] = sc;
        }
        if (sel && prob_nms) prob_nms[(long long)b * n + idx] = sc;        // utils.py:120
    }
    if (tid == 0 && kp_count) kp_count[b] = out_base;      // may exceed K when topk == 0: overflow
}

__global__ __launch_bounds__(BT) void extract_threshold_kernel(const float* __restrict__ map, int H,
                                                              int W, float thr, int K,
                                                              int* __restrict__ kp_yx,
                                                              float* __restrict__ kp_score,
                                                              int* __restrict__ kp_count)
{
    __shared__ int s_wave[BT / 64];
    const int b = blockIdx.x;
    const int n = H * W;
    const int nk = compact_image<1>(map + (long long)b * n, n, thr, W, K, nullptr,
                                    kp_score ? kp_score + (long long)b * K : nullptr,
                                    kp_yx + (long long)b * K * 2, s_wave);
    if (threadIdx.x == 0) kp_count[b] = nk;
}

}  // namespace

void launch_select_keypoints(const float* work, int B, int H, int W, int topk, int K, int* list_idx,
                             float* list_score, int list_cap, int* kp_yx, float* kp_score,
                             int* kp_count, float* prob_nms, hipStream_t s)
{
    if (B <= 0) return;
    hipLaunchKernelGGL(select_keypoints_kernel, dim3(B), dim3(BT), 0, s, work, H, W, topk, K, list_idx,
                       list_score, list_cap, kp_yx, kp_score, kp_count, prob_nms);
}

void launch_extract_threshold(const float* map, int B, int H, int W, float thr, int K, int* kp_yx,
                              float* kp_score, int* kp_count, hipStream_t s)
{
    if (B <= 0) return;
    hipLaunchKernelGGL(extract_threshold_kernel, dim3(B), dim3(BT), 0, s, map, H, W, thr, K, kp_yx,
                       kp_score, kp_count);
}

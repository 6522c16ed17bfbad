// Single-image detector metrics on the GPU: the per-sample arithmetic of utils.compute_detector_metrics /
// compute_tp_fp_dist (reference multipoint/utils/evaluation.py:10-97), the `-e` mode of predict_keypoints.py for
// single-image datasets (predict_keypoints.py:88-104).
//
//   predictions  = pixels with prob > zero_threshold, ranked by prob descending            (:67-72)
//   ground truth = nonzero pixels of the label map, in row-major order                      (:65)
//   matches[p][g] = || pred_p - gt_g ||_2 <= distance_thresh  (float32 norm of integer offsets: exact)   (:80-82)
//   greedy loop over the ranked predictions (:84-93): gt_idx = FIRST ground-truth point (row-major) matching the
//   prediction; the prediction is a true positive iff that point has not been claimed by a better-ranked one.
//
// The loop has a closed form: every prediction names exactly one ground-truth point g(p) (or none), and the true
// positive of a ground-truth point is the best-ranked prediction naming it.  One thread per pixel: scan the
// (2R+1)^2 window of the label map in row-major order (R = floor(distance_thresh) <= 2), push a packed
// (prob bits, ~flat index) key to the named point with a 64-bit atomicMax, then a second pass compares keys.
// Stated tie-break for equal probabilities (torch.sort leaves it open): lower flat index ranks first.
//
// Records: one per prediction, appended per image with a wave-aggregated atomic (order is irrelevant: the caller
// ranks them): flat index, prob, bits [0..24] = window positions (dy+2)*5+(dx+2) whose ground-truth point matches
// (the entries of dist[matches], :97), bit 31 = true positive.
#include "mp_common.h"

namespace {

__device__ __forceinline__ unsigned long long det_key(float prob, int flat)
{
    return ((unsigned long long)__float_as_uint(prob) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)flat);
}

// window scan shared by both passes: returns the match bits and the flat index of the first matching point (-1: none)
__device__ __forceinline__ unsigned scan_window(const unsigned char* __restrict__ gt, int y, int x, int H, int W,
                                                int R, float thr, int& first)
{
    unsigned bits = 0;
    first = -1;
    for (int dy = -R; dy <= R; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -R; dx <= R; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            if (sqrtf((float)(dy * dy + dx * dx)) > thr) continue;      // torch.norm(diff.float()) <= thresh
            if (gt[(size_t)yy * W + xx]) {
                bits |= 1u << ((dy + 2) * 5 + (dx + 2));
                if (first < 0) first = yy * W + xx;
            }
        }
    }
    return bits;
}

__global__ __launch_bounds__(256) void det_claim_kernel(const float* __restrict__ prob,
                                                        const unsigned char* __restrict__ gt, int H, int W,
                                                        float zero_thr, int R, float thr,
                                                        unsigned long long* __restrict__ best, int* __restrict__ n_gt)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    const bool in = x < W && y < H;
    const size_t base = (size_t)b * H * W;
    int is_gt = 0;
    if (in) {
        const int flat = y * W + x;
        is_gt = gt[base + flat] != 0;
        const float p = prob[base + flat];
        if (p > zero_thr) {
            int first;
            scan_window(gt + base, y, x, H, W, R, thr, first);
            if (first >= 0) atomicMax(&best[base + first], det_key(p, flat));
        }
    }
    const unsigned long long bal = __ballot(is_gt);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&n_gt[b], __popcll(bal));
}

__global__ __launch_bounds__(256) void det_record_kernel(const float* __restrict__ prob,
                                                         const unsigned char* __restrict__ gt, int H, int W,
                                                         float zero_thr, int R, float thr,
                                                         const unsigned long long* __restrict__ best,
                                                         int* __restrict__ rec_index, float* __restrict__ rec_prob,
                                                         unsigned* __restrict__ rec_bits, int* __restrict__ rec_count)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    const size_t base = (size_t)b * H * W;
    bool pred = false;
    float p = 0.f;
    unsigned bits = 0;
    const int flat = y * W + x;
    if (x < W && y < H) {
        p = prob[base + flat];
        if (p > zero_thr) {
            pred = true;
            int first;
            bits = scan_window(gt + base, y, x, H, W, R, thr, first);
            if (first >= 0 && best[base + first] == det_key(p, flat)) bits |= 0x80000000u;
        }
    }
    // wave-aggregated append
    const unsigned long long bal = __ballot(pred);
    if (!bal) return;
    const int lane = threadIdx.x & 63;
    int slot0 = 0;
    if (lane == __ffsll((long long)bal) - 1) slot0 = atomicAdd(&rec_count[b], __popcll(bal));
    slot0 = __shfl(slot0, __ffsll((long long)bal) - 1);
    if (pred) {
        const size_t slot = base + slot0 + __popcll(bal & ((1ull << lane) - 1ull));
        rec_index[slot] = flat;
        rec_prob[slot] = p;
        rec_bits[slot] = bits;
    }
}

}  // namespace

void launch_detector_metrics(const float* prob, const unsigned char* gt, int B, int H, int W, float zero_thr,
                             float distance_thr, unsigned long long* best, int* rec_index, float* rec_prob,
                             unsigned* rec_bits, int* rec_count, int* n_gt, hipStream_t s)
{
    const int R = (int)floorf(distance_thr);
    const dim3 g((W + 63) / 64, (H + 3) / 4, B);
    det_claim_kernel<<<g, 256, 0, s>>>(prob, gt, H, W, zero_thr, R, distance_thr, best, n_gt);
    det_record_kernel<<<g, 256, 0, s>>>(prob, gt, H, W, zero_thr, R, distance_thr, best, rec_index, rec_prob, rec_bits, rec_count);
}

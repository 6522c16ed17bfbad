// Keypoint list extraction: deterministic ordered stream compaction + exact top-k.
//
// Replaces, from the reference:
//   * per-image top-k of box_nms           multipoint/utils/utils.py:109-116
//     ("first k kept indices in descending score order"; ties: lower row-major index first)
//   * scatter into zeros_like(prob)        multipoint/utils/utils.py:119-120
//   * torch.nonzero(prob > thr)            predict_align_image_pair.py:170-171, evaluation.py:262-263
//     -> (N,2) (y,x) rows in row-major order
// Ballot/shuffle prefix scans give every kept pixel its row-major rank without atomics (so the output order is
// deterministic and equals nonzero()).  The dense map is compacted by MANY workgroups per image in two passes over
// 16 Ki-pixel segments -- count, then ordered write at the segment's prefix (a single workgroup per image streams a
// 1024x1280 map at only ~260 GB/s) -- and one 1024-thread workgroup per image then works on the short list only:
// a 4-pass 8-bit radix select over the fp32 score bits finds the exact k-th score; ties at the threshold are
// admitted in row-major order.
#include "mp_common.h"

namespace {

// 256 threads and < 64 VGPRs.  (Sized in round 1 to fit on a CU next to a resident F(2x2,3x3) convolution workgroup; the
// F(4x4,3x3) workgroups of rounds 2 / 3 own their CU -- 512 threads x 231-256 registers, 158 KiB of LDS -- so these kernels now
// run at launch boundaries and beside the head tail)
constexpr int BT = 256;                       // threads per workgroup (4 waves)

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(v, off);
        if (lane >= off) v += n;
    }
    return v;
}

// exclusive prefix of `c` over a workgroup of NT threads in thread order; returns block total via *total
template <int NT = BT>
__device__ __forceinline__ int block_excl_scan(int c, int* s_wave, int* total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int inc = wave_incl_scan(c, lane);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) {
        const int x = s_wave[i];
        if (i < w) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - c;
}

constexpr int BTS = 1024;                     // threads of the per-image selection workgroup

constexpr int SEG = 16384;                    // pixels per segment (a multiple of BT * 4)

// MODE 0: kept pixels of an NMS work map (v < 0, score = -v) -> list
// MODE 1: v > thr -> direct keypoint output (torch.nonzero semantics)
template <int MODE>
__device__ __forceinline__ bool keep_px(float v, float thr) { return MODE == 0 ? (v < 0.f) : (v > thr); }

// pass 1: survivors per segment (and zeros_like(prob) for the dense NMS output, utils.py:119)
// VEC: n % 4 == 0 (16-byte loads); otherwise (mp_extract_keypoints on a map whose pixel count is no multiple of 4) four scalar loads with bounds
template <int MODE, bool VEC = true>
__global__ __launch_bounds__(BT) void count_segments_kernel(const float* __restrict__ map, int n, float thr, int nseg,
                                                            int* __restrict__ seg_count, float* __restrict__ zero_fill,
                                                            const unsigned char* __restrict__ mask)
{
    __shared__ int s_wave[BT / 64];
    const int b = blockIdx.y, sg = blockIdx.x;
    const float* img = map + (long long)b * n;
    const int lo = sg * SEG, hi = min(n, lo + SEG);
    int c = 0;
    for (int i = lo + threadIdx.x * 4; i < hi; i += BT * 4) {
        f32x4 v;
        uchar4 m = make_uchar4(1, 1, 1, 1);
        if constexpr (VEC) {
            v = *reinterpret_cast<const f32x4*>(img + i);          // n % 4 == 0
            if (mask) m = *reinterpret_cast<const uchar4*>(mask + (long long)b * n + i);
        } else {
            const unsigned char* mk1 = mask ? mask + (long long)b * n : nullptr;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (i + e < hi) ? img[i + e] : (MODE == 0 ? 0.f : -__builtin_inff());      // beyond the map: never kept
            if (mk1) m = make_uchar4(i < hi ? mk1[i] : 0, i + 1 < hi ? mk1[i + 1] : 0, i + 2 < hi ? mk1[i + 2] : 0, i + 3 < hi ? mk1[i + 3] : 0);
        }
        const bool mk[4] = {m.x != 0, m.y != 0, m.z != 0, m.w != 0};      // (prob > thr) * valid_mask, evaluation.py:156-157
#pragma unroll
        for (int e = 0; e < 4; ++e) c += keep_px<MODE>(v[e], thr) && mk[e];
        if (zero_fill) *reinterpret_cast<f32x4*>(zero_fill + (long long)b * n + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int tot;
    block_excl_scan(c, s_wave, &tot);
    if (threadIdx.x == 0) seg_count[b * nseg + sg] = tot;
}

// pass 2: ordered write of the segment's survivors at the prefix of the segments before it
template <int MODE, bool VEC = true>
__global__ __launch_bounds__(BT) void compact_segments_kernel(const float* __restrict__ map, int n, float thr, int W,
                                                              int nseg, const int* __restrict__ seg_count, int cap,
                                                              long long out_stride, int* __restrict__ o_idx,
                                                              float* __restrict__ o_score, int* __restrict__ o_yx,
                                                              int* __restrict__ total, const unsigned char* __restrict__ mask)
{
    __shared__ int s_wave[BT / 64];
    const int b = blockIdx.y, sg = blockIdx.x;
    const float* img = map + (long long)b * n;
    int base = 0;
    for (int i = 0; i < sg; ++i) base += seg_count[b * nseg + i];           // nseg <= ~100: uniform, cached
    if (total && sg == nseg - 1 && threadIdx.x == 0) total[b] = base + seg_count[b * nseg + sg];
    if (seg_count[b * nseg + sg] == 0) return;
    if (MODE == 0) { o_idx += (long long)b * out_stride; o_score += (long long)b * out_stride; }
    else { o_yx += (long long)b * out_stride * 2; if (o_score) o_score += (long long)b * out_stride; }
    const int lo = sg * SEG, hi = min(n, lo + SEG);
    for (int start = lo; start < hi; start += BT * 4) {
        const int i = start + threadIdx.x * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        uchar4 m = make_uchar4(1, 1, 1, 1);
        if (i < hi) {
            if constexpr (VEC) {
                v = *reinterpret_cast<const f32x4*>(img + i);
                if (mask) m = *reinterpret_cast<const uchar4*>(mask + (long long)b * n + i);
            } else {
                const unsigned char* mk1 = mask ? mask + (long long)b * n : nullptr;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (i + e < hi) ? img[i + e] : (MODE == 0 ? 0.f : -__builtin_inff());
                if (mk1) m = make_uchar4(mk1[i], i + 1 < hi ? mk1[i + 1] : 0, i + 2 < hi ? mk1[i + 2] : 0, i + 3 < hi ? mk1[i + 3] : 0);
            }
        }
        const bool mk[4] = {m.x != 0, m.y != 0, m.z != 0, m.w != 0};
        bool k[4];
        int c = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { k[e] = (i < hi) && keep_px<MODE>(v[e], thr) && mk[e]; c += k[e]; }
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
}

__global__ __launch_bounds__(BTS) void select_keypoints_kernel(
    const float* __restrict__ work, int H, int W, int topk, int K, int* __restrict__ list_idx,
    float* __restrict__ list_score, int list_cap, int* __restrict__ kp_yx, float* __restrict__ kp_score,
    int* __restrict__ kp_count, float* __restrict__ prob_nms, const int* __restrict__ list_count,
    float tie_eps, int tie_min, int* __restrict__ tie_state, int Wout, int* __restrict__ tie_pairs, int pairs_min)
{
    __shared__ int s_wave[BTS / 64];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_sel[2];
    __shared__ int s_near[2];          // survivors within tie_eps of the k-th score: [0] admitted, [1] cut off
    if (threadIdx.x < 2) s_near[threadIdx.x] = 0;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = H * W;
    const float* img = work + (long long)b * n;
    int* lidx = list_idx + (long long)b * list_cap;
    float* lsc = list_score + (long long)b * list_cap;

    // the list (row-major survivors of image b) was written by compact_segments_kernel<0>; prob_nms is already zero
    const int nk = min(list_count[b], list_cap);

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
            for (int i = tid; i < nk; i += BTS) {
                const unsigned key = __float_as_uint(lsc[i]);
                if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            // the digit d with (#keys of a larger digit) < kk <= (#keys of digit >= d), found by all 256 threads at once (thread t
            // holds bin t; suffix count = total - exclusive prefix): a single thread walking the bins down took a dependent LDS
            // read per bin, 256 x 4 passes = half of the kernel's 66 us
            {
                const int hcnt = tid < 256 ? (int)s_hist[tid] : 0;
                int tot;
                const int before = block_excl_scan<BTS>(hcnt, s_wave, &tot);      // bins 0 .. t-1
                const int ge = tot - before;                                 // bins t .. 255
                if (tid == 0) { s_sel[0] = 0u; s_sel[1] = (unsigned)(tot - hcnt); }      // (fewer than kk keys: digit 0, as the walk did)
                __syncthreads();
                if (tid > 0 && tid < 256 && ge >= kk && ge - hcnt < kk) { s_sel[0] = (unsigned)tid; s_sel[1] = (unsigned)(ge - hcnt); }
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
    for (int start = 0; start < nk; start += BTS) {
        const int i = start + tid;
        unsigned key = 0;
        int idx = 0;
        float sc = 0.f;
        if (i < nk) { sc = lsc[i]; key = __float_as_uint(sc); idx = lidx[i]; }
        bool sel = i < nk;
        if (select) {
            const int eq = (i < nk) && (key == T);
            int tie_rank = 0;
            if (__syncthreads_or(eq)) {                 // (scores AT the k-th one are rare: most chunks skip their rank scan)
                int tot_eq;
                tie_rank = tie_base + block_excl_scan<BTS>(eq, s_wave, &tot_eq);
                tie_base += tot_eq;
            }
            sel = (i < nk) && ((key > T) || (eq && tie_rank < need));
            // top-k tie guard: how many survivors sit within the convolution's rounding noise of the cut, on either side of it
            // (a cut inside a plateau of (near-)tied scores picks members of the plateau by that noise; a few hits per image)
            if (tie_state && i < nk && fabsf(sc - __uint_as_float(T)) <= tie_eps) atomicAdd(&s_near[sel ? 0 : 1], 1);
        }
        int tot;
        const int pos = out_base + block_excl_scan<BTS>(sel ? 1 : 0, s_wave, &tot);
        out_base += tot;
        if (sel && pos < K) {
            if (kp_yx) {
                kp_yx[((long long)b * K + pos) * 2] = idx / W;
                kp_yx[((long long)b * K + pos) * 2 + 1] = idx % W;
            }
            if (kp_score) kp_score[(long long)b * K + pos] = sc;
        }
        // utils.py:120 (Wout: the dense output's row stride -- the caller's width; W is that rounded up to a multiple of 4)
        if (sel && prob_nms) prob_nms[(long long)b * H * Wout + (long long)(idx / W) * Wout + idx % W] = sc;
    }
    if (tid == 0 && kp_count) kp_count[b] = out_base;      // may exceed K when topk == 0: overflow
    if (tie_state) {
        __syncthreads();
        if (tid == 0) {
            // (1) the cut lies inside a plateau of near-tied scores (>= tie_min survivors within tie_eps on EITHER side of it);
            // (2) the cut splits a run of EXACTLY equal scores (tie_base of them, `need` admitted): whatever their number -- lowest index
            //     first is the reference's rule only if the reference's map holds the same tie;
            // (3) footprint tie guard: >= pairs_min NMS decisions of this image were taken between scores within tie_eps (nms.hip), AND they
            //     are >= 1 % of its survivors -- maps of independent scores hold ~1 such pair per 1000 survivors whatever the frame size
            //     (0-5 at 480x640), a plateau inside a footprint several per survivor
            int flag = tie_min > 0 && select && ((s_near[0] >= tie_min && s_near[1] >= tie_min) || (need > 0 && tie_base > need));
            if (tie_pairs) {
                if (pairs_min > 0 && tie_pairs[b] >= pairs_min && (long long)tie_pairs[b] * 100 >= nk) flag = 1;
                tie_pairs[b] = 0;
            }
            if (b < MP_TIE_MAX_IMAGES) tie_state[1 + b] = flag;
            if (flag) atomicAdd(&tie_state[0], 1);
        }
    }
}

}  // namespace

// scratch layout behind `seg_scratch`: [B * nseg] segment counts, then [B] list totals
void launch_select_keypoints(const float* work, int B, int H, int W, int topk, int K, int* list_idx,
                             float* list_score, int list_cap, int* kp_yx, float* kp_score,
                             int* kp_count, float* prob_nms, int* seg_scratch, hipStream_t s, float tie_eps, int tie_min,
                             int* tie_state, int Wout, int* tie_pairs, int pairs_min)
{
    if (B <= 0) return;
    if (Wout <= 0) Wout = W;
    const int n = H * W, nseg = (n + SEG - 1) / SEG;
    int* seg_count = seg_scratch;
    int* list_count = seg_scratch + (size_t)B * nseg;
    const dim3 g(nseg, B);
    // zeros_like(prob) (utils.py:119): fused into the counting pass when the dense output has the work map's geometry
    if (prob_nms && Wout != W) (void)hipMemsetAsync(prob_nms, 0, (size_t)B * H * Wout * sizeof(float), s);
    hipLaunchKernelGGL(count_segments_kernel<0>, g, dim3(BT), 0, s, work, n, 0.f, nseg, seg_count, Wout == W ? prob_nms : (float*)nullptr,
                       (const unsigned char*)nullptr);
    hipLaunchKernelGGL(compact_segments_kernel<0>, g, dim3(BT), 0, s, work, n, 0.f, W, nseg, seg_count, list_cap,
                       (long long)list_cap, list_idx, list_score, (int*)nullptr, list_count, (const unsigned char*)nullptr);
    hipLaunchKernelGGL(select_keypoints_kernel, dim3(B), dim3(BTS), 0, s, work, H, W, topk, K, list_idx,
                       list_score, list_cap, kp_yx, kp_score, kp_count, prob_nms, list_count, tie_eps, tie_min, tie_state, Wout,
                       tie_pairs, pairs_min);
}

void launch_extract_threshold(const float* map, const unsigned char* mask, int B, int H, int W, float thr, int K, int* kp_yx,
                              float* kp_score, int* kp_count, int* seg_scratch, hipStream_t s)
{
    if (B <= 0) return;
    const int n = H * W, nseg = (n + SEG - 1) / SEG;
    const dim3 g(nseg, B);
    if (n % 4 == 0) {
        hipLaunchKernelGGL(count_segments_kernel<1>, g, dim3(BT), 0, s, map, n, thr, nseg, seg_scratch, (float*)nullptr, mask);
        hipLaunchKernelGGL(compact_segments_kernel<1>, g, dim3(BT), 0, s, map, n, thr, W, nseg, seg_scratch, K, (long long)K,
                           (int*)nullptr, kp_score, kp_yx, kp_count, mask);
    } else {          // any H x W (torch.nonzero takes any map): scalar loads
        hipLaunchKernelGGL((count_segments_kernel<1, false>), g, dim3(BT), 0, s, map, n, thr, nseg, seg_scratch, (float*)nullptr, mask);
        hipLaunchKernelGGL((compact_segments_kernel<1, false>), g, dim3(BT), 0, s, map, n, thr, W, nseg, seg_scratch, K, (long long)K,
                           (int*)nullptr, kp_score, kp_yx, kp_count, mask);
    }
}

// ints of scratch the two launchers above need
size_t keypoint_scratch_ints(int B, int H, int W) { return (size_t)B * ((H * W + SEG - 1) / SEG) + (size_t)B; }

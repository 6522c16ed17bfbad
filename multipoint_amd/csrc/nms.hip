// Box non-maximum suppression on the detector heat map.
//
// Replaces utils.box_nms (reference multipoint/utils/utils.py:78-122), i.e. nonzero(prob > min_prob)
// -> square boxes of side `size` centred on each candidate pixel -> torchvision greedy NMS
// (iou threshold) -> per-image top-k -> scatter into a dense map.
//
// Greedy NMS over a fixed priority order (score descending, row-major index ascending -- the
// stable descending sort torchvision applies to candidates listed by nonzero()) has a unique
// answer: a candidate is KEPT iff no kept candidate of higher priority overlaps it with IoU > thr.
// Because every box has the same size, "IoU > thr" depends only on the pixel offset (dy,dx): a
// translation-invariant footprint that the host evaluates once with torchvision's fp32 formula.
// The kernels below run the monotone fixed-point iteration
//     undecided -> dead   if some footprint neighbour is kept
//     undecided -> kept   if every higher-priority footprint neighbour is dead
// Both transitions are sound under arbitrarily stale neighbour reads (a kept neighbour always has
// higher priority than an undecided pixel it overlaps), so the result is identical to the
// sequential greedy algorithm regardless of scheduling; only the number of rounds varies.
// State lives in one fp32 map: >0 undecided (score), 0 dead / never a candidate, <0 kept (-score).
#include "mp_common.h"

namespace {

constexpr int NT = 32;                     // tile side handled by one workgroup

__global__ __launch_bounds__(256) void nms_init_kernel(const float* __restrict__ prob,
                                                      const uint8_t* __restrict__ mask, float min_prob,
                                                      float* __restrict__ work, long long n4)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(prob)[i];
    if (mask) {
        const uchar4 m = reinterpret_cast<const uchar4*>(mask)[i];
        // prob * valid_mask (predict_align_image_pair.py:128, evaluation.py:231-232)
        v[0] *= m.x ? 1.f : 0.f; v[1] *= m.y ? 1.f : 0.f; v[2] *= m.z ? 1.f : 0.f; v[3] *= m.w ? 1.f : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] > min_prob) ? v[e] : 0.f;     // utils.py:97
    reinterpret_cast<f32x4*>(work)[i] = v;
}

// flags: [2][B*tiles] ping-pong: the tile's count of undecided pixels after the round (0: the next round skips the tile).  Their
// sum is what the host asks for (nms_accumulate_kernel) -- NOT a counter every workgroup adds to: 19 200 atomics on one address
// from eight XCDs take 180 us, which was most of round 0's 216 us.
// Work inside a tile is list driven: the undecided pixels are kept as a compact LDS list, one thread
// per list entry, so an iteration costs O(#undecided) instead of O(tile) -- candidates are sparse
// (a few % of the pixels) and most of them are decided after two or three iterations.
// RT > 0: footprint radius known at compile time (fully unrolled scan, row masks in SGPRs).
// INIT (round 0 only): the tile is read from the probability map itself -- prob * valid_mask, thresholded (utils.py:97;
// what nms_init_kernel writes) -- instead of from a work map a separate launch would have to write and this one re-read.
// LOOP (rounds >= 1: on detector maps round 1 still has work in a third of the tiles, rounds 3-7 find nothing to do): a grid of at
// most 2048 workgroups walks the tiles instead of one workgroup per tile -- a round whose flags are all zero then costs 2048
// workgroups x 10 flag reads instead of the dispatch of 19 200 workgroups that each need 13.5 KiB of LDS on a CU before they can
// read their flag and leave (8 us per such round, five of them per batch, each taking a launch-boundary slot away from the next
// batch's convolutions).  Measured in the pipeline (same box, 3 alternating runs of 40 steps): 9.65 / 9.65 / 9.69 ms per step
// against 9.77 / 9.76 / 9.86 with one workgroup per tile in every round; grids of 256 / 1024 / 2048 / 4096 from round 1 or 2:
// 9.79-9.86 / 9.65-9.68 / 9.63 / 9.66-9.68; forward-only on that box 9.60 -- the post-processing that remains visible is 0.03 ms.
template <int RT, bool INIT, bool LOOP = false>
__global__ __launch_bounds__(256) void nms_round_kernel(float* __restrict__ work, int H, int W,
                                                       int tiles_x, int tiles_y, NmsFootprint fp,
                                                       int* __restrict__ flags, int ntiles_total,
                                                       int round,
                                                       const float* __restrict__ prob, const uint8_t* __restrict__ mask,
                                                       float min_prob, int Ws, float tie_eps, int* __restrict__ tie_pairs)
{
    // Ws (INIT only): row stride = true width of prob / mask; the work map's W is Ws rounded up to a multiple of 4 (api.hip), the
    // columns beyond Ws are never candidates.
    // tie_pairs (footprint tie guard, optional): per image, the candidates that die to a kept neighbour whose score is within
    // tie_eps of their own and to no kept neighbour with a clear margin -- decisions the convolution's rounding noise could flip
    constexpr int TR = RT > 0 ? RT : MP_NMS_MAX_R;
    __shared__ float t[(NT + 2 * TR) * (NT + 2 * TR)];
    __shared__ unsigned short list[2][NT * NT];
    __shared__ int cnt[2];
    const int R = RT > 0 ? RT : fp.R;
    const int LW = NT + 2 * R;
    const int tid = threadIdx.x;
    const int* fin = flags + (round & 1) * ntiles_total;
    int* fout = flags + ((round + 1) & 1) * ntiles_total;
    for (int tile_id = blockIdx.x; tile_id < ntiles_total; tile_id += LOOP ? (int)gridDim.x : ntiles_total) {
    if (!INIT && round > 0 && fin[tile_id] == 0) {
        if (tid == 0) fout[tile_id] = 0;
        continue;
    }
    int tt = tile_id;
    const int tx = tt % tiles_x; tt /= tiles_x;
    const int ty = tt % tiles_y;
    const int b = tt / tiles_y;
    const int y0 = ty * NT, x0 = tx * NT;
    float* img = work + (long long)b * H * W;

    if (tid < 2) cnt[tid] = 0;
    __syncthreads();
    auto place = [&](int f, float v) __attribute__((always_inline)) {
        const int ly = f / LW, lx = f - ly * LW;
        t[f] = v;
        if (v > 0.f && ly >= R && ly < R + NT && lx >= R && lx < R + NT)
            list[0][atomicAdd(&cnt[0], 1)] = (unsigned short)f;
    };
    auto fetch = [&](int f) __attribute__((always_inline)) -> float {
        const int ly = f / LW, lx = f - ly * LW;
        const int gy = y0 + ly - R, gx = x0 + lx - R;
        float v = 0.f;
        if constexpr (INIT) {
            if (gy >= 0 && gy < H && gx >= 0 && gx < Ws) {
                const long long gi = (long long)b * H * Ws + (long long)gy * Ws + gx;
                v = prob[gi];
                if (mask) v *= mask[gi] ? 1.f : 0.f;            // prob * valid_mask (predict_align_image_pair.py:128)
                v = (v > min_prob) ? v : 0.f;                   // utils.py:97
            }
        } else if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            v = work[(long long)b * H * W + (long long)gy * W + gx];
        }
        return v;
    };
    if constexpr (RT > 0) {
        // the tile + halo as 16-byte groups: rows of 40 floats from the aligned column x0 - 4 (W is a multiple of 4: a group is
        // inside the image or outside as a whole), (NT + 2 RT) x 10 groups = 1.5 per thread, ALL loads issued before the first
        // value is used (the kernel is latency-bound: one trip to memory per workgroup, a quarter of the 4-byte form's requests)
        constexpr int LWC = NT + 2 * RT, NG = LWC * 10, NL = (NG + 255) / 256;
        f32x4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * 256;
            const int row = idx / 10, g = idx - row * 10;
            const int gy = y0 + row - RT, gx = x0 - 4 + 4 * g;
            v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < NG && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const long long gi = (long long)b * H * W + (long long)gy * W + gx;
                if constexpr (INIT) {
                    v[k] = *reinterpret_cast<const f32x4*>(prob + gi);
                    if (mask) {
                        const uchar4 m = *reinterpret_cast<const uchar4*>(mask + gi);
                        // prob * valid_mask (predict_align_image_pair.py:128)
                        v[k][0] *= m.x ? 1.f : 0.f; v[k][1] *= m.y ? 1.f : 0.f; v[k][2] *= m.z ? 1.f : 0.f; v[k][3] *= m.w ? 1.f : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[k][e] = (v[k][e] > min_prob) ? v[k][e] : 0.f;      // utils.py:97
                } else {
                    v[k] = *reinterpret_cast<const f32x4*>(work + gi);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * 256;
            const int row = idx / 10, g = idx - row * 10;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int lx = 4 * g + e - 4 + RT;
                if (idx < NG && lx >= 0 && lx < LWC) {
                    const int f = row * LWC + lx;
                    t[f] = v[k][e];
                    if (v[k][e] > 0.f && row >= RT && row < RT + NT && lx >= RT && lx < RT + NT)
                        list[0][atomicAdd(&cnt[0], 1)] = (unsigned short)f;
                }
            }
        }
    } else {
        for (int f = tid; f < LW * LW; f += 256) place(f, fetch(f));
    }
    __syncthreads();

    int cur = 0;
    if constexpr (RT > 0) {
        // EIGHT lanes per candidate, one window row each (a tile holds a few dozen candidates: one thread per candidate left
        // seven waves in eight idle and the eighth walking 48 neighbours), 32 candidates per pass of the workgroup; the lanes'
        // verdicts meet by three xor-shuffles and lane 0 of the eight applies the transition AT ONCE -- both rules are sound under
        // any mixture of old and new neighbour states -- and re-lists the candidate if it is still undecided.
        static_assert(2 * RT + 1 <= 8, "one lane per window row");
        const int sub = tid & 7, grp = tid >> 3;
        const int dy = sub - RT;
        const bool row_on = sub <= 2 * RT, row_earlier = dy < 0, row_same = dy == 0;
        unsigned rm = 0u;
#pragma unroll
        for (int q = 0; q <= 2 * RT; ++q) rm = (sub == q) ? fp.rowmask[q] : rm;
        const int row_off = row_on ? dy * LW : 0;
        for (int iter = 0; iter < 64; ++iter) {
            const int n = cnt[cur];
            if (n == 0) break;
            if (tid == 0) cnt[cur ^ 1] = 0;
            __syncthreads();
            int changed = 0;
            for (int base = 0; base < n; base += 32) {
                const int i = base + grp;
                const bool valid = i < n;
                const int c = list[cur][valid ? i : 0];
                const float s = t[c];
                float nb[2 * RT + 1];
#pragma unroll
                for (int dx = -RT; dx <= RT; ++dx) nb[dx + RT] = t[c + row_off + dx];
                bool kill = false, blocked = false, near = false, clear = false;
#pragma unroll
                for (int dx = -RT; dx <= RT; ++dx) {
                    const bool on = row_on && ((rm >> (dx + RT)) & 1u) && !(dx == 0 && row_same);
                    const float v = on ? nb[dx + RT] : 0.f;             // 0: neither kept (< 0) nor of higher priority (s > 0)
                    kill |= v < 0.f;
                    const bool tied = (v < 0.f) && (-v - s <= tie_eps);         // a kept neighbour within the guard's window
                    near |= tied; clear |= (v < 0.f) && !tied;
                    const bool earlier = row_earlier || (row_same && dx < 0);   // lower flat index
                    blocked |= (v > s) || (v == s && earlier);
                }
                unsigned f = (kill ? 1u : 0u) | (blocked ? 2u : 0u) | (near ? 4u : 0u) | (clear ? 8u : 0u);
                f |= (unsigned)__shfl_xor((int)f, 1);
                f |= (unsigned)__shfl_xor((int)f, 2);
                f |= (unsigned)__shfl_xor((int)f, 4);
                if (valid && sub == 0) {
                    const float nvv = (f & 1u) ? 0.f : ((f & 2u) ? s : -s);
                    if (nvv != s) {
                        changed = 1;
                        t[c] = nvv;
                        if (tie_pairs && (f & 13u) == 5u) atomicAdd(&tie_pairs[b], 1);     // died to near-tied kept neighbours only
                        if constexpr (!INIT) {      // later rounds change a handful of pixels: those go to the work map one by one
                            const int ly = c / LW, lx = c - ly * LW;
                            img[(long long)(y0 + ly - RT) * W + (x0 + lx - RT)] = nvv;
                        }
                    }
                    if (nvv > 0.f) list[cur ^ 1][atomicAdd(&cnt[cur ^ 1], 1)] = (unsigned short)c;
                }
            }
            const int any = __syncthreads_or(changed);
            cur ^= 1;                                            // (both lists hold the same entries if nothing changed)
            if (!any) break;                                     // nothing can change without new halo data
        }
    } else {
    for (int iter = 0; iter < 64; ++iter) {
        const int n = cnt[cur];
        if (n == 0) break;
        if (tid == 0) cnt[cur ^ 1] = 0;
        float nv[4];
        int pos[4];
        int changed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + k * 256;
            pos[k] = -1;
            nv[k] = 0.f;
            if (i < n) {
                const int c = list[cur][i];
                const float s = t[c];
                bool kill = false, blocked = false, near = false, clear = false;
                for (int dy = -R; dy <= R; ++dy) {
                    const unsigned rmask = fp.rowmask[dy + R];
                    for (int dx = -R; dx <= R; ++dx)
                        if (((rmask >> (dx + R)) & 1u) && (dy != 0 || dx != 0)) {
                            const float nb = t[c + dy * LW + dx];
                            kill |= nb < 0.f;
                            const bool tied = (nb < 0.f) && (-nb - s <= tie_eps);
                            near |= tied; clear |= (nb < 0.f) && !tied;
                            const bool earlier = (dy < 0) || (dy == 0 && dx < 0);       // lower flat index
                            blocked |= (nb > s) || (nb == s && earlier);
                        }
                }
                pos[k] = c;
                nv[k] = kill ? 0.f : (blocked ? s : -s);
                changed |= (nv[k] != s);
                if (tie_pairs && kill && near && !clear) atomicAdd(&tie_pairs[b], 1);
            }
        }
        const int any = __syncthreads_or(changed);          // all reads of t[] done
        if (!any) break;                                     // nothing can change without new halo data
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (pos[k] >= 0) {
                t[pos[k]] = nv[k];
                if (nv[k] > 0.f) list[cur ^ 1][atomicAdd(&cnt[cur ^ 1], 1)] = (unsigned short)pos[k];
            }
        __syncthreads();
        cur ^= 1;
    }
    }

    // each thread owns 4 consecutive pixels of one tile row: one 16-byte store (round 0 writes the whole thresholded tile; the
    // later rounds of the compile-time footprints have written their few changes already)
    if (INIT || RT == 0) {
        const int py = tid >> 3, px = (tid & 7) * 4;
        const int gy = y0 + py, gx = x0 + px;
        if (gy < H && gx < W) {
            const float* const r = &t[(py + R) * LW + px + R];
            *reinterpret_cast<f32x4*>(&img[(long long)gy * W + gx]) = f32x4{r[0], r[1], r[2], r[3]};
        }
    }
    if (tid == 0) {
        const int und = cnt[cur];
        fout[tile_id] = und;
    }
    if (LOOP) __syncthreads();                               // t[], list[], cnt[] are reused by the next tile
    }
}

// undecided pixels after a round = sum of the tiles' counts -> *slot (what the host reads), and added to *total (optional)
__global__ __launch_bounds__(1024) void nms_accumulate_kernel(const int* __restrict__ counts, int ntiles, int* __restrict__ slot,
                                                              int* __restrict__ total)
{
    __shared__ int part[16];
    int sum = 0;
    for (int i = threadIdx.x; i < ntiles; i += 1024) sum += counts[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int all = 0;
        for (int w = 0; w < 16; ++w) all += part[w];
        *slot = all;
        if (total && all) atomicAdd(total, all);
    }
}

}  // namespace

// the undecided count after round `round` (flags / slots as laid out below) -> its slot, and onto the persistent counter
// (read + reset by mp_nms_unresolved) when `total` is given
void launch_nms_accumulate(int* remaining, int B, int H, int W, int round, int* total, hipStream_t s)
{
    const int ntiles = B * ((W + NT - 1) / NT) * ((H + NT - 1) / NT);
    const int* counts = remaining + 64 + ((round + 1) & 1) * ntiles;
    hipLaunchKernelGGL(nms_accumulate_kernel, dim3(1), dim3(1024), 0, s, counts, ntiles, remaining + (round & 63), total);
}

void launch_nms_init(const float* prob, const uint8_t* mask, float min_prob, float* work,
                     long long n, hipStream_t s)
{
    const long long n4 = n / 4;
    if (n4 <= 0) return;
    hipLaunchKernelGGL(nms_init_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, prob,
                       mask, min_prob, work, n4);
}

// layout behind `remaining`: remaining[0..63] per-round undecided totals (written by nms_accumulate_kernel when the host asks),
// then the 2 * ntiles ping-pong tile counts
void launch_nms_round(float* work, int B, int H, int W, const NmsFootprint& fp, int* remaining,
                      int round, hipStream_t s, float tie_eps, int* tie_pairs)
{
    const int tiles_x = (W + NT - 1) / NT, tiles_y = (H + NT - 1) / NT;
    const int ntiles = B * tiles_x * tiles_y;
    if (ntiles <= 0) return;
    int* flags = remaining + 64;
    const float* np = nullptr;
    const uint8_t* nm = nullptr;
#ifndef MP_NMS_LOOP_FROM
#define MP_NMS_LOOP_FROM 1
#endif
#ifndef MP_NMS_LOOP_GRID
#define MP_NMS_LOOP_GRID 2048
#endif
#ifndef MP_NMS_NO_LOOP       // (developer A/B: tools/build_variant.sh nl "-DMP_NMS_NO_LOOP" nms.hip)
    if (fp.R == 3 && round >= MP_NMS_LOOP_FROM)       // the usual footprint (size 4): later rounds on a small grid that walks the tiles
        hipLaunchKernelGGL((nms_round_kernel<3, false, true>), dim3((unsigned)(ntiles < MP_NMS_LOOP_GRID ? ntiles : MP_NMS_LOOP_GRID)), dim3(256), 0, s, work, H, W,
                           tiles_x, tiles_y, fp, flags, ntiles, round, np, nm, 0.f, W, tie_eps, tie_pairs);
    else
#endif
    if (fp.R == 3)
        hipLaunchKernelGGL((nms_round_kernel<3, false>), dim3((unsigned)ntiles), dim3(256), 0, s, work, H, W, tiles_x,
                           tiles_y, fp, flags, ntiles, round, np, nm, 0.f, W, tie_eps, tie_pairs);
    else if (fp.R == 1)
        hipLaunchKernelGGL((nms_round_kernel<1, false>), dim3((unsigned)ntiles), dim3(256), 0, s, work, H, W, tiles_x,
                           tiles_y, fp, flags, ntiles, round, np, nm, 0.f, W, tie_eps, tie_pairs);
    else
        hipLaunchKernelGGL((nms_round_kernel<0, false>), dim3((unsigned)ntiles), dim3(256), 0, s, work, H, W, tiles_x,
                           tiles_y, fp, flags, ntiles, round, np, nm, 0.f, W, tie_eps, tie_pairs);
}

// round 0 with the candidate listing fused in: replaces launch_nms_init + launch_nms_round(round 0)
// W: width of the work map (a multiple of 4); Ws <= W: true width = row stride of prob / mask (Ws != W: the generic kernel's scalar loads)
void launch_nms_round0(const float* prob, const uint8_t* mask, float min_prob, float* work, int B, int H, int W,
                       const NmsFootprint& fp, int* remaining, hipStream_t s, int Ws, float tie_eps, int* tie_pairs)
{
    const int tiles_x = (W + NT - 1) / NT, tiles_y = (H + NT - 1) / NT;
    const int ntiles = B * tiles_x * tiles_y;
    if (ntiles <= 0) return;
    int* flags = remaining + 64;
    if (fp.R == 3 && Ws == W)
        hipLaunchKernelGGL((nms_round_kernel<3, true>), dim3((unsigned)ntiles), dim3(256), 0, s, work, H, W, tiles_x,
                           tiles_y, fp, flags, ntiles, 0, prob, mask, min_prob, Ws, tie_eps, tie_pairs);
    else if (fp.R == 1 && Ws == W)
        hipLaunchKernelGGL((nms_round_kernel<1, true>), dim3((unsigned)ntiles), dim3(256), 0, s, work, H, W, tiles_x,
                           tiles_y, fp, flags, ntiles, 0, prob, mask, min_prob, Ws, tie_eps, tie_pairs);
    else
        hipLaunchKernelGGL((nms_round_kernel<0, true>), dim3((unsigned)ntiles), dim3(256), 0, s, work, H, W, tiles_x,
                           tiles_y, fp, flags, ntiles, 0, prob, mask, min_prob, Ws, tie_eps, tie_pairs);
}

// Descriptor sampling and mutual-nearest-neighbour matching.
//
// sample_desc : utils.interpolate_descriptors (reference multipoint/utils/utils.py:159-167):
//     g = kp / (S * 0.5) - 1  ->  F.grid_sample(bilinear, zeros padding, align_corners=True)
//     -> F.normalize(p=2, dim=1).  One wave per keypoint, lane = channel; the coarse descriptor
//     map is channels-last, so each of the 4 bilinear taps is one coalesced 256-byte load.
// match       : NNMatcher.match (reference multipoint/utils/matching.py:41-72), which is also how
//     cv2.BFMatcher(NORM_L2, crossCheck=True) (matching.py:7,31) is restated for unit descriptors:
//         dmat = sqrt(2 - 2 * clip(d1 . d2^T, -1, 1)); idx = argmin(axis=1); idx2 = argmin(axis=0)
//         keep (i, idx[i]) iff idx2[idx[i]] == i [and dmat < threshold]; lowest index wins ties.
//     The N x M similarity tiles are computed on v_mfma_f32_32x32x2_f32 (exact fp32); the distance
//     and a packed (distance bits << 32 | index) running arg-min are fused behind the MFMAs, so
//     the matrix never exists in memory.  Each direction (rows of A against B, rows of B against A)
//     is one pass; a*b is commutative and both passes add the products in the same k order, so the
//     two passes see bit-identical distances and the mutual test is exact.
#include "mp_common.h"

namespace {

__global__ __launch_bounds__(256) void sample_desc_kernel(const float* __restrict__ desc, int B, int Hc,
                                                         int Wc, int D, int H, int W,
                                                         const int* __restrict__ kp_yx,
                                                         const int* __restrict__ kp_count, int K,
                                                         float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long long)B * K) return;
    const int b = (int)(wid / K), k = (int)(wid % K);
    const int cnt = min(kp_count[b], K);
    if (k >= cnt) {                                   // rows beyond the image's keypoint count: zeros (the output is fully written)
        for (int r = 0; r < (D >> 6); ++r) out[wid * D + r * 64 + lane] = 0.f;
        return;
    }
    const int y = kp_yx[wid * 2], x = kp_yx[wid * 2 + 1];
    // utils.py:162-163 (fp32) then ATen grid_sampler unnormalize, align_corners=True
    const float gy = (float)y / ((float)H * 0.5f) - 1.0f;
    const float gx = (float)x / ((float)W * 0.5f) - 1.0f;
    const float iy = ((gy + 1.f) / 2.f) * (float)(Hc - 1);
    const float ix = ((gx + 1.f) / 2.f) * (float)(Wc - 1);
    const float y0f = floorf(iy), x0f = floorf(ix);
    const float y1f = y0f + 1.f, x1f = x0f + 1.f;
    const float w_nw = (x1f - ix) * (y1f - iy), w_ne = (ix - x0f) * (y1f - iy);
    const float w_sw = (x1f - ix) * (iy - y0f), w_se = (ix - x0f) * (iy - y0f);
    const int y0 = (int)y0f, x0 = (int)x0f, y1 = y0 + 1, x1 = x0 + 1;
    const bool vy0 = y0 >= 0 && y0 < Hc, vy1 = y1 >= 0 && y1 < Hc;
    const bool vx0 = x0 >= 0 && x0 < Wc, vx1 = x1 >= 0 && x1 < Wc;
    const float* base = desc + (long long)b * Hc * Wc * D;
    float ss = 0.f;
    float vals[4];                      // D <= 256
    const int nrep = D >> 6;
    for (int r = 0; r < nrep; ++r) {
        const int c = r * 64 + lane;
        const float nw = (vy0 && vx0) ? base[((long long)y0 * Wc + x0) * D + c] : 0.f;
        const float ne = (vy0 && vx1) ? base[((long long)y0 * Wc + x1) * D + c] : 0.f;
        const float sw = (vy1 && vx0) ? base[((long long)y1 * Wc + x0) * D + c] : 0.f;
        const float se = (vy1 && vx1) ? base[((long long)y1 * Wc + x1) * D + c] : 0.f;
        const float v = nw * w_nw + ne * w_ne + sw * w_sw + se * w_se;
        vals[r] = v;
        ss += v * v;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) ss += __shfl_xor(ss, off);
    const float denom = fmaxf(sqrtf(ss), 1e-12f);          // F.normalize eps
    for (int r = 0; r < nrep; ++r) out[wid * D + r * 64 + lane] = vals[r] / denom;
}

// best[x] = min over y of (dist(x,y) bits << 32 | y), X rows against all Y rows.
// grid: (row-block groups, pairs, 2 directions); each wave owns one 32-row block of X; the four waves of a workgroup walk the
// same 32-column tiles of Y, which reach them through LDS: a tile is 32 x D contiguous floats, staged with fully coalesced
// 16-byte loads (one 1 KiB run per wave instruction) one tile ahead, and read back as MFMA fragments (padded rows: conflict
// free).  Fetching the fragments straight from global memory -- 16 bytes out of every 128-byte line of 32 lines per load
// instruction, each wave for itself -- kept the kernel at twice its MFMA time (131 us for 62 us of v_mfma_f32_32x32x2_f32).
template <int D>
__global__ __launch_bounds__(256) void nn_rows_kernel(const float* __restrict__ dA, const int* __restrict__ nA,
                                                     const float* __restrict__ dB, const int* __restrict__ nB,
                                                     long long pair_stride, int count_stride, int K,
                                                     unsigned long long* __restrict__ bestA,
                                                     unsigned long long* __restrict__ bestB, int* __restrict__ match_count,
                                                     int nsplit)
{
    // (the pair's match counter, which mutual_kernel adds to behind this launch, is zeroed here: one fill launch fewer)
    if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) match_count[blockIdx.y] = 0;
    constexpr int RS = D + 4;                            // LDS row stride in floats
    __shared__ __attribute__((aligned(16))) float ytile[2][32 * RS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    // blockIdx.z = direction + 2 * column share: the Y columns are cut into `nsplit` contiguous shares of whole tiles, each share
    // leaves its own arg-min array (mutual_kernel takes the minimum of the shares' keys) -- twice the waves per SIMD for the same
    // work: the epilogue of one wave has another wave's MFMAs to hide behind
    const int p = blockIdx.y, dir = blockIdx.z & 1, share = blockIdx.z >> 1;
    const float* X = (dir == 0 ? dA : dB) + (long long)p * pair_stride;
    const float* Y = (dir == 0 ? dB : dA) + (long long)p * pair_stride;
    const int nx = min((dir == 0 ? nA : nB)[p * count_stride], K);
    const int ny = min((dir == 0 ? nB : nA)[p * count_stride], K);
    unsigned long long* best = (dir == 0 ? bestA : bestB) + ((long long)share * gridDim.y + p) * K;
    const int ntile = (ny + 31) >> 5, per = (ntile + nsplit - 1) / nsplit;
    const int c_begin = min(share * per, ntile) * 32, c_end = min(min((share + 1) * per, ntile) * 32, ny);
    if ((int)blockIdx.x * 128 >= nx) return;             // (the whole workgroup)
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    const bool active = r0 < nx;                         // a wave without rows still stages tiles and meets the barriers

    constexpr int NG = D / 8;
    f32x4 a[NG];
    {
        const int row = min(r0 + li, nx - 1);
#pragma unroll
        for (int g = 0; g < NG; ++g)
            a[g] = *reinterpret_cast<const f32x4*>(X + (long long)row * D + g * 8 + half * 4);
    }
    // The products are formed TRANSPOSED (Y fragment as the A operand): lane li owns X row r0 + li, its 16 accumulator registers
    // are 16 columns of the tile -- ONE running arg-min per lane, over columns that arrive in ascending order, so a later column
    // replaces the best one only with a strictly smaller distance.  d = sqrt(u), u = 2 - 2 clip(x.y), is monotone in u:
    // "u < the smallest u seen" is a necessary condition that costs one compare, and the correctly rounded sqrt (~20
    // instructions) + 64-bit key update run only for candidates that pass it -- a lane has seen 16 (t - 1) columns before tile
    // t, so few do.  Keys, hence ties (lowest index wins), are exactly those of the per-element form.
    unsigned long long run = ~0ull;
    float ub = __builtin_inff();

    // staging: the tile's 32 * D / 4 granules of 16 bytes, D / 32 per thread (rows beyond ny repeat row ny - 1; never selected)
    constexpr int GPT = D / 32, GPR = D / 4;
    f32x4 stage[GPT];
    auto gload = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            const int gran = tid + k * 256;
            const int row = gran / GPR, q = gran - row * GPR;
            stage[k] = *reinterpret_cast<const f32x4*>(Y + (long long)min(c0 + row, ny - 1) * D + q * 4);
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            const int gran = tid + k * 256;
            const int row = gran / GPR, q = gran - row * GPR;
            *reinterpret_cast<f32x4*>(&ytile[buf][row * RS + q * 4]) = stage[k];
        }
    };
    if (c_begin < c_end) { gload(c_begin); lstore(0); }
    __syncthreads();
    for (int c0 = c_begin, buf = 0; c0 < c_end; c0 += 32, buf ^= 1) {
        const bool more = c0 + 32 < c_end;
        if (more) gload(c0 + 32);                        // in flight across this tile's MFMAs
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&ytile[buf][li * RS + g * 8 + half * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[e], a[g][e], acc, 0, 0, 0);      // acc[r]: column i(r) of the tile, row li
        }
        if (active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = c0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float t = fminf(fmaxf(acc[r], -1.f), 1.f);               // np.clip, matching.py:51
                const float u = 2.f - 2.f * t;
                if (col < ny && u < ub) {
                    ub = u;
                    const float d = sqrtf(u);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)col;
                    run = key < run ? key : run;
                }
            }
        }
        if (more) lstore(buf ^ 1);                       // last read one barrier ago
        __syncthreads();
    }
    if (!active) return;
    // the two half-waves hold the two halves of the row's columns
    {
        const unsigned long long o = __shfl_xor(run, 32);
        run = o < run ? o : run;
        const int row = r0 + li;
        if (half == 0 && row < nx) best[row] = run;
    }
}

__global__ __launch_bounds__(256) void mutual_kernel(const unsigned long long* __restrict__ bestA,
                                                    const unsigned long long* __restrict__ bestB,
                                                    const int* __restrict__ nA, const int* __restrict__ nB,
                                                    int count_stride, int K, float thr,
                                                    int* __restrict__ match_idx, float* __restrict__ match_dist,
                                                    int* __restrict__ match_count, int nsplit)
{
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const long long share_stride = (long long)gridDim.y * K;        // the column shares' arrays lie [share][pair][K]
    auto best_of = [&](const unsigned long long* b, long long at) -> unsigned long long {
        unsigned long long v = b[at];
        for (int sh = 1; sh < nsplit; ++sh) { const unsigned long long o = b[at + sh * share_stride]; v = o < v ? o : v; }
        return v;
    };
    const int na = min(nA[p * count_stride], K), nb = min(nB[p * count_stride], K);
    int hit = 0;
    if (i < K) {
        int j = -1;
        float d = 0.f;
        if (i < na && nb > 0) {
            const unsigned long long v = best_of(bestA, (long long)p * K + i);
            const int jj = (int)(v & 0xffffffffu);
            d = __uint_as_float((unsigned)(v >> 32));
            const unsigned long long w = best_of(bestB, (long long)p * K + jj);
            const bool mutual = (int)(w & 0xffffffffu) == i;                   // matching.py:58-59
            const bool close = (thr < 0.f) || (d < thr);                        // matching.py:56
            if (mutual && close) j = jj;
        }
        match_idx[(long long)p * K + i] = j;
        match_dist[(long long)p * K + i] = j >= 0 ? d : 0.f;
        hit = j >= 0;
    }
    const int c = __syncthreads_count(hit);
    if (threadIdx.x == 0 && c) atomicAdd(&match_count[p], c);
}

}  // namespace

void launch_sample_desc(const float* desc, int B, int Hc, int Wc, int D, int H, int W,
                        const int* kp_yx, const int* kp_count, int K, float* out, hipStream_t s)
{
    const long long waves = (long long)B * K;
    if (waves <= 0) return;
    hipLaunchKernelGGL(sample_desc_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, desc, B,
                       Hc, Wc, D, H, W, kp_yx, kp_count, K, out);
}

// rowbest/colbest: [MATCH_SHARES][P][K] packed each; match_count is zeroed by the first launch
void launch_match_impl(const float* dA, const int* nA, const float* dB, const int* nB,
                       long long pair_stride, int count_stride, int P, int K, int D, float thr,
                       unsigned long long* rowbest, unsigned long long* colbest, int* match_idx,
                       float* match_dist, int* match_count, hipStream_t s)
{
    if (P <= 0 || K <= 0) return;
    const dim3 grid((K + 127) / 128, P, 2 * MATCH_SHARES);
    if (D == 64)
        hipLaunchKernelGGL(nn_rows_kernel<64>, grid, dim3(256), 0, s, dA, nA, dB, nB, pair_stride,
                           count_stride, K, rowbest, colbest, match_count, MATCH_SHARES);
    else if (D == 128)
        hipLaunchKernelGGL(nn_rows_kernel<128>, grid, dim3(256), 0, s, dA, nA, dB, nB, pair_stride,
                           count_stride, K, rowbest, colbest, match_count, MATCH_SHARES);
    else
        hipLaunchKernelGGL(nn_rows_kernel<256>, grid, dim3(256), 0, s, dA, nA, dB, nB, pair_stride,
                           count_stride, K, rowbest, colbest, match_count, MATCH_SHARES);
    hipLaunchKernelGGL(mutual_kernel, dim3((K + 255) / 256, P), dim3(256), 0, s, rowbest, colbest, nA,
                       nB, count_stride, K, thr, match_idx, match_dist, match_count, MATCH_SHARES);
}

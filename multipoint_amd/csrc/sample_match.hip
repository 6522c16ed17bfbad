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
    if (k >= cnt) return;
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
// grid: (row-block groups, pairs, 2 directions); each wave owns one 32-row block of X.
// (D = 64: at most 96 registers per lane so that a wave fits beside a resident Winograd convolution wave, see keypoints.hip)
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(D == 64 ? 5 : 1, D == 64 ? 5 : 8))) void nn_rows_kernel(const float* __restrict__ dA, const int* __restrict__ nA,
                                                     const float* __restrict__ dB, const int* __restrict__ nB,
                                                     long long pair_stride, int count_stride, int K,
                                                     unsigned long long* __restrict__ bestA,
                                                     unsigned long long* __restrict__ bestB)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int p = blockIdx.y, dir = blockIdx.z;
    const float* X = (dir == 0 ? dA : dB) + (long long)p * pair_stride;
    const float* Y = (dir == 0 ? dB : dA) + (long long)p * pair_stride;
    const int nx = min((dir == 0 ? nA : nB)[p * count_stride], K);
    const int ny = min((dir == 0 ? nB : nA)[p * count_stride], K);
    unsigned long long* best = (dir == 0 ? bestA : bestB) + (long long)p * K;
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    if (r0 >= nx) return;

    constexpr int NG = D / 8;
    f32x4 a[NG];
    {
        const int row = min(r0 + li, nx - 1);
#pragma unroll
        for (int g = 0; g < NG; ++g)
            a[g] = *reinterpret_cast<const f32x4*>(X + (long long)row * D + g * 8 + half * 4);
    }
    unsigned long long run[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) run[r] = ~0ull;

    for (int c0 = 0; c0 < ny; c0 += 32) {
        const int col = c0 + li;
        const int colc = min(col, ny - 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(Y + (long long)colc * D + g * 8 + half * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][e], bv[e], acc, 0, 0, 0);
        }
        if (col < ny) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t = fminf(fmaxf(acc[r], -1.f), 1.f);               // np.clip, matching.py:51
                const float d = sqrtf(2.f - 2.f * t);
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)col;
                run[r] = key < run[r] ? key : run[r];
            }
        }
    }
    // reduce over the 32 columns held by the lanes of each half-wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned long long v = run[r];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const unsigned long long o = __shfl_xor(v, off);
            v = o < v ? o : v;
        }
        const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (li == 0 && row < nx) best[row] = v;
    }
}

__global__ __launch_bounds__(256) void mutual_kernel(const unsigned long long* __restrict__ bestA,
                                                    const unsigned long long* __restrict__ bestB,
                                                    const int* __restrict__ nA, const int* __restrict__ nB,
                                                    int count_stride, int K, float thr,
                                                    int* __restrict__ match_idx, float* __restrict__ match_dist,
                                                    int* __restrict__ match_count)
{
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int na = min(nA[p * count_stride], K), nb = min(nB[p * count_stride], K);
    int hit = 0;
    if (i < K) {
        int j = -1;
        float d = 0.f;
        if (i < na && nb > 0) {
            const unsigned long long v = bestA[(long long)p * K + i];
            const int jj = (int)(v & 0xffffffffu);
            d = __uint_as_float((unsigned)(v >> 32));
            const unsigned long long w = bestB[(long long)p * K + jj];
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

// rowbest/colbest: [P][K] packed; match_count must be zeroed by the caller
void launch_match_impl(const float* dA, const int* nA, const float* dB, const int* nB,
                       long long pair_stride, int count_stride, int P, int K, int D, float thr,
                       unsigned long long* rowbest, unsigned long long* colbest, int* match_idx,
                       float* match_dist, int* match_count, hipStream_t s)
{
    if (P <= 0 || K <= 0) return;
    const dim3 grid((K + 127) / 128, P, 2);
    if (D == 64)
        hipLaunchKernelGGL(nn_rows_kernel<64>, grid, dim3(256), 0, s, dA, nA, dB, nB, pair_stride,
                           count_stride, K, rowbest, colbest);
    else if (D == 128)
        hipLaunchKernelGGL(nn_rows_kernel<128>, grid, dim3(256), 0, s, dA, nA, dB, nB, pair_stride,
                           count_stride, K, rowbest, colbest);
    else
        hipLaunchKernelGGL(nn_rows_kernel<256>, grid, dim3(256), 0, s, dA, nA, dB, nB, pair_stride,
                           count_stride, K, rowbest, colbest);
    hipLaunchKernelGGL(mutual_kernel, dim3((K + 255) / 256, P), dim3(256), 0, s, rowbest, colbest, nA,
                       nB, count_stride, K, thr, match_idx, match_dist, match_count);
}

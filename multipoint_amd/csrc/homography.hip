// Batched RANSAC homography estimation for the matched keypoints of P pairs: the step that follows
// matching in the reference (predict_align_image_pair.py:205-216, evaluation.py:330-349:
// cv2.findHomography(optical_pts, thermal_pts, cv2.RANSAC, ransacReprojThreshold)).
// OpenCV is a third-party dependency that is absent here and its RANSAC draws from its own RNG, so this is not a
// bit-level restatement ("parity unpinned"); it follows the published algorithm:
//   T hypotheses per pair, each from 4 distinct random matches (counter-based RNG -> reproducible), exact 4-point
//   solve (8x8 Gaussian elimination, fp64), score = number of matches with forward reprojection error <= threshold,
//   best = most inliers (lowest hypothesis index on ties), final model = normalised DLT over the best inlier set.
// One thread per hypothesis (the matches of the pair sit in LDS); the winner is picked with a 64-bit atomicMax.
#include "mp_common.h"

namespace {

__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// 4 distinct indices in [0, n) for hypothesis t of pair p
__device__ __forceinline__ void sample4(unsigned long long seed, int p, int t, int n, int idx[4])
{
    unsigned long long ctr = mix64(seed ^ ((unsigned long long)p << 32) ^ (unsigned long long)t);
    for (int k = 0; k < 4; ++k) {
        for (;;) {
            ctr = mix64(ctr);
            const int c = (int)(ctr % (unsigned long long)n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup |= (idx[j] == c);
            if (!dup) { idx[k] = c; break; }
        }
    }
}

// exact homography through 4 correspondences (x,y) -> (u,v); h[8] = 1.  false if (near-)singular.
__device__ bool solve4(const double* x, const double* y, const double* u, const double* v, double* h)
{
    double a[8][9];
    for (int k = 0; k < 4; ++k) {
        double* r0 = a[2 * k];
        double* r1 = a[2 * k + 1];
        r0[0] = x[k]; r0[1] = y[k]; r0[2] = 1.0; r0[3] = 0.0; r0[4] = 0.0; r0[5] = 0.0; r0[6] = -u[k] * x[k]; r0[7] = -u[k] * y[k]; r0[8] = u[k];
        r1[0] = 0.0; r1[1] = 0.0; r1[2] = 0.0; r1[3] = x[k]; r1[4] = y[k]; r1[5] = 1.0; r1[6] = -v[k] * x[k]; r1[7] = -v[k] * y[k]; r1[8] = v[k];
    }
    for (int c = 0; c < 8; ++c) {
        int piv = c;
        double best = fabs(a[c][c]);
        for (int r = c + 1; r < 8; ++r)
            if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
        if (best < 1e-10) return false;
        if (piv != c)
            for (int k = c; k < 9; ++k) { const double tmp = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = tmp; }
        const double inv = 1.0 / a[c][c];
        for (int r = c + 1; r < 8; ++r) {
            const double f = a[r][c] * inv;
            for (int k = c; k < 9; ++k) a[r][k] -= f * a[c][k];
        }
    }
    for (int c = 7; c >= 0; --c) {
        double s = a[c][8];
        for (int k = c + 1; k < 8; ++k) s -= a[c][k] * h[k];
        h[c] = s / a[c][c];
    }
    h[8] = 1.0;
    return true;
}

__device__ __forceinline__ bool inlier(const double* h, double x, double y, double u, double v, double thr2)
{
    const double w = h[6] * x + h[7] * y + h[8];
    if (fabs(w) < 1e-12) return false;
    const double iw = 1.0 / w;
    const double du = (h[0] * x + h[1] * y + h[2]) * iw - u, dv = (h[3] * x + h[4] * y + h[5]) * iw - v;
    return du * du + dv * dv <= thr2;
}

// gather the matched (x,y)->(u,v) pairs of pair p into LDS / a compact list; returns their number
// pts: [n][4] floats in LDS order x, y, u, v
__device__ int gather(const int* kp_yx, const int* kp_count, const int* match_idx, int p, int K, float* pts, int* qidx)
{
    __shared__ int n_s;
    if (threadIdx.x == 0) n_s = 0;
    __syncthreads();
    const int no = min(kp_count[2 * p], K), nt = min(kp_count[2 * p + 1], K);
    // ordered compaction in chunks of blockDim: keeps query order (as the reference's list comprehension does)
    for (int i0 = 0; i0 < no; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        int j = -1;
        if (i < no) { j = match_idx[(size_t)p * K + i]; if (j >= nt) j = -1; }
        const unsigned long long bal = __ballot(j >= 0);
        __shared__ int wave_base[16];
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        if (ln == 0) wave_base[wv] = __popcll(bal);
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = n_s;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { const int c = wave_base[w]; wave_base[w] = run; run += c; }
            n_s = run;
        }
        __syncthreads();
        if (j >= 0) {
            const int pos = wave_base[wv] + __popcll(bal & ((1ull << ln) - 1ull));
            pts[pos * 4 + 0] = (float)kp_yx[((size_t)(2 * p) * K + i) * 2 + 1];
            pts[pos * 4 + 1] = (float)kp_yx[((size_t)(2 * p) * K + i) * 2];
            pts[pos * 4 + 2] = (float)kp_yx[((size_t)(2 * p + 1) * K + j) * 2 + 1];
            pts[pos * 4 + 3] = (float)kp_yx[((size_t)(2 * p + 1) * K + j) * 2];
            if (qidx) qidx[pos] = i;
        }
        __syncthreads();
    }
    return n_s;
}

__global__ __launch_bounds__(256) void ransac_kernel(const int* __restrict__ kp_yx, const int* __restrict__ kp_count,
                                                     const int* __restrict__ match_idx, int K, int T, double thr,
                                                     unsigned long long seed, unsigned long long* __restrict__ best)
{
    extern __shared__ float pts[];            // [K][4]
    const int p = blockIdx.y;
    const int n = gather(kp_yx, kp_count, match_idx, p, K, pts, nullptr);
    if (n < 4) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    int idx[4];
    sample4(seed, p, t, n, idx);
    double x[4], y[4], u[4], v[4], h[9];
    for (int k = 0; k < 4; ++k) { x[k] = pts[idx[k] * 4]; y[k] = pts[idx[k] * 4 + 1]; u[k] = pts[idx[k] * 4 + 2]; v[k] = pts[idx[k] * 4 + 3]; }
    if (!solve4(x, y, u, v, h)) return;
    const double thr2 = thr * thr;
    int cnt = 0;
    for (int i = 0; i < n; ++i) cnt += inlier(h, pts[i * 4], pts[i * 4 + 1], pts[i * 4 + 2], pts[i * 4 + 3], thr2);
    // most inliers, then the LOWEST hypothesis index
    atomicMax(&best[p], ((unsigned long long)cnt << 32) | (unsigned long long)(0x7fffffff - t));
}

// symmetric 9x9 eigen-decomposition by cyclic Jacobi; returns the eigenvector of the smallest eigenvalue
__device__ void smallest_eigvec9(double a[9][9], double* out)
{
    double vv[9][9];
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) vv[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < 9; ++i) for (int j = i + 1; j < 9; ++j) off += a[i][j] * a[i][j];
        if (off < 1e-30) break;
        for (int pch = 0; pch < 9; ++pch)
            for (int q = pch + 1; q < 9; ++q) {
                if (fabs(a[pch][q]) < 1e-300) continue;
                const double theta = (a[q][q] - a[pch][pch]) / (2.0 * a[pch][q]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 9; ++k) {
                    const double akp = a[k][pch], akq = a[k][q];
                    a[k][pch] = c * akp - s * akq; a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 9; ++k) {
                    const double apk = a[pch][k], aqk = a[q][k];
                    a[pch][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 9; ++k) {
                    const double vkp = vv[k][pch], vkq = vv[k][q];
                    vv[k][pch] = c * vkp - s * vkq; vv[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    for (int i = 1; i < 9; ++i) if (a[i][i] < a[m][m]) m = i;
    for (int k = 0; k < 9; ++k) out[k] = vv[k][m];
}

// one workgroup per pair: re-derive the winning hypothesis, mark its inliers, refit by normalised DLT over them
__global__ __launch_bounds__(256) void refit_kernel(const int* __restrict__ kp_yx, const int* __restrict__ kp_count,
                                                    const int* __restrict__ match_idx, int K, double thr,
                                                    unsigned long long seed, const unsigned long long* __restrict__ best,
                                                    double* __restrict__ H_out, unsigned char* __restrict__ mask,
                                                    int* __restrict__ n_inliers)
{
    extern __shared__ float pts[];            // [K][4] floats, then [K] ints (query index of each match)
    int* qidx = reinterpret_cast<int*>(pts + (size_t)K * 4);
    __shared__ double red[256];
    __shared__ double hsh[9];
    __shared__ double stat[8];
    __shared__ double ata[81];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n = gather(kp_yx, kp_count, match_idx, p, K, pts, qidx);
    const unsigned long long b = best[p];
    const int cnt = (int)(b >> 32);
    if (n < 4 || cnt < 4) {
        if (tid == 0) { n_inliers[p] = 0; for (int k = 0; k < 9; ++k) H_out[p * 9 + k] = 0.0; }
        return;
    }
    if (tid == 0) {
        const int t = 0x7fffffff - (int)(b & 0xffffffffull);
        int idx[4];
        sample4(seed, p, t, n, idx);
        double x[4], y[4], u[4], v[4], h[9];
        for (int k = 0; k < 4; ++k) { x[k] = pts[idx[k] * 4]; y[k] = pts[idx[k] * 4 + 1]; u[k] = pts[idx[k] * 4 + 2]; v[k] = pts[idx[k] * 4 + 3]; }
        solve4(x, y, u, v, h);
        for (int k = 0; k < 9; ++k) hsh[k] = h[k];
    }
    __syncthreads();
    // inlier flags of the best hypothesis + normalisation statistics (centroid, mean distance) of both point sets
    const double thr2 = thr * thr;
    double h[9];
    for (int k = 0; k < 9; ++k) h[k] = hsh[k];
    auto block_sum = [&](double v) -> double {
        red[tid] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    double sx = 0, sy = 0, su = 0, sv = 0, sc = 0;
    for (int i = tid; i < n; i += 256) {
        const bool in = inlier(h, pts[i * 4], pts[i * 4 + 1], pts[i * 4 + 2], pts[i * 4 + 3], thr2);
        mask[(size_t)p * K + qidx[i]] = in ? 1 : 0;
        if (in) { sx += pts[i * 4]; sy += pts[i * 4 + 1]; su += pts[i * 4 + 2]; sv += pts[i * 4 + 3]; sc += 1.0; }
        else pts[i * 4] = __int_as_float(0x7fc00000);          // NaN marks an outlier for the passes below
    }
    const double m = block_sum(sc);
    const double cx = block_sum(sx) / m, cy = block_sum(sy) / m, cu = block_sum(su) / m, cv = block_sum(sv) / m;
    double d1 = 0, d2 = 0;
    for (int i = tid; i < n; i += 256) {
        if (pts[i * 4] != pts[i * 4]) continue;
        d1 += sqrt((pts[i * 4] - cx) * (pts[i * 4] - cx) + (pts[i * 4 + 1] - cy) * (pts[i * 4 + 1] - cy));
        d2 += sqrt((pts[i * 4 + 2] - cu) * (pts[i * 4 + 2] - cu) + (pts[i * 4 + 3] - cv) * (pts[i * 4 + 3] - cv));
    }
    const double md1 = block_sum(d1) / m, md2 = block_sum(d2) / m;
    const double s1 = md1 > 1e-12 ? 1.4142135623730951 / md1 : 1.0, s2 = md2 > 1e-12 ? 1.4142135623730951 / md2 : 1.0;
    // A^T A of the 2m x 9 DLT matrix on the normalised points: thread (r, c) owns one entry
    for (int e = tid; e < 81; e += 256) ata[e] = 0.0;
    __syncthreads();
    double acc9[45];
    for (int k = 0; k < 45; ++k) acc9[k] = 0.0;
    for (int i = tid; i < n; i += 256) {
        if (pts[i * 4] != pts[i * 4]) continue;
        const double x = (pts[i * 4] - cx) * s1, y = (pts[i * 4 + 1] - cy) * s1;
        const double u = (pts[i * 4 + 2] - cu) * s2, v = (pts[i * 4 + 3] - cv) * s2;
        const double r0[9] = {x, y, 1.0, 0.0, 0.0, 0.0, -u * x, -u * y, -u};
        const double r1[9] = {0.0, 0.0, 0.0, x, y, 1.0, -v * x, -v * y, -v};
        int k = 0;
        for (int a = 0; a < 9; ++a) for (int c = a; c < 9; ++c, ++k) acc9[k] += r0[a] * r0[c] + r1[a] * r1[c];
    }
    {
        int k = 0;
        for (int a = 0; a < 9; ++a)
            for (int c = a; c < 9; ++c, ++k) {
                const double s = block_sum(acc9[k]);
                if (tid == 0) { ata[a * 9 + c] = s; ata[c * 9 + a] = s; }
            }
    }
    __syncthreads();
    if (tid == 0) {
        double a[9][9], hv[9];
        for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) a[i][j] = ata[i * 9 + j];
        smallest_eigvec9(a, hv);
        // denormalise: H = T2^-1 * Hn * T1, T = [[s,0,-s*c],[0,s,-s*c],[0,0,1]]
        const double hn[3][3] = {{hv[0], hv[1], hv[2]}, {hv[3], hv[4], hv[5]}, {hv[6], hv[7], hv[8]}};
        double tmp[3][3];
        for (int r = 0; r < 3; ++r) {            // Hn * T1
            tmp[r][0] = hn[r][0] * s1; tmp[r][1] = hn[r][1] * s1;
            tmp[r][2] = -hn[r][0] * s1 * cx - hn[r][1] * s1 * cy + hn[r][2];
        }
        double out[3][3];
        for (int c = 0; c < 3; ++c) {            // T2^-1 = [[1/s,0,cu],[0,1/s,cv],[0,0,1]]
            out[0][c] = tmp[0][c] / s2 + cu * tmp[2][c];
            out[1][c] = tmp[1][c] / s2 + cv * tmp[2][c];
            out[2][c] = tmp[2][c];
        }
        const double nrm = fabs(out[2][2]) > 1e-300 ? 1.0 / out[2][2] : 1.0;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H_out[p * 9 + r * 3 + c] = out[r][c] * nrm;
        n_inliers[p] = (int)m;
    }
}

}  // namespace

void launch_ransac_homography(const int* kp_yx, const int* kp_count, const int* match_idx, int P, int K, int T, double thr,
                              unsigned long long seed, unsigned long long* best, double* H_out, unsigned char* mask,
                              int* n_inliers, hipStream_t s)
{
    const size_t lds1 = (size_t)K * 4 * sizeof(float), lds2 = lds1 + (size_t)K * sizeof(int);
    hipLaunchKernelGGL(ransac_kernel, dim3((T + 255) / 256, P), dim3(256), lds1, s, kp_yx, kp_count, match_idx, K, T, thr, seed, best);
    hipLaunchKernelGGL(refit_kernel, dim3(P), dim3(256), lds2, s, kp_yx, kp_count, match_idx, K, thr, seed, best, H_out, mask,
                       n_inliers);
}

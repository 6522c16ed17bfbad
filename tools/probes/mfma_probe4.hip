// Developer microbenchmark 4: LDS-free implicit GEMM -- MFMA A fragments loaded straight from the NHWC
// activation tensor (L1/L2), B fragments from packed weights; no barriers, waves fully independent.
// Question: can the TA/L1/L2 path feed 16 MFMAs per step per wave at 3 waves per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 64, H = 480, W = 640, B = 64;

template <int DIST>
__global__ __launch_bounds__(256, 3) void probe(const float* __restrict__ w, const float* __restrict__ act, float* out, int tiles_per_wave)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const long long gw = (long long)blockIdx.x * 4 + wave;          // global wave id
    const int total_waves = gridDim.x * 4;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const f32x4* wp0 = reinterpret_cast<const f32x4*>(w) + lane;
    constexpr int RB = DIST + 1;
    constexpr int NSTEP = 72;
    static_assert(NSTEP % RB == 0, "");
    f32x4 af[RB][2], bf[RB][2];
    float sum = 0.f;
    for (int t = 0; t < tiles_per_wave; ++t) {
        int zoff = 0; asm volatile("" : "+v"(zoff));
        const f32x4* wp = wp0 + zoff;              // keeps the weight loads inside the tile loop
        const long long tile = gw + (long long)t * total_waves;
        const int tx = (int)(tile % (W / 32)); const long long r1 = tile / (W / 32);
        const int ty = (int)(r1 % (H / 2)); const int img = (int)((r1 / (H / 2)) % B);
        const float* base = act + (((long long)img * H + ty * 2) * W + tx * 32) * C;     // wave-uniform
        int aoff[9][2];                                                                   // per-lane 32-bit offsets
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                int dy = tap / 3 - 1 + mb, dx = li + tap % 3 - 1;
                if (ty * 2 + dy < 0) dy = 1; if (ty * 2 + dy >= H) dy = -1 + mb;
                if (tx * 32 + dx < 0) dx = 1; if (tx * 32 + dx >= W) dx = 30;
                aoff[tap][mb] = (dy * W + dx) * C + half * 4;
            }
        // 72 steps = 9 taps x 8 channel groups; operands prefetched DIST steps ahead
        auto lda = [&](int s, int mb) { return *reinterpret_cast<const f32x4*>(base + aoff[s >> 3][mb] + (s & 7) * 8); };
#pragma unroll
        for (int d = 0; d < DIST; ++d) {
            af[d][0] = lda(d, 0); af[d][1] = lda(d, 1);
            bf[d][0] = wp[d * 128]; bf[d][1] = wp[d * 128 + 64];
        }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s + DIST < NSTEP) {
                af[(s + DIST) % RB][0] = lda(s + DIST, 0);
                af[(s + DIST) % RB][1] = lda(s + DIST, 1);
                bf[(s + DIST) % RB][0] = wp[(s + DIST) * 128];
                bf[(s + DIST) % RB][1] = wp[(s + DIST) * 128 + 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s % RB][0][e], bf[s % RB][0][e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s % RB][0][e], bf[s % RB][1][e], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s % RB][1][e], bf[s % RB][0][e], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s % RB][1][e], bf[s % RB][1][e], acc[3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) { sum += acc[a][r]; acc[a][r] = 0.f; }
    }
    if (sum == 123.456f) out[tid] = sum;
}

template <int DIST>
void run(const float* w, const float* act, float* out, int bpc, int tiles)
{
    const int grid = 256 * bpc;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<DIST><<<grid, 256>>>(w, act, out, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<DIST><<<grid, 256>>>(w, act, out, tiles);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * 4 * tiles * 72 * 16 * 4096.0;
    printf("DIST=%d blocks/CU=%d  %.3f ms  %.1f TF/s (%.1f%%)  %s\n", DIST, bpc, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100,
           hipGetErrorString(hipGetLastError()));
}

int main()
{
    float *w, *out, *act;
    const size_t n = 72 * 128 * 4 + 4096;
    const size_t act_elems = (size_t)B * H * W * C + 65536;
    hipMalloc(&w, n * 4 * 4); hipMalloc(&out, 4096); hipMalloc(&act, act_elems * 4);
    std::vector<float> hw(n * 4); unsigned x = 12345;
    for (auto& v : hw) { x = x * 1664525u + 1013904223u; v = ((x >> 8) / 16777216.0f - 0.5f) * 0.2f; }
    hipMemcpy(w, hw.data(), n * 16, hipMemcpyHostToDevice);
    hipMemset(act, 0, act_elems * 4);
    for (int rep = 0; rep < 2; ++rep)
        for (int bpc = 2; bpc <= 3; ++bpc) { run<2>(w, act, out, bpc, 20); run<3>(w, act, out, bpc, 20); run<5>(w, act, out, bpc, 20); }
    return 0;
}

// Developer microbenchmark 3: does a chunk structure with ONE barrier per 18 steps, LDS double
// buffering and rolling register staging (6 HBM loads at steps 0-5, 6 ds_write_b128 at steps 9-14)
// keep the fp32 MFMA pipe busy at 2 workgroups per CU?  (design check for the conv kernel)
//   MODE 0: no barrier, no staging   MODE 1: + barrier per chunk   MODE 2: + staging loads/writes
//   MODE 3: MODE 2 with half of the workgroups delayed by half a chunk (de-phasing)
//   MODE 4: staging loads only   MODE 5: LDS writes only   MODE 6: MODE 2 with weights fetched 4 steps ahead
//   MODE 7: MODE 2 with L2-resident staging source   MODE 8: MODE 6 + all staging loads in one burst at step 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int STEPS = 18;
constexpr int PSF = 20;                // pixel stride (floats) for a 16-channel chunk
constexpr int TILE_F = 340 * PSF;      // floats per LDS buffer

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ w, const float* __restrict__ act,
                                                float* out, int iters, long long act_elems)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * TILE_F; i += 256) lds[i] = __uint_as_float(0x3f000000u | ((i * 2654435761u) >> 9)) - 0.75f;
    __syncthreads();
    if (MODE == 3 && (blockIdx.x >= gridDim.x / 2)) { for (int k = 0; k < 3; ++k) __builtin_amdgcn_s_sleep(127); }
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const f32x4* wp0 = reinterpret_cast<const f32x4*>(w) + lane;
    const int ab = ((2 * wave) * 34 + (lane & 31)) * PSF + (lane >> 5) * 4;
    constexpr int DIST = (MODE == 6 || MODE == 8) ? 4 : 2;
    constexpr int RB = (DIST == 4) ? 6 : 3;
    f32x4 af[2][2], bf[RB][2], stg[6];
#pragma unroll
    for (int d = 0; d < DIST; ++d) { bf[d][0] = wp0[d * 128]; bf[d][1] = wp0[d * 128 + 64]; }
#pragma unroll
    for (int j = 0; j < 6; ++j) stg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    long long apos = ((long long)blockIdx.x * 7919 * 4096 + tid * 4) % (act_elems - 65536);
    for (int it = 0; it < iters; ++it) {
        const float* cur = lds + (it & 1) * TILE_F;
        float* nxt = lds + ((it + 1) & 1) * TILE_F;
        const f32x4* wc = wp0 + (it & 7) * (STEPS * 128);
        af[0][0] = *reinterpret_cast<const f32x4*>(&cur[ab]);
        af[0][1] = *reinterpret_cast<const f32x4*>(&cur[ab + 34 * PSF]);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            bf[(s + DIST) % RB][0] = wc[(s + DIST) * 128];
            bf[(s + DIST) % RB][1] = wc[(s + DIST) * 128 + 64];
            if (s + 1 < STEPS) {
                const int sn = s + 1, tap = sn >> 1, gg = sn & 1;
                const int aoff = ((tap / 3) * 34 + tap % 3) * PSF + gg * 8;
                af[sn & 1][0] = *reinterpret_cast<const f32x4*>(&cur[ab + aoff]);
                af[sn & 1][1] = *reinterpret_cast<const f32x4*>(&cur[ab + 34 * PSF + aoff]);
            }
            if (MODE >= 2) {
                if (MODE == 8) {
                    if (s == 0) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) stg[j] = *reinterpret_cast<const f32x4*>(act + apos + j * 16384);
                    }
                } else if (s < 6 && MODE != 5) stg[s] = *reinterpret_cast<const f32x4*>(act + (MODE == 7 ? (apos & 0xfffff) : apos) + s * 16384);
                if (s >= 9 && s < 15 && MODE != 4) {
                    const int f = tid + (s - 9) * 256;
                    if (f < 1360) *reinterpret_cast<f32x4*>(&nxt[(f >> 2) * PSF + (f & 3) * 4]) = stg[s - 9];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][0][e], bf[s % RB][0][e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][0][e], bf[s % RB][1][e], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][1][e], bf[s % RB][0][e], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][1][e], bf[s % RB][1][e], acc[3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        apos += 98304; if (apos > act_elems - 200000) apos -= (act_elems - 400000);
        if (MODE >= 1) __syncthreads();
    }
    if (MODE == 4) { for (int j = 0; j < 6; ++j) asm volatile("" :: "v"(stg[j])); }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[tid] = s;
}

template <int MODE>
void run(const float* w, const float* act, long long act_elems, float* out, int iters)
{
    const int lds = 2 * TILE_F * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<grid, 256, lds>>>(w, act, out, 8, act_elems);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<grid, 256, lds>>>(w, act, out, iters, act_elems);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * 4 * iters * STEPS * 16 * 4096.0;
    printf("MODE=%d  %.3f ms  %.1f TF/s (%.1f%%)  %s\n", MODE, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100,
           hipGetErrorString(hipGetLastError()));
}

int main()
{
    float *w, *out, *act;
    const size_t n = STEPS * 128 * 16 * 9 + 4096;
    const long long act_elems = 1LL << 30;     // 4 GiB of activations: staging loads miss L2/MALL
    hipMalloc(&w, n * 4); hipMalloc(&out, 4096); hipMalloc(&act, act_elems * 4);
    std::vector<float> hw(n); unsigned x = 12345;
    for (auto& v : hw) { x = x * 1664525u + 1013904223u; v = ((x >> 8) / 16777216.0f - 0.5f) * 0.2f; }
    hipMemcpy(w, hw.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(act, 0, act_elems * 4);
    run<1>(w, act, act_elems, out, 400);
    run<1>(w, act, act_elems, out, 800); run<2>(w, act, act_elems, out, 800); run<4>(w, act, act_elems, out, 800);
    run<5>(w, act, act_elems, out, 800); run<6>(w, act, act_elems, out, 800); run<7>(w, act, act_elems, out, 800);
    run<8>(w, act, act_elems, out, 800);
    return 0;
}

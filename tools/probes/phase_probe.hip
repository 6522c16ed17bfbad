// Developer probe (round 6): what a burst of packed fp32 vector instructions costs next to the fp32 MFMA stream of a two-waves-per-SIMD
// workgroup, by WHERE it is issued.  8 waves (one workgroup per CU), each 36 v_mfma_f32_16x16x4_f32 + KV v_pk_fma_f32 per iteration:
//   mode 0  the kernel's placement: two bursts of KV / 2 behind MFMAs 10 and 22 of the wave's own stream, the SIMD's other wave anywhere
//   mode 1  phase-separated: s_barrier, then BOTH waves of every SIMD issue their whole burst, then 36 MFMAs with nothing in between
//   mode 2  as 1 without the barrier (burst at the top of the iteration; the waves drift)
//   mode 3  one burst in the MIDDLE of the wave's stream (behind MFMA 18), no barrier
//   mode 4  phase-separated with ONE barrier per TWO iterations' worth (burst 2 KV, 72 MFMAs)
// KV = 0: the MFMA floor (2304 cycles per SIMD and iteration); MFMA-less: the burst alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int KV, bool MFMA>
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters)
{
    f32x4 acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = (float)threadIdx.x, b = 1.5f;
    f32x2 v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = f32x2{(float)i, (float)threadIdx.x};
    const f32x2 kc = {1.0001f, 0.9999f};
    auto burst = [&](const int n) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < n; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[j % 12]) : "v"(kc));
    };
    auto mfmas = [&](const int lo, const int hi) __attribute__((always_inline)) {
#pragma unroll
        for (int i = lo; i < hi; ++i) {
            if (MFMA) acc[i % 36] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % 36], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { mfmas(0, 10); burst(KV / 2); mfmas(10, 22); burst(KV - KV / 2); mfmas(22, 36); }
        else if (MODE == 1) { __builtin_amdgcn_s_barrier(); burst(KV); mfmas(0, 36); }
        else if (MODE == 2) { burst(KV); mfmas(0, 36); }
        else if (MODE == 3) { mfmas(0, 18); burst(KV); mfmas(18, 36); }
        else { if ((it & 1) == 0) { __builtin_amdgcn_s_barrier(); burst(2 * KV); } mfmas(0, 36); }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 36; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i][0] + v[i][1];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE, int KV, bool MFMA>
double run(float* out)
{
    const int iters = 2000;
    probe<MODE, KV, MFMA><<<256, 512>>>(out, 50);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<MODE, KV, MFMA><<<256, 512>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;          // ns per iteration
}

template <int KV>
void rows(float* out, double n0)
{
    const double ghz = 2304.0 / n0;
    const double alone = run<2, KV, false>(out);
    const double m[5] = {run<0, KV, true>(out), run<1, KV, true>(out), run<2, KV, true>(out), run<3, KV, true>(out), run<4, KV, true>(out)};
    printf("KV = %2d per wave (%d per SIMD): burst alone %.0f cycles per iteration (%.1f per instruction and SIMD)\n", KV, 2 * KV, alone * ghz, alone * ghz / (2 * KV));
    const char* names[5] = {"two bursts inside the stream", "barrier + burst + MFMAs", "burst + MFMAs, no barrier", "one burst mid-stream", "one barrier per two units"};
    for (int i = 0; i < 5; ++i)
        printf("   mode %d %-30s %6.0f cycles per iteration = floor + %5.0f  (%.1f per vector instruction and SIMD)\n", i, names[i], m[i] * ghz,
               m[i] * ghz - 2304.0, (m[i] * ghz - 2304.0) / (2 * KV));
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 1 << 16);
    const double n0 = run<2, 0, true>(out);
    printf("MFMA floor: %.0f ns per iteration of 72 MFMAs per SIMD -> %.3f GHz\n", n0, 2304.0 / n0);
    const double nb = run<1, 0, true>(out);
    printf("with one s_barrier per iteration: %.0f ns (+%.0f cycles)\n", nb, (nb - n0) * 2304.0 / n0);
    rows<12>(out, n0); rows<24>(out, n0); rows<34>(out, n0); rows<48>(out, n0);
    return 0;
}

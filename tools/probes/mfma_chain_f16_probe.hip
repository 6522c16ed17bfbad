// Developer probe (round 5): how many INDEPENDENT accumulator chains does ONE wave need to keep its SIMD's matrix pipe busy with
// v_mfma_f32_32x32x16_f16?  NW waves per workgroup (one workgroup per CU: NW / 4 waves per SIMD), each issuing 144 * 4 / NW MFMAs per
// iteration round-robin over NACC accumulators; ns per iteration against the 144 x 32 = 4608 pipe cycles per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NW, int NACC>
__global__ __launch_bounds__(NW * 64, 1) void probe(float* out, int iters)
{
    constexpr int NM = 144 * 4 / NW;
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); b[e] = (_Float16)(0.02f * (threadIdx.x - e)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int NW, int NACC>
void run(float* out)
{
    const int iters = 1000;
    probe<NW, NACC><<<256, NW * 64>>>(out, 50);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<NW, NACC><<<256, NW * 64>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters;
    printf("%d waves/SIMD, %2d accumulator chains per wave: %.0f ns per 4608 pipe cycles -> %.2f GHz-equivalent (TFLOP/s %.0f)\n", NW / 4, NACC, ns,
           4608.0 / ns, 256.0 * 4 * 144 * 32768.0 / ns / 1e3);
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 1 << 16);
    run<4, 1>(out); run<4, 2>(out); run<4, 4>(out); run<4, 6>(out); run<4, 8>(out); run<4, 12>(out);
    run<8, 1>(out); run<8, 2>(out); run<8, 4>(out); run<8, 8>(out);
    run<12, 2>(out); run<12, 4>(out);
    run<16, 1>(out); run<16, 2>(out); run<16, 4>(out);
    return 0;
}

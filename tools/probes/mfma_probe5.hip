// Developer microbenchmark 5: what does ONE extra VALU instruction cost a wave that streams fp32 MFMAs?
// (fp32 MFMA runs at the fp32 vector rate -- do v_fma_f32 and v_mfma_f32_32x32x2_f32 share the pipe?)
// NV = extra independent v_fma_f32 per 16-MFMA step, spread one per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ w, float* out, int iters)
{
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(tid + i) * 1e-3f;
    const float a0 = w[lane], b0 = w[lane + 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 36; ++s) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[e & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[e & 3], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < (NV + 15 - e) / 16; ++v)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(e + v) & 7]) : "v"(a0), "v"(b0));
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456f) out[tid] = s;
}

template <int NV>
void run(const float* w, float* out, int bpc)
{
    const int grid = 256 * bpc, iters = 300;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<NV><<<grid, 256>>>(w, out, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<NV><<<grid, 256>>>(w, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * 4 * iters * 36 * 16 * 4096.0;
    printf("extra VALU per 16 MFMA = %3d  blocks/CU=%d  %.3f ms  %.1f TF/s (%.1f%%)\n", NV, bpc, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main()
{
    float *w, *out;
    hipMalloc(&w, 4096); hipMalloc(&out, 4096); hipMemset(w, 0, 4096);
    for (int bpc = 1; bpc <= 2; ++bpc) { run<0>(w, out, bpc); run<0>(w, out, bpc); run<8>(w, out, bpc); run<16>(w, out, bpc); run<32>(w, out, bpc); run<64>(w, out, bpc); }
    return 0;
}

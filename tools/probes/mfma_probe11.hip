// Developer microbenchmark 11: the F(4x4,3x3) kernel's situation -- TWO waves per SIMD (512 threads, one workgroup per CU), each
// streaming v_mfma_f32_16x16x4_f32 over 36 accumulators of 4 registers -- with the accumulators in VGPRs (the form hipcc picks
// for conv_wino43.hip) or in AGPRs, and N LDS stores / loads per 36 MFMAs.  Prints cycles per unit of 36 MFMAs of a wave
// (floor: 2 waves x 36 x 32 = 2304).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool AGPR, int FILL, int N>   // N fillers per unit of 36 MFMAs, one per MFMA gap from the first gap on
__global__ __launch_bounds__(512, 1) void probe(const float* __restrict__ w, float* out, int iters, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int tid = threadIdx.x, lane = tid & 63;
    f32x4 acc[36];
    for (int a = 0; a < 36; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 16384; i += 512) lds[i] = 0.f;
    __syncthreads();
    f32x2 x[4];
    for (int i = 0; i < 4; ++i) x[i] = f32x2{(float)(tid + i) * 1e-3f, 1.f};
    f32x4 av = *reinterpret_cast<const f32x4*>(w + lane * 4), bv = *reinterpret_cast<const f32x4*>(w + 256 + lane * 4);
    f32x4 r4[4];
    for (int i = 0; i < 4; ++i) r4[i] = av;
    const unsigned la = (unsigned)(size_t)(lds) + tid * 16;
    const unsigned la8 = (unsigned)(size_t)(lds) + tid * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 36; ++m) {
            if (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(av[m & 3]), "v"(bv[m & 3]));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(av[m & 3]), "v"(bv[m & 3]));
            if (m < N) {
                const int q = m & 3;
                if (FILL == 1) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:64" :: "v"(la8), "v"(x[q]), "v"(x[(q + 1) & 3]) : "memory");
                if (FILL == 2) asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:64" :: "v"(la8), "v"(x[q][0]), "v"(x[q][1]) : "memory");
                if (FILL == 3) asm volatile("ds_write_b64 %0, %1" :: "v"(la8), "v"(x[q]) : "memory");
                if (FILL == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(r4[q]) : "v"(la));
                if (FILL == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[q]) : "v"(x[(q + 1) & 3]));
                if (FILL == 6) asm volatile("ds_write_b32 %0, %1" :: "v"(la8), "v"(x[q][0]) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 36; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    for (int i = 0; i < 4; ++i) s += x[i][0] + x[i][1] + r4[i][0] + r4[i][3];
    if (s == 123.456f) out[tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <bool AGPR, int FILL, int N>
void run(const char* name, const float* w, float* out, unsigned long long* cyc)
{
    const int iters = 2000;
    probe<AGPR, FILL, N><<<256, 512>>>(w, out, 4, cyc);
    (void)hipDeviceSynchronize();
    probe<AGPR, FILL, N><<<256, 512>>>(w, out, iters, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-5s %-34s N=%2d  %8.0f cycles per unit (floor 2304)\n", AGPR ? "AGPR" : "VGPR", name, N, (double)c / iters);
}

template <bool AGPR>
void all(const float* w, float* out, unsigned long long* cyc)
{
    run<AGPR, 0, 0>("MFMA only", w, out, cyc);
    run<AGPR, 1, 3>("ds_write2_b64", w, out, cyc);
    run<AGPR, 1, 9>("ds_write2_b64", w, out, cyc);
    run<AGPR, 2, 6>("ds_write2_b32", w, out, cyc);
    run<AGPR, 2, 18>("ds_write2_b32", w, out, cyc);
    run<AGPR, 3, 6>("ds_write_b64", w, out, cyc);
    run<AGPR, 3, 18>("ds_write_b64", w, out, cyc);
    run<AGPR, 6, 12>("ds_write_b32", w, out, cyc);
    run<AGPR, 6, 36>("ds_write_b32", w, out, cyc);
    run<AGPR, 4, 18>("ds_read_b128", w, out, cyc);
    run<AGPR, 4, 36>("ds_read_b128", w, out, cyc);
    run<AGPR, 5, 12>("v_pk_add_f32", w, out, cyc);
    run<AGPR, 5, 36>("v_pk_add_f32", w, out, cyc);
}

int main()
{
    float *w, *out; unsigned long long* cyc;
    (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 0, 1 << 20);
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 64);
    all<false>(w, out, cyc);
    all<true>(w, out, cyc);
    return 0;
}

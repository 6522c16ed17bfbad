// Developer microbenchmark 10: price of ONE instruction of each kind issued between two v_mfma_f32_32x32x2_f32 of a wave
// that is alone on its SIMD (the Winograd kernel's situation): cycles added per instruction = ticks per MFMA - 64.6.
// All fillers are inline asm (no compiler-inserted waits); memory results are drained once per 64 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int FILL, int PER>   // PER fillers after every MFMA
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ w, float* out, int iters, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[16];
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int i = tid; i < 16384; i += 256) lds[i] = 0.f;
    __syncthreads();
    f32x2 x[4];
    for (int i = 0; i < 4; ++i) x[i] = f32x2{(float)(tid + i) * 1e-3f, 1.f};
    float y[8];
    for (int i = 0; i < 8; ++i) y[i] = (float)(tid + i);
    f32x4 av = *reinterpret_cast<const f32x4*>(w + lane * 4), bv = *reinterpret_cast<const f32x4*>(w + 256 + lane * 4);
    f32x4 r4[4]; f32x2 r2[4];
    for (int i = 0; i < 4; ++i) { r4[i] = av; r2[i] = x[i]; }
    const unsigned la = (unsigned)(size_t)(lds) + tid * 16;       // conflict-free 16-byte slots (LDS address = low 32 bits)
    const unsigned la8 = (unsigned)(size_t)(lds) + tid * 8;
    const float* gp = w + tid * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            const int s = m >> 2, e = m & 3;
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int q = (m * PER + k) & 3;
                if (FILL == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[q]) : "v"(x[(q + 1) & 3]));
                if (FILL == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(y[(m * PER + k) & 7]) : "v"(y[(m * PER + k + 1) & 7]));
                if (FILL == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[q]) : "v"(x[(q + 1) & 3]));
                if (FILL == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[(m * PER + k) & 7]) : "v"(y[(m * PER + k + 1) & 7]));
                if (FILL == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(r4[q]) : "v"(la));
                if (FILL == 6) asm volatile("ds_read_b64 %0, %1" : "=v"(r2[q]) : "v"(la8));
                if (FILL == 7) asm volatile("ds_write_b64 %0, %1" :: "v"(la8), "v"(x[q]) : "memory");
                if (FILL == 8) asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(av) : "memory");
                if (FILL == 9) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r4[q]) : "v"(gp));
                if (FILL == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(y[(m * PER + k) & 7]) : "v"(y[(m * PER + k + 1) & 7]));
                if (FILL == 11) asm volatile("v_max_i32 %0, %0, %1" : "+v"(y[(m * PER + k) & 7]) : "v"(y[(m * PER + k + 1) & 7]));
                if (FILL == 12) asm volatile("s_nop 0");
                if (FILL == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y[(m * PER + k) & 7]) : "v"(y[(m * PER + k + 1) & 7]));
                if (FILL == 14) asm volatile("ds_read_b32 %0, %1" : "=v"(y[(m * PER + k) & 7]) : "v"(la8));
                if (FILL == 15) asm volatile("ds_write_b32 %0, %1" :: "v"(la8), "v"(y[q]) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 4; ++i) s += x[i][0] + x[i][1] + r4[i][0] + r4[i][3] + r2[i][0] + r2[i][1];
    for (int i = 0; i < 8; ++i) s += y[i];
    if (s == 123.456f) out[tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int FILL, int PER>
void run(const char* name, const float* w, float* out, unsigned long long* cyc)
{
    const int grid = 256, iters = 400;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<FILL, PER><<<grid, 256>>>(w, out, 4, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    probe<FILL, PER><<<grid, 256>>>(w, out, iters, cyc);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = (double)grid * 4 * iters * 64 * 4096.0;
    const double t = (double)c / (iters * 64.0);
    printf("%-22s x%d per MFMA: %.3f ms %5.1f%% of peak  %6.1f ticks/MFMA  -> %+5.1f ticks per instruction\n", name, PER, ms,
           flop / ms / 1e9 / 157.3 * 100, t, PER ? (t - 64.6) / PER : 0.0);
}

int main()
{
    float *w, *out; unsigned long long* cyc;
    (void)hipMalloc(&w, 65536); (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 64); (void)hipMemset(w, 0, 65536);
    run<0, 0>("none", w, out, cyc); run<0, 0>("none", w, out, cyc);
#define R(F, N) run<F, 1>(N, w, out, cyc); run<F, 2>(N, w, out, cyc); run<F, 4>(N, w, out, cyc);
    R(1, "v_pk_add_f32") R(2, "v_add_f32") R(3, "v_pk_fma_f32") R(4, "v_fma_f32") R(10, "v_mov_b32") R(11, "v_max_i32") R(13, "v_cndmask_b32")
    R(12, "s_nop") R(5, "ds_read_b128") R(6, "ds_read_b64") R(14, "ds_read_b32") R(7, "ds_write_b64") R(8, "ds_write_b128") R(15, "ds_write_b32")
    R(9, "global_load_dwordx4")
    return 0;
}

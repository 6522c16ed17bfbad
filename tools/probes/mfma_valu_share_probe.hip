// Developer probe: do vector instructions cost matrix-pipe time next to a stream of v_mfma_f32_16x16x4_f32?  One wave per SIMD
// (4 waves per workgroup, one workgroup per CU) issues 36 INDEPENDENT MFMAs per iteration (1152 cycles of the pipe) with K
// independent vector instructions of one class spread between them; ticks per iteration = 1152 + K * (cycles the class steals).
//   classes: v_pk_fma_f32 (the transforms' instruction), v_fma_f32, v_pk_add_f32, v_add_u32 (integer), v_mov_b32
// and the same with TWO waves per SIMD (8 waves: both stream MFMAs, 18 each, and both carry K / 2 vector instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NW, int K, int CLS>
__global__ __launch_bounds__(NW * 64, 1) void probe(float* out, unsigned long long* cyc, int iters)
{
    constexpr int NM = 36 * 4 / NW;             // MFMAs per wave and iteration (36 per SIMD)
    constexpr int KW = K * 4 / NW;              // vector instructions per wave and iteration (K per SIMD)
    f32x4 acc[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = (float)threadIdx.x, b = 1.5f;
    f32x2 v[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = f32x2{(float)i, (float)threadIdx.x}; u[i] = threadIdx.x + i; }
    const f32x2 kc = {1.0001f, 0.9999f};
    __shared__ __attribute__((aligned(16))) float lds[NW * 64 * 4 * 2];
    for (int i = threadIdx.x; i < NW * 64 * 8; i += NW * 64) lds[i] = (float)i;
    __syncthreads();
    const unsigned la = (unsigned)(size_t)(lds + threadIdx.x * 4);          // 16 bytes per lane: conflict-free for ds_read_b128
    const unsigned la8 = (unsigned)(size_t)(lds + threadIdx.x * 2);         // 8 bytes per lane: conflict-free for the 8-byte accesses
    f32x4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = f32x4{1.f, 2.f, 3.f, (float)i};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // vector instructions of slot i: KW spread over NM slots, eight independent chains
            constexpr int per = 1;
#pragma unroll
            for (int j = (i * KW) / NM; j < ((i + 1) * KW) / NM; ++j) {
                const int r = j & 7;
                if (CLS == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(kc));
                else if (CLS == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r][0]) : "v"(kc[0]));
                else if (CLS == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(kc));
                else if (CLS == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[r]) : "v"(u[(r + 1) & 7]));
                else if (CLS == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(u[r]) : "v"(u[(r + 1) & 7]));
                else if (CLS == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(q[r]) : "v"(la));
                else if (CLS == 6) asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:1" :: "v"(la8), "v"(v[r][0]), "v"(v[r][1]) : "memory");
                else if (CLS == 7) asm volatile("ds_read_b64 %0, %1" : "=v"(v[r]) : "v"(la8));
                else if (CLS == 8) asm volatile("ds_write_b64 %0, %1" :: "v"(la8), "v"(v[r]) : "memory");
                else asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:64" :: "v"(la8), "v"(v[r]), "v"(v[(r + 1) & 7]) : "memory");
            }
            (void)per;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (CLS >= 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1] + (float)u[i] + q[i][0] + q[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NW, int K, int CLS>
double run(float* out, unsigned long long* cyc)
{
    const int iters = 2000;
    probe<NW, K, CLS><<<256, NW * 64>>>(out, cyc, 50);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<NW, K, CLS><<<256, NW * 64>>>(out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)c;
    return ms * 1e6 / iters;          // ns per iteration
}

template <int NW, int CLS>
void row(const char* name, float* out, unsigned long long* cyc)
{
    const double n0 = run<NW, 0, CLS>(out, cyc), n12 = run<NW, 12, CLS>(out, cyc), n36 = run<NW, 36, CLS>(out, cyc), n72 = run<NW, 72, CLS>(out, cyc);
    // clock from the MFMA-only run: 1152 cycles per iteration
    const double ghz = 1152.0 / n0;
    printf("%d waves/SIMD  %-14s ns/iter K=0 %.0f  K=12 %.0f  K=36 %.0f  K=72 %.0f   -> pipe cycles per instruction (at %.2f GHz): %.1f  %.1f  %.1f\n",
           NW / 4, name, n0, n12, n36, n72, ghz, (n12 - n0) * ghz / 12, (n36 - n0) * ghz / 36, (n72 - n0) * ghz / 72);
}

int main()
{
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 1 << 16); (void)hipMalloc(&cyc, 64);
    row<4, 0>("v_pk_fma_f32", out, cyc); row<4, 1>("v_fma_f32", out, cyc); row<4, 2>("v_pk_add_f32", out, cyc);
    row<4, 3>("v_add_u32", out, cyc); row<4, 4>("v_mov_b32", out, cyc);
    row<8, 0>("v_pk_fma_f32", out, cyc); row<8, 1>("v_fma_f32", out, cyc); row<8, 2>("v_pk_add_f32", out, cyc);
    row<8, 3>("v_add_u32", out, cyc); row<8, 4>("v_mov_b32", out, cyc);
    row<4, 5>("ds_read_b128", out, cyc); row<4, 6>("ds_write2_b32", out, cyc); row<4, 7>("ds_read_b64", out, cyc); row<4, 8>("ds_write_b64", out, cyc);
    row<8, 5>("ds_read_b128", out, cyc); row<8, 6>("ds_write2_b32", out, cyc); row<8, 7>("ds_read_b64", out, cyc); row<8, 8>("ds_write_b64", out, cyc);
    row<4, 9>("ds_write2_b64", out, cyc); row<8, 9>("ds_write2_b64", out, cyc);
    return 0;
}

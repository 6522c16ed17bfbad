// Developer microbenchmark 9: does the ORDER of fp32 MFMAs over accumulators matter when other instructions sit between
// them?  The Winograd kernel issues 4 consecutive v_mfma_f32_32x32x2_f32 on the SAME accumulator (k = 0..3 of a position)
// with one memory / vector instruction between each pair; the direct kernel alternates between 4 accumulators.
// ORDER 0: acc[s] x4 in a row;  1: pairs (s, s+1) alternating;  2: quads alternating.
// FILL  0: none; 1: one v_pk_add_f32 per MFMA; 2: one ds_read_b128 per MFMA; 3: the kernel's mix per 4 MFMAs
//          (global_load_dwordx4, ds_read_b128, ds_read_b64 + 2 pk_add + ds_write_b64 every other group).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ORDER, int FILL>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ w, float* out, int iters, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[16];
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int i = tid; i < 8192; i += 256) lds[i] = 0.f;
    __syncthreads();
    f32x2 x[4];
    for (int i = 0; i < 4; ++i) x[i] = f32x2{(float)(tid + i) * 1e-3f, 1.f};
    f32x4 av = *reinterpret_cast<const f32x4*>(w + lane * 4), bv = *reinterpret_cast<const f32x4*>(w + 256 + lane * 4);
    f32x4 ld = av, g = bv;
    f32x2 l2 = x[0];
    const float* gp = w + lane * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            int s, e;
            if (ORDER == 0) { s = m >> 2; e = m & 3; }
            else if (ORDER == 1) { s = ((m >> 3) << 1) + (m & 1); e = (m >> 1) & 3; }
            else { s = ((m >> 4) << 2) + (m & 3); e = (m >> 2) & 3; }
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (FILL == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[m & 3]) : "v"(x[(m + 1) & 3]));
            if (FILL == 2) ld = *reinterpret_cast<volatile f32x4*>(&lds[(lane * 4 + m * 256) & 8191]);
            if (FILL == 3) {
                const int k = m & 3, grp = m >> 2;
                if (k == 0) g = *reinterpret_cast<const volatile f32x4*>(gp + ((grp * 1024) & 4095));
                else if (k == 1) ld = *reinterpret_cast<volatile f32x4*>(&lds[(lane * 4 + grp * 256) & 8191]);
                else if (k == 2) {
                    if (grp < 8) l2 = *reinterpret_cast<volatile f32x2*>(&lds[(lane * 2 + grp * 128) & 8191]);
                    else { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(x[1])); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[2]) : "v"(x[3]));
                           asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[1]) : "v"(x[2])); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[3]) : "v"(x[0])); }
                } else {
                    if (grp >= 8) *reinterpret_cast<volatile f32x2*>(&lds[(lane * 2 + grp * 128) & 8191]) = x[grp & 3];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        av[0] += g[0] * 0.f + l2[0] * 0.f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 4; ++i) s += x[i][0] + x[i][1];
    s += ld[0] + g[1] + l2[1];
    if (s == 123.456f) out[tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int ORDER, int FILL>
void run(const float* w, float* out, unsigned long long* cyc)
{
    const int grid = 256, iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<ORDER, FILL><<<grid, 256>>>(w, out, 4, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<ORDER, FILL><<<grid, 256>>>(w, out, iters, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = (double)grid * 4 * iters * 64 * 4096.0;
    printf("order %d fill %d: %.3f ms  %.1f TF/s (%.1f%% of 157.3)  %.1f memtime ticks per MFMA\n", ORDER, FILL, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 157.3 * 100, (double)c / (iters * 64.0) * 1.0);
}

int main()
{
    float *w, *out; unsigned long long* cyc;
    hipMalloc(&w, 65536); hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMemset(w, 0, 65536);
    run<0, 0>(w, out, cyc);
    run<0, 0>(w, out, cyc); run<1, 0>(w, out, cyc); run<2, 0>(w, out, cyc);
    run<0, 1>(w, out, cyc); run<1, 1>(w, out, cyc); run<2, 1>(w, out, cyc);
    run<0, 2>(w, out, cyc); run<1, 2>(w, out, cyc); run<2, 2>(w, out, cyc);
    run<0, 3>(w, out, cyc); run<1, 3>(w, out, cyc); run<2, 3>(w, out, cyc);
    return 0;
}

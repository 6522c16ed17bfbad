// NOTE: the conclusion this probe suggested (5-8 cycles per foreign VALU instruction next to an MFMA stream) was WRONG:
// waves 0-3 and 4-7 of one 512-thread workgroup do not share SIMDs the way it assumes.  mfma_probe7.hip places the
// streaming and the timed wave in different workgroups on the same CU and shows total starvation.  Kept for the record.
// Developer microbenchmark 6: how fast does a wave advance through ordinary instructions while the
// OTHER wave of its SIMD streams fp32 MFMAs back to back?  (explains the 25-60k-cycle prologues /
// epilogues measured with tools/conv_timing.py)   512 threads: waves 0-3 stream, waves 4-7 are timed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int PRIO>     // KIND 0: v_add_u32 chain, 1: independent v_add, 2: s_add chain, 3: mixed
__global__ __launch_bounds__(512, 2) void probe(float* out, unsigned long long* cyc, int stream_iters, int partner_streams)
{
    const int tid = threadIdx.x, wave = tid >> 6;
    if (wave < 4) {
        if (!partner_streams) return;
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        const float a0 = (float)tid, b0 = 1.f;
        for (int it = 0; it < stream_iters; ++it)
#pragma unroll
            for (int e = 0; e < 64; ++e) acc[e & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[e & 3], 0, 0, 0);
        float s = 0.f;
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        if (s == 123.456f) out[tid] = s;
        return;
    }
    // timed wave: wait a little so that the partner is streaming
    for (int i = 0; i < 20; ++i) __builtin_amdgcn_s_sleep(100);
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    int x = tid, y = tid * 3, z = tid * 5, w = tid * 7;
    int sx = blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 256; ++i) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
        if (KIND == 1) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(w)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"(w));
                         asm volatile("v_add_u32 %0, %0, %1" : "+v"(z) : "v"(w)); }
        if (KIND == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
        if (KIND == 3) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y)); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((tid & 63) == 0 && blockIdx.x < 64) cyc[blockIdx.x * 4 + (wave - 4)] = t1 - t0;
    if (x + y + z + sx == 123456789) out[tid] = 1.f;
}

template <int KIND, int PRIO>
void run(float* out, unsigned long long* cyc, int partner)
{
    hipMemset(cyc, 0, 64 * 4 * 8);
    probe<KIND, PRIO><<<256, 512>>>(out, cyc, 400, partner);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int i = 0; i < 256; ++i) if (h[i]) { s += (double)h[i]; ++n; }
    const int ninstr = 256 * (KIND == 1 ? 3 : KIND == 3 ? 2 : 1);
    printf("kind %d prio %d partner-streams %d : %.0f cycles for %d instr = %.1f cycles/instr (%s)\n", KIND, PRIO, partner, s / n, ninstr, s / n / ninstr,
           hipGetErrorString(hipGetLastError()));
}

int main()
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64 * 4 * 8);
    run<0, 0>(out, cyc, 0); run<0, 0>(out, cyc, 1); run<0, 1>(out, cyc, 1);
    run<1, 0>(out, cyc, 0); run<1, 0>(out, cyc, 1); run<1, 1>(out, cyc, 1);
    run<2, 0>(out, cyc, 0); run<2, 0>(out, cyc, 1); run<2, 1>(out, cyc, 1);
    run<3, 0>(out, cyc, 0); run<3, 0>(out, cyc, 1); run<3, 1>(out, cyc, 1);
    return 0;
}

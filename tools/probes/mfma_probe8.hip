// Developer microbenchmark 8: what does s_memtime count, and what clock does the chip sustain under a full fp32-MFMA
// load?  Every SIMD of every CU streams independent v_mfma_f32_32x32x2_f32 (2 waves per SIMD) for ~10 ms; compares
// s_memtime ticks with hipEvent wall time and with the known MFMA count.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int F16>
__global__ __launch_bounds__(256, 2) void stream(float* out, unsigned long long* ticks, int iters)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float a0 = (float)threadIdx.x, b0 = 1.f;
    h8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(threadIdx.x & 7); bh[i] = (_Float16)1; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int e = 0; e < 64; ++e) {
            if (F16) acc[e & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[e & 3], 0, 0, 0);
            else acc[e & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[e & 3], 0, 0, 0);
        }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// fp32 MFMA stream that also moves operands like the conv loop: per 16 MFMAs MODE&1: 2 ds_read_b128 (LDS),
// MODE&2: 2 global_load_dwordx4 from a 1.2 MB L2-resident buffer (the weight stream), both fed to the MFMAs
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void stream_ops(const float* __restrict__ wsrc, float* out, unsigned long long* ticks, int iters)
{
    __shared__ float lds[12288];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 12288; i += 256) lds[i] = (float)(i & 15);
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 av[2], bv[2];
    av[0] = av[1] = bv[0] = bv[1] = f32x4{1.f, 2.f, 3.f, 4.f};
    const f32x4* wp = reinterpret_cast<const f32x4*>(wsrc) + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            f32x4 an[2] = {av[0], av[1]}, bn[2] = {bv[0], bv[1]};
            if (MODE & 1) {
                an[0] = *reinterpret_cast<const f32x4*>(&lds[(lane * 36 + st * 8 + (it & 7) * 144) % 12000]);
                an[1] = *reinterpret_cast<const f32x4*>(&lds[(lane * 36 + st * 8 + 4608 + (it & 7) * 144) % 12000]);
            }
            if (MODE & 2) {
                bn[0] = wp[((it * 4 + st) & 2047) * 128];
                bn[1] = wp[((it * 4 + st) & 2047) * 128 + 64];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t >> 1][kk], bv[t & 1][kk], acc[t], 0, 0, 0);
            av[0] = an[0]; av[1] = an[1]; bv[0] = bn[0]; bv[1] = bn[1];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run_ops(const float* w, float* out, unsigned long long* ticks, int iters, const char* what)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    stream_ops<MODE><<<512, 256>>>(w, out, ticks, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    stream_ops<MODE><<<512, 256>>>(w, out, ticks, iters);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    static unsigned long long h[512];
    hipMemcpy(h, ticks, 512 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 512; ++i) mean += (double)h[i]; mean /= 512;
    const double mfma_per_wave = (double)iters * 64;
    printf("%-34s 512 workgroups: %.3f ms, %.3f GHz tick rate; %.1f ticks per MFMA per wave (128 = pipe saturated); %.1f TFLOP/s\n",
           what, ms, mean / ms * 1e-6, mean / mfma_per_wave, mfma_per_wave * 512 * 4 * 4096.0 / ms * 1e-9);
}

template <int F16>
void run(float* out, unsigned long long* ticks, int nblk, int iters, const char* what)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    stream<F16><<<nblk, 256>>>(out, ticks, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    stream<F16><<<nblk, 256>>>(out, ticks, iters);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    static unsigned long long h[512];
    hipMemcpy(h, ticks, nblk * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < nblk; ++i) mean += (double)h[i]; mean /= nblk;
    const double mfma_per_wave = (double)iters * 64;
    const double flop = mfma_per_wave * nblk * 4 * (F16 ? 32768.0 : 4096.0);
    printf("%-34s %3d workgroups: %.3f ms, %.0f ticks per workgroup -> %.3f GHz tick rate; %.1f ticks per MFMA per wave; %.1f TFLOP/s\n",
           what, nblk, ms, mean, mean / ms * 1e-6, mean / mfma_per_wave, flop / ms * 1e-9);
}

int main()
{
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 1 << 20); hipMalloc(&ticks, 512 * 8);
    run<0>(out, ticks, 256, 1200, "fp32 MFMA, 1 wave per SIMD");
    run<0>(out, ticks, 512, 1200, "fp32 MFMA, 2 waves per SIMD");
    run<0>(out, ticks, 32, 1200, "fp32 MFMA, 32 workgroups only");
    float* w; hipMalloc(&w, 2048 * 128 * 16 + 4096); hipMemset(w, 0, 2048 * 128 * 16 + 4096);
    run_ops<0>(w, out, ticks, 1200, "fp32 MFMA only (loop form)");
    run_ops<1>(w, out, ticks, 1200, "fp32 MFMA + LDS operand reads");
    run_ops<2>(w, out, ticks, 1200, "fp32 MFMA + L2 weight stream");
    run_ops<3>(w, out, ticks, 1200, "fp32 MFMA + LDS + L2 (conv loop)");
    run<1>(out, ticks, 512, 2400, "f16 MFMA, 2 waves per SIMD");
    run<1>(out, ticks, 32, 2400, "f16 MFMA, 32 workgroups only");
    return 0;
}

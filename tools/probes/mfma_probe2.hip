// Developer microbenchmark 2: cost of streaming MFMA B-fragments from L2 inside the loop.
//  MODE 0: no B loads (upper bound)    MODE 1: 2x global_load_dwordx4 -> VGPR per step (current kernel)
//  MODE 2: 1x global_load_dwordx4 per step (half the loads)
//  MODE 3: 2x global_load_lds_dwordx4 (LDS-DMA) + ds_read_b128
//  MODE 4: like 1 but loads issued in the middle of the MFMA group
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 3) void probe(const float* __restrict__ w, float* out, int iters, int wchunks)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8192; i += 256) lds[i] = __uint_as_float(0x3f000000u | ((i * 2654435761u) >> 9)) - 0.75f;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const f32x4* wp0 = reinterpret_cast<const f32x4*>(w) + lane;
    const int ab = (lane & 31) * 36 + (lane >> 5) * 4;
    float* ring = lds + 2048 + wave * 1536;          // 3 slots x 512 floats per wave (MODE 3)
    f32x4 a0 = *reinterpret_cast<f32x4*>(&lds[ab]), a1 = *reinterpret_cast<f32x4*>(&lds[ab + 1224]);
    f32x4 b0 = wp0[0], b1 = wp0[64], c0 = wp0[128], c1 = wp0[192];
    for (int it = 0; it < iters; ++it) {
        int zoff = 0; asm volatile("" : "+v"(zoff));
        const f32x4* wp = wp0 + zoff + (it % wchunks) * (36 * 128);
        if (MODE == 5) { b0 = wp[0]; b1 = wp[64]; c0 = wp[128]; c1 = wp[192]; a0 = *reinterpret_cast<f32x4*>(&lds[ab + (it & 7) * 4]); a1 = *reinterpret_cast<f32x4*>(&lds[ab + 1224 + (it & 7) * 4]); }
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            f32x4 na0, na1, n0 = c0, n1 = c1;
            na0 = *reinterpret_cast<f32x4*>(&lds[ab + ((s * 36 + 8) & 1023)]);
            na1 = *reinterpret_cast<f32x4*>(&lds[ab + 1224 + ((s * 36) & 1023)]);
            if (MODE == 1 || MODE == 5) { n0 = wp[(s * 128)]; n1 = wp[s * 128 + 64]; }
            if (MODE == 2) { n0 = wp[(s * 128)]; }
            if (MODE == 3) {
                float* slot = ring + ((s + 2) % 3) * 512;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + s * 128),
                                                 (__attribute__((address_space(3))) void*)slot, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + s * 128 + 64),
                                                 (__attribute__((address_space(3))) void*)(slot + 256), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                float* rs = ring + (s % 3) * 512;
                n0 = *reinterpret_cast<f32x4*>(&rs[lane * 4]);
                n1 = *reinterpret_cast<f32x4*>(&rs[256 + lane * 4]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[1], 0, 0, 0);
                if (MODE == 4 && e == 1) { __builtin_amdgcn_sched_barrier(0); n0 = wp[(s * 128)]; n1 = wp[s * 128 + 64]; __builtin_amdgcn_sched_barrier(0); }
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = c0; b1 = c1; c0 = n0; c1 = n1;
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[tid] = s;
}

template <int MODE>
void run(const float* w, float* out, int bpc, int iters, int wchunks = 1)
{
    const int lds = bpc == 1 ? 150000 : bpc == 2 ? 80000 : 50000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = 256 * bpc;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<grid, 256, lds>>>(w, out, 2, wchunks);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<grid, 256, lds>>>(w, out, iters, wchunks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * 4 * iters * 36 * 16 * 4096.0;
    printf("wchunks=%d MODE=%d blocks/CU=%d  %.3f ms  %.1f TF/s (%.1f%%)  %s\n", wchunks, MODE, bpc, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100,
           hipGetErrorString(hipGetLastError()));
}

int main()
{
    float *w, *out;
    const size_t n = 36 * 128 * 16 * 16 + 4096;
    hipMalloc(&w, n * 4); hipMalloc(&out, 4096);
    std::vector<float> hw(n); unsigned x = 12345;
    for (auto& v : hw) { x = x * 1664525u + 1013904223u; v = ((x >> 8) / 16777216.0f - 0.5f) * 0.2f; }
    hipMemcpy(w, hw.data(), n * 4, hipMemcpyHostToDevice);
    run<1>(w, out, 3, 300, 1);
    for (int bpc = 1; bpc <= 3; ++bpc) { run<1>(w, out, bpc, 304, 2); run<5>(w, out, bpc, 304, 2); }
    return 0;
}

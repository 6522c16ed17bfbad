// Developer microbenchmark 7: why does a wave in the real conv kernel crawl (≈84 cycles per VALU instruction)
// next to a streaming partner when mfma_probe6 measured 5-8?  Vary what differs: the partner is a different
// WORKGROUP (role from a per-CU arrival counter), the partner's loop also issues ds_read_b128 / global loads /
// s_waitcnt like the real loop, and the number of registers the kernel is compiled for.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LOADS: partner issues 2 ds_read_b128 + 2 global_load_dwordx4 per 16 MFMAs and feeds them to the MFMAs
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
template <int LOADS, int SEPARATE, int PRIO = 0, int REVERSE = 0, int F16 = 0>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ wsrc, float* out, unsigned long long* cyc,
                                                int* arrival, int stream_iters)
{
    __shared__ float lds[16384];
    __shared__ int role_s;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) lds[i] = (float)i;
    if (tid == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned slot = (((xcc & 15) * 8 + se) * 2 + sh) * 16 + cu;
        role_s = atomicAdd(&arrival[slot & 4095], 1);
    }
    __syncthreads();
    const int role = SEPARATE ? ((role_s & 1) ^ REVERSE) : 0;      // 0: stream, 1: timed
    if (REVERSE && role == 0) for (int i = 0; i < 10; ++i) __builtin_amdgcn_s_sleep(100);   // younger streamer starts first anyway
    if (PRIO == 1 && role == 1) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2 && role == 0) __builtin_amdgcn_s_setprio(3);
    if (role == 0) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        f32x4 av[2], bv[2];
        av[0] = av[1] = bv[0] = bv[1] = f32x4{1.f, 2.f, 3.f, 4.f};
        const f32x4* wp = reinterpret_cast<const f32x4*>(wsrc) + lane;
        for (int it = 0; it < stream_iters; ++it) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f32x4 an[2], bn[2];
                if (LOADS) {
                    an[0] = *reinterpret_cast<const f32x4*>(&lds[(lane * 36 + st * 8) & 16383]);
                    an[1] = *reinterpret_cast<const f32x4*>(&lds[(lane * 36 + st * 8 + 4608) & 16383]);
                    bn[0] = wp[(it * 4 + st) * 128];
                    bn[1] = wp[(it * 4 + st) * 128 + 64];
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (F16) {
                            h8v ah, bh;
                            __builtin_memcpy(&ah, &av[t >> 1], 16); __builtin_memcpy(&bh, &bv[t & 1], 16);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
                        } else {
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t >> 1][kk], bv[t & 1][kk], acc[t], 0, 0, 0);
                        }
                    }
                if (LOADS) { av[0] = an[0]; av[1] = an[1]; bv[0] = bn[0]; bv[1] = bn[1]; }
            }
        }
        float s = 0.f;
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        if (s == 123.456f) out[tid] = s;
        if (SEPARATE) return;
    }
    if (!SEPARATE && role == 0) { /* fallthrough: same block also measures afterwards (unobstructed) */ }
    if (SEPARATE) for (int i = 0; i < (REVERSE ? 60 : 20); ++i) __builtin_amdgcn_s_sleep(100);
    int x = tid;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 256; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(tid));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
    if (x == 123456789) out[tid] = 1.f;
}

template <int LOADS, int SEPARATE, int PRIO = 0, int REVERSE = 0, int F16 = 0>
void run(const float* w, float* out, unsigned long long* cyc, int* arrival, const char* what)
{
    hipMemset(cyc, 0, 512 * 4 * 8); hipMemset(arrival, 0, 4096 * 4);
    probe<LOADS, SEPARATE, PRIO, REVERSE, F16><<<512, 256>>>(w, out, cyc, arrival, 2000);
    hipDeviceSynchronize();
    static unsigned long long h[2048];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; int n = 0; unsigned long long mn = ~0ull, mx = 0;
    for (int i = 0; i < 2048; ++i) if (h[i]) { s += (double)h[i]; ++n; if (h[i] < mn) mn = h[i]; if (h[i] > mx) mx = h[i]; }
    printf("%-44s timed waves %4d: mean %.1f cycles/instr  (min %.1f max %.1f)  %s\n", what, n, n ? s / n / 256 : 0.0, mn / 256.0, mx / 256.0,
           hipGetErrorString(hipGetLastError()));
}

int main()
{
    float *w, *out; unsigned long long* cyc; int* arrival;
    hipMalloc(&w, 64 << 20); hipMemset(w, 0, 64 << 20);
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 512 * 4 * 8); hipMalloc(&arrival, 4096 * 4);
    run<0, 0>(w, out, cyc, arrival, "no partner (after own stream)");
    run<0, 1>(w, out, cyc, arrival, "partner workgroup streams, MFMA only");
    run<1, 1>(w, out, cyc, arrival, "partner workgroup streams, MFMA + loads");
    run<0, 1, 1>(w, out, cyc, arrival, "MFMA only, timed wave s_setprio 3");
    run<1, 1, 1>(w, out, cyc, arrival, "MFMA + loads, timed wave s_setprio 3");
    run<0, 1, 2>(w, out, cyc, arrival, "MFMA only, STREAM wave s_setprio 3");
    run<0, 1, 0, 1>(w, out, cyc, arrival, "MFMA only, timed wave is the OLDER workgroup");
    run<1, 1, 0, 1>(w, out, cyc, arrival, "MFMA + loads, timed wave is the OLDER workgroup");
    run<0, 1, 0, 0, 1>(w, out, cyc, arrival, "f16 MFMA only (32x32x16), partner workgroup");
    run<1, 1, 0, 0, 1>(w, out, cyc, arrival, "f16 MFMA + loads, partner workgroup");
    return 0;
}

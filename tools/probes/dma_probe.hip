// Developer probe: LDS-DMA (global_load_lds_dwordx4, SGPR base + VGPR offset form) semantics on gfx950:
//  (1) destination = M0 + lane*16 for LDS byte addresses beyond 64 KiB; inactive lanes write nothing;
//  (2) price of one DMA per MFMA next to a wave streaming fp32 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const float* sbase, unsigned voff, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

__global__ __launch_bounds__(256, 1) void sem(const float* __restrict__ src, float* out)
{
    __shared__ __attribute__((aligned(16))) float lds[40000];      // 160000 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 40000; i += 256) lds[i] = -1.f;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds;
    // wave w copies 1 KiB from src + w*4096 floats to LDS byte offsets {1024*w, 70000-ish, 150000-ish}
    const unsigned dsts[3] = {1024u * wave, 69632u + 1024u * wave, 151552u + 1024u * wave};
    for (int k = 0; k < 3; ++k) {
        const float* sb = src + wave * 4096 + k * 1024;
        if (k == 2) { if (lane < 40) dma16(sb, lane * 16, base + dsts[k]); }      // partial exec mask
        else dma16(sb, lane * 16, base + dsts[k]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 40000; i += 256) out[i] = lds[i];
    if (tid == 0) out[40000] = (float)base;
}

template <int PER>
__global__ __launch_bounds__(256, 1) void price(const float* __restrict__ w, float* out, int iters, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc[16];
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 av = *reinterpret_cast<const f32x4*>(w + lane * 4), bv = *reinterpret_cast<const f32x4*>(w + 256 + lane * 4);
    const unsigned base = (unsigned)(size_t)lds + wave * 16384;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            const int s = m >> 2, e = m & 3;
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (PER == 1 || (PER == 4 && (m & 3) == 0) || (PER == 8 && (m & 7) == 0))
                dma16(w + ((m * 256) & 8191), lane * 16, base + ((m & 15) << 10));
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    s += lds[tid];
    if (s == 123.456f) out[tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int PER>
void run(const float* w, float* out, unsigned long long* cyc)
{
    const int iters = 400;
    price<PER><<<256, 256>>>(w, out, 4, cyc);
    (void)hipDeviceSynchronize();
    price<PER><<<256, 256>>>(w, out, iters, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double t = (double)c / (iters * 64.0);
    const int n = PER == 0 ? 0 : 64 / PER;
    printf("one DMA per %d MFMA: %.1f ticks/MFMA -> %+.1f ticks per DMA\n", PER, t, n ? (t - 64.2) * 64.0 / n : 0.0);
}

int main()
{
    float *src, *out; unsigned long long* cyc;
    (void)hipMalloc(&src, 1 << 20); (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 64);
    std::vector<float> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    (void)hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
    sem<<<1, 256>>>(src, out);
    (void)hipDeviceSynchronize();
    std::vector<float> o(40001);
    (void)hipMemcpy(o.data(), out, 40001 * 4, hipMemcpyDeviceToHost);
    printf("LDS base address of the array: %.0f\n", o[40000]);
    int bad = 0, written = 0;
    for (int i = 0; i < 40000; ++i) {
        float want = -1.f;
        for (int w = 0; w < 4; ++w) {
            const int d[3] = {1024 * w, 69632 + 1024 * w, 151552 + 1024 * w};
            for (int k = 0; k < 3; ++k) {
                const int lo = d[k] / 4, n = (k == 2 ? 40 * 4 : 256);
                if (i >= lo && i < lo + n) want = (float)(w * 4096 + k * 1024 + (i - lo));
            }
        }
        if (o[i] != -1.f) ++written;
        if (o[i] != want) { if (bad < 8) printf("  mismatch at float %d: got %.0f want %.0f\n", i, o[i], want); ++bad; }
    }
    printf("semantics: %d floats written, %d mismatches (expect 2688 written, 0 mismatches)\n", written, bad);
    run<0>(src, out, cyc); run<8>(src, out, cyc); run<4>(src, out, cyc); run<1>(src, out, cyc);
    return 0;
}

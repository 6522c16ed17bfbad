// Developer probe: does the instruction's immediate offset of global_load_lds_dwordx4 move the SOURCE address only, or the LDS
// destination as well?  (conv_wino43b.hip folds a wave's weight-block index into the immediate.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64, 1) void k(const float* __restrict__ src, float* out)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = -1.f;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds + 4096u;       // destination: float 1024
    unsigned keep;
    const float* sb = src + 2048;                               // source base: float 2048
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"((unsigned)lane * 16u), "s"(sb), "s"(base) : "memory");
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:-2048\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"((unsigned)lane * 16u), "s"(sb), "s"(base + 8192u) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 4096; i += 64) out[i] = lds[i];
}
int main()
{
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (float)i;
    float *d, *o; (void)hipMalloc(&d, 8192 * 4); (void)hipMalloc(&o, 4096 * 4);
    (void)hipMemcpy(d, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o);
    std::vector<float> r(4096); (void)hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
    int first = -1, firstb = -1;
    for (int i = 0; i < 3072 && first < 0; ++i) if (r[i] >= 0.f) first = i;
    for (int i = 3072; i < 4096 && firstb < 0; ++i) if (r[i] >= 0.f) firstb = i;
    printf("offset:+1024 -> first written float %d (1024 = destination unmoved, 1280 = moved) holds source float %g (2304 = source moved by +1024 B)\n", first, first >= 0 ? r[first] : -1.f);
    printf("offset:-2048 -> first written float %d (3072 = destination unmoved) holds source float %g (1536 = source moved by -2048 B)\n", firstb, firstb >= 0 ? r[firstb] : -1.f);
    int n = 0; for (int i = 0; i < 4096; ++i) n += r[i] >= 0.f;
    printf("floats written: %d (512 expected)\n", n);
    return 0;
}

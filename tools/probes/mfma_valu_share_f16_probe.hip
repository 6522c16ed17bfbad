// Developer probe (round 5): what does a vector instruction cost next to a stream of v_mfma_f32_32x32x16_f16 (32 cycles each)?
// Per iteration a SIMD issues 36 independent-enough MFMAs (4 accumulator chains: 1152 cycles of the pipe) and K vector
// instructions of one class, eight independent chains.  Three arrangements:
//   MODE 0, 4 waves : one wave per SIMD carries both streams, the vector instructions spread between its own MFMAs
//   MODE 0, 8 waves : two waves per SIMD, each 18 MFMAs + K / 2 vector instructions
//   MODE 1, 8 waves : waves 0-3 issue ONLY the 36 MFMAs, waves 4-7 ONLY the K vector instructions (the fp16 convolution's
//                     "one group multiplies while the other runs its epilogue / tile production" premise)
// ns per iteration = 1152 cycles / clock + K x (what the class steals from the matrix pipe).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int NW, int K, int CLS, int MODE>
__global__ __launch_bounds__(NW * 64, 1) void probe(float* out, int iters)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || wave < 4, do_vec = MODE == 0 || wave >= 4;
    constexpr int NM = MODE == 1 ? 36 : 36 * 4 / NW;       // MFMAs per (multiplying) wave and iteration
    constexpr int KW = MODE == 1 ? K : K * 4 / NW;          // vector instructions per (vector) wave and iteration
    constexpr int NS = NM > KW ? NM : (KW > 0 ? KW : 1);    // slots
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); b[e] = (_Float16)(0.02f * (threadIdx.x - e)); }
    f32x2 v[8]; h2 hh[8]; unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = f32x2{(float)i, (float)threadIdx.x}; hh[i] = h2{(_Float16)i, (_Float16)1}; u[i] = threadIdx.x + i; }
    const f32x2 kc = {1.0001f, 0.9999f};
    __shared__ __attribute__((aligned(16))) float lds[8 * 64 * 4 * 2];
    for (int i = threadIdx.x; i < 8 * 64 * 8; i += NW * 64) lds[i] = (float)i;
    __syncthreads();
    const unsigned la = (unsigned)(size_t)(lds + threadIdx.x * 4), la8 = (unsigned)(size_t)(lds + threadIdx.x * 2);
    f32x4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = f32x4{1.f, 2.f, 3.f, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (do_mfma && i < NM) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (do_vec) {
#pragma unroll
                for (int j = (i * KW) / NS; j < ((i + 1) * KW) / NS; ++j) {
                    const int r = j & 7;
                    if (CLS == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(kc));
                    else if (CLS == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hh[r]) : "v"(v[r][0]), "v"(v[r][1]));
                    else if (CLS == 2) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(hh[r]) : "v"(hh[(r + 1) & 7]));
                    else if (CLS == 3) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[r][0]) : "v"(hh[r]));
                    else if (CLS == 4) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(hh[r]) : "v"(hh[(r + 1) & 7]), "v"(kc[0]), "v"(kc[1]));
                    else if (CLS == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(kc));
                    else if (CLS == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(u[r]) : "v"(u[(r + 1) & 7]));
                    else if (CLS == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[r]) : "v"(u[(r + 1) & 7]));
                    else if (CLS == 8) asm volatile("ds_read_b128 %0, %1" : "=v"(q[r]) : "v"(la));
                    else if (CLS == 9) asm volatile("ds_write_b64 %0, %1" :: "v"(la8), "v"(v[r]) : "memory");
                    else if (CLS == 10) asm volatile("v_max_f16 %0, %0, %1" : "+v"(hh[r]) : "v"(hh[(r + 1) & 7]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r][0]) : "v"(kc[0]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (CLS == 8 || CLS == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1] + (float)u[i] + q[i][0] + q[i][3] + (float)hh[i][0] + (float)hh[i][1];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int NW, int K, int CLS, int MODE>
double run(float* out)
{
    const int iters = 2000;
    probe<NW, K, CLS, MODE><<<256, NW * 64>>>(out, 50);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<NW, K, CLS, MODE><<<256, NW * 64>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;
}

template <int NW, int CLS, int MODE>
void row(const char* name, float* out)
{
    const double n0 = run<NW, 0, CLS, MODE>(out), n36 = run<NW, 36, CLS, MODE>(out), n72 = run<NW, 72, CLS, MODE>(out), n144 = run<NW, 144, CLS, MODE>(out),
                 n288 = run<NW, 288, CLS, MODE>(out);
    const double ghz = 1152.0 / n0;
    printf("%s %d waves/SIMD  %-18s ns/iter K=0 %.0f  K=36 %.0f  K=72 %.0f  K=144 %.0f  K=288 %.0f  -> cycles per instruction (at %.2f GHz): %.1f  %.1f  %.1f  %.1f\n",
           MODE ? "split " : "shared", NW / 4, name, n0, n36, n72, n144, n288, ghz, (n36 - n0) * ghz / 36, (n72 - n0) * ghz / 72, (n144 - n0) * ghz / 144,
           (n288 - n0) * ghz / 288);
}

#define ROWS(NW, MODE)                                                                                                     \
    row<NW, 0, MODE>("v_pk_fma_f32", out); row<NW, 1, MODE>("v_cvt_pk_f16_f32", out); row<NW, 2, MODE>("v_pk_max_f16", out);      \
    row<NW, 3, MODE>("v_cvt_f32_f16", out); row<NW, 4, MODE>("v_fma_mixlo_f16", out); row<NW, 5, MODE>("v_pk_add_f32", out);       \
    row<NW, 6, MODE>("v_mov_b32", out); row<NW, 7, MODE>("v_add_u32", out); row<NW, 11, MODE>("v_fma_f32", out);                    \
    row<NW, 8, MODE>("ds_read_b128", out); row<NW, 9, MODE>("ds_write_b64", out);

int main()
{
    float* out;
    (void)hipMalloc(&out, 1 << 16);
    ROWS(4, 0)
    ROWS(8, 0)
    ROWS(8, 1)
    return 0;
}

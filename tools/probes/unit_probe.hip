// Developer probe 13: the WHOLE unit body of the F(4x4,3x3) kernel (conv_wino43.hip) as a synthetic loop, in two geometries:
//   NW = 8: the shipped one -- two waves per SIMD, a wave = 16 tiles x 16 couts, 36 MFMAs + 18 fragment reads + its share of the
//           input transform (24 packed VALU, 6 + 6 reads, 3 + 6 stores) + 7 LDS-DMAs per unit, one barrier;
//   NW = 4: ONE wave per SIMD, a wave = 16 tiles x 32 couts -- 72 MFMAs on 288 accumulator registers, one V fragment shared by
//           two MFMAs (27 fragment reads), twice the transform share, 12 LDS-DMAs.
// Same LDS layout, addresses and DMA sources (weights L2-hot, patches streamed) as the kernel; results are meaningless.
// ELIM bits switch classes of work off: 1 transform, 2 DMA, 4 fragment reads, 8 barrier/wait.
// Prints ns per unit and CU and the matrix-pipe share that implies (2304 cycles of MFMA per unit at the measured clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NPIX = 18 * 34, PX = 34;
constexpr int VB4 = 4 * 32 * 36, UB4 = 4 * 64 * 36, RB4 = 11 * 64 * 4, SW4 = 8 * 36 * 2;

__device__ __forceinline__ void dma16(const float* sbase, unsigned voff_bytes, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }
template <int HI>
__device__ __forceinline__ f32x2 pk_fma_k(f32x2 a, unsigned long long k, f32x2 c)
{
    f32x2 d;
    if (HI) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}
template <int HI>
__device__ __forceinline__ f32x2 pk_fnma_k(f32x2 a, unsigned long long k, f32x2 c)
{
    f32x2 d;
    if (HI) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}
__device__ __forceinline__ void bt6(const f32x2 d[6], f32x2 r[6], unsigned long long K_A2B2)
{
    const unsigned long long K_PS = K_A2B2 + 0x0010000000100000ull, K_AB = K_A2B2 + 0x0020000000200000ull;
    const f32x2 t0 = pk_fnma_k<1>(d[2], K_A2B2, d[4]);
    const f32x2 t1 = pk_fnma_k<1>(d[1], K_A2B2, d[3]);
    const f32x2 t2 = pk_fnma_k<0>(d[2], K_A2B2, d[4]);
    const f32x2 t3 = pk_fnma_k<0>(d[1], K_A2B2, d[3]);
    r[0] = pk_fma_k<0>(d[0], K_PS, pk_fnma_k<1>(d[2], K_PS, d[4]));
    r[1] = pk_fma_k<0>(t1, K_AB, t0);
    r[2] = pk_fnma_k<0>(t1, K_AB, t0);
    r[3] = pk_fma_k<1>(t3, K_AB, t2);
    r[4] = pk_fnma_k<1>(t3, K_AB, t2);
    r[5] = pk_fma_k<0>(d[1], K_PS, pk_fnma_k<1>(d[3], K_PS, d[5]));
}

template <int NW, int ELIM, int VAR = 0>
__global__ __launch_bounds__(NW * 64, 1) void unit_probe(const float* __restrict__ wsrc, const float* __restrict__ rsrc, long long rsrc_floats,
                                                       float* out, int units, unsigned long long* cyc)
{
    constexpr int NT = NW * 64;
    constexpr bool DUTY = (VAR & 16) != 0;      // NW = 8: waves 0-3 transform in even units, waves 4-7 in odd units (two tasks per lane)
    constexpr int TW = DUTY ? 2 : 8 / NW;       // transform tasks per lane (512 lane-tasks per unit)
    constexpr int NM = 8 / NW;                  // MFMAs per position and wave
    // design A (VAR & 128): V' [buf][wave][12 blocks of 64 dwords (+ skew)] replaces V and the transform scratch; it sits at LDS
    // offset 0 (ds_write_addtid_b32 takes its base from M0[15:0])
    constexpr bool DA = (VAR & 128) != 0;
    constexpr int WR = 788;                      // dwords per wave region: 12 * 64 + row skew, and WR / 4 = 1 (mod 4)
    constexpr int VPB = 8 * WR;                  // dwords per V' buffer
    constexpr int V2B = 2 * 32 * 74 * 4;         // bytes per V2 buffer (fits the 18432-byte V buffer + slack? 18944: the probe's V area is 2 * VB4 floats = 36864 bytes: ok)
    constexpr int SCR = (VAR & 1024) ? 0 : 8 * SW4;       // (the half-window transform has no scratch)
    __shared__ __attribute__((aligned(16))) float smem[(DA ? 2 * VPB : 2 * VB4 + SCR) + 2 * UB4 + 3 * RB4 + 256];
    float* const Vs = smem;                      // (design A: V')
    float* const scr = smem + 2 * VB4;           // (legacy only)
    float* const Us = smem + (DA ? 2 * VPB : 2 * VB4 + SCR);
    float* const raw = Us + 2 * UB4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (int)(sizeof(smem) / 4); i += NT) smem[i] = 0.f;
    __syncthreads();
    const int tb = wave & 1, cb = wave >> 1;
    // operands (NW = 4: channel block = 32 couts = two 16-cout fragments 16 * 36 floats apart)
    const int a_base = ((lane >> 4) * 64 + cb * (16 * NM) + (lane & 15)) * 36;
    const int b_base = ((lane >> 4) * 32 + tb * 16 + (lane & 15)) * 36;
    // design A: lane l = (ch = l >> 4 = 2 cp + c, tile = tb * 16 + (l & 15)); V'[W = tile >> 2][block (j, c)][(tile & 3) * 16 + cp * 8 + r]
    const unsigned vp_lane = lds_addr(Vs) + (unsigned)(((tb * 4 + ((lane & 15) >> 2)) * WR + ((lane >> 4) & 1) * 64 + (lane & 3) * 16 + (lane >> 5) * 8) * 4);
    typedef const __attribute__((address_space(3))) f32x4* lds_quad_ptr0;
    typedef const __attribute__((address_space(3))) f32x2* lds_pair_ptr0;
    constexpr bool UG = (VAR & 256) != 0;       // U fragments by global_load_dwordx4 (L1 / L2) instead of LDS-DMA + ds_read
    f32x4 af[3][NM], bf[3];
    const float* const ug_lane = wsrc + (cb * 16 * NM + (lane & 15)) * 36 * 4 + (lane >> 4) * 4;     // (any 16-byte pattern the two tile-block waves share)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        bf[i] = f32x4{1.f, 2.f, 3.f, (float)tid};
#pragma unroll
        for (int m = 0; m < NM; ++m) af[i][m] = f32x4{1.f, 2.f, (float)lane, 4.f};
    }
    f32x4 acc[36][NM];
#pragma unroll
    for (int s = 0; s < 36; ++s)
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[s][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // transform tasks
    unsigned p1_base[TW], scr_w[TW], scr_r[TW], p2_addr[TW], p2v2[TW];
#pragma unroll
    for (int k = 0; k < TW; ++k) {
        const int t = DUTY ? (tid & 255) + 256 * k : tid + NT * k;
        const int win = t >> 3, sub = t & 7, sub6 = sub < 6 ? sub : 5;
        const int w_tile = win >> 1, w_cp = win & 1, w_ty = w_tile / 8, w_tx = w_tile % 8;
        p1_base[k] = lds_addr(raw) + (unsigned)(((4 * w_ty) * PX + 4 * w_tx + sub6) * 4 + 2 * w_cp) * 4u;
        float* const myscr = scr + (t >> 6) * SW4 + (win & 7) * 72;
        scr_w[k] = lds_addr(myscr + sub6 * 2);
        scr_r[k] = lds_addr(myscr + sub6 * 12);
        p2_addr[k] = lds_addr(Vs) + (unsigned)((2 * w_cp * 32 + w_tile) * 36 + 6 * sub6) * 4u;
        p2v2[k] = lds_addr(Vs) + (unsigned)(((w_cp * 32 + w_tile) * 74 + 12 * sub6) * 4);
    }
    f32x2 td[TW][6], tr[TW][6];
#pragma unroll
    for (int k = 0; k < TW; ++k)
#pragma unroll
        for (int i = 0; i < 6; ++i) { td[k][i] = f32x2{(float)tid, 1.f}; tr[k][i] = f32x2{2.f, (float)i}; }
    const unsigned long long kc = 0x3f8000003f000000ull;
    typedef const __attribute__((address_space(3))) f32x2* lds_pair_ptr;
    const unsigned raw_lds = lds_addr(raw), us_lds = lds_addr(Us);
    // DMA shares: 36 weight blocks + 10 raw blocks of 1 KiB per unit
    constexpr int NU = (36 + NW - 1) / NW + (NW == 8 ? 0 : 0);      // 5 (8 waves) / 9 (4 waves)
    constexpr int NR = NW == 8 ? 2 : 3;
    const float* rptr = rsrc + ((long long)blockIdx.x * 7919 * 2560) % (rsrc_floats - 4 * 2560);

    auto ufrag_read = [&](const int slot, const float* ub, const int g) __attribute__((always_inline)) {
        if (ELIM & 4) return;
#pragma unroll
        for (int m = 0; m < NM; ++m) af[slot][m] = *reinterpret_cast<const f32x4*>(ub + (ug_lane - wsrc) + m * 2304 + g * 1024);
    };
    auto frag_read = [&](const int slot, const int vb, const int g) __attribute__((always_inline)) {
        if (ELIM & 4) return;
        if (!UG) {
#pragma unroll
        for (int m = 0; m < NM; ++m) af[slot][m] = *reinterpret_cast<const f32x4*>(&Us[vb * UB4 + a_base + m * 16 * 36 + 4 * g]);
        }
        if (DA) {
            const unsigned a = vp_lane + (unsigned)vb * (VPB * 4u);
            if (g < 6) bf[slot] = reinterpret_cast<lds_quad_ptr0>(a + (unsigned)(2 * g * 64) * 4u)[0];          // rows 0..3 of column g
            else {                                                                                           // rows 4, 5 of columns 2k, 2k+1
                const f32x2 x = reinterpret_cast<lds_pair_ptr0>(a + (unsigned)(2 * (2 * (g - 6)) * 64 + 4) * 4u)[0];
                const f32x2 y = reinterpret_cast<lds_pair_ptr0>(a + (unsigned)(2 * (2 * (g - 6) + 1) * 64 + 4) * 4u)[0];
                bf[slot] = f32x4{x[0], x[1], y[0], y[1]};
            }
            return;
        }
        if (VAR & 524288) {
            typedef const __attribute__((address_space(3))) float* lds_f_ptr;
            const unsigned a = lds_addr(Vs) + (unsigned)(vb * V2B + (((lane >> 5) * 32 + tb * 16 + (lane & 15)) * 74 + ((lane >> 4) & 1)) * 4);
            const lds_f_ptr pp = reinterpret_cast<lds_f_ptr>(a);
            bf[slot] = f32x4{pp[8 * g], pp[8 * g + 2], pp[8 * g + 4], pp[8 * g + 6]};
            return;
        }
        bf[slot] = *reinterpret_cast<const f32x4*>(&Vs[vb * VB4 + b_base + 4 * g]);
    };
    auto dma16m = [&](const float* sbase, unsigned voff_bytes, unsigned lds_byte, int wv, auto lim_tag) __attribute__((always_inline)) {
        constexpr int LIM = decltype(lim_tag)::value;
        unsigned keep; unsigned long long save;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_mov_b64 %1, exec\n\ts_cmp_lt_u32 %5, %6\n\ts_cselect_b64 exec, exec, 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(save) : "v"(voff_bytes), "s"(sbase), "s"(lds_byte), "s"(wv), "n"(LIM) : "memory", "scc");
    };
    auto u_dma = [&](const float* ub, const int buf, const int i) __attribute__((always_inline)) {
        if (ELIM & 2) return;
        if (VAR & 16384) {         // the wave's 9 blocks are contiguous: one base pair, instruction offsets, M0 written with the first DMA of a run only
            const float* sb = ub + wave * 2304 + 1024;
            const unsigned m0v = us_lds + (unsigned)(buf * UB4 + wave * 2304 + 1024) * 4u;
            const unsigned vo = (unsigned)lane * 16u;
#define UD(OFF, SETM0) do { if (SETM0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #OFF :: "v"(vo), "s"(sb), "s"(m0v) : "memory"); \
                            else asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #OFF :: "v"(vo), "s"(sb) : "memory"); } while (0)
            if (i == 0) UD(-4096, 1); else if (i == 1) UD(-3072, 0); else if (i == 2) UD(-2048, 0); else if (i == 3) UD(-1024, 0);
            else if (i == 4) UD(0, 0); else if (i == 5) UD(1024, 1); else if (i == 6) UD(2048, 0); else if (i == 7) UD(3072, 0);
            else asm volatile("s_add_u32 m0, %2, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(vo + 1024u), "s"(sb), "s"(m0v) : "memory", "scc");
#undef UD
            return;
        }
        if (VAR & 8192) {
            int bb = wave + NW * i; bb = bb < 36 ? bb : 35;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((unsigned)lane * 16u), "s"(ub + bb * 256), "s"(us_lds + (unsigned)(buf * UB4 + bb * 256) * 4u) : "memory");
            return;
        }
        int b = wave + NW * i;
        b = b < 36 ? b : 35;
        if ((VAR & 32) && NW == 8 && i == 4) { dma16m(ub + b * 256, (unsigned)lane * 16u, us_lds + (unsigned)(buf * UB4 + b * 256) * 4u, wave, std::integral_constant<int, 36 - NW * 4>{}); return; }
        dma16(ub + b * 256, (unsigned)lane * 16u, us_lds + (unsigned)(buf * UB4 + b * 256) * 4u);
    };
    auto raw_dma = [&](const float* src, const unsigned boff, const int j) __attribute__((always_inline)) {
        if (ELIM & 2) return;
        const int b = wave + NW * j;
        const unsigned dst = b < 10 ? raw_lds + (unsigned)b * 1024u + boff : raw_lds + 3u * RB4 * 4u;
        if ((VAR & 32) && NW == 4 && j == 2) { dma16m(src + (b < 10 ? b : 0) * 256, (unsigned)lane * 16u, dst, wave, std::integral_constant<int, 10 - 2 * NW>{}); return; }
        if ((VAR & 32) && NW == 8 && j == 1) { dma16m(src + (b < 10 ? b : 0) * 256, (unsigned)lane * 16u, dst, wave, std::integral_constant<int, 10 - NW>{}); return; }
        dma16(src + (b < 10 ? b : 0) * 256, (unsigned)lane * 16u, dst);
    };
    // transform pieces of task k
    auto tf_p1 = [&](const int k, const unsigned rbyte) __attribute__((always_inline)) {
        if (ELIM & (1 | 16)) return;
        const unsigned a = p1_base[k] + rbyte;
#pragma unroll
        for (int i = 0; i < 6; ++i) td[k][i] = reinterpret_cast<lds_pair_ptr>(a)[i * PX * 2];
    };
    auto tf_p1b = [&](const int k) __attribute__((always_inline)) { if (!(ELIM & (1 | 32))) bt6(td[k], tr[k], kc); };
    const unsigned addtid_m0 = DA ? 0u : (lds_addr(scr) + (unsigned)wave * 4096u) & 0xffffu;   // (legacy variants: any valid wave-private address)
    // design A: this wave's region of V'[buf]; the hand-over X blocks (i, c) at (2 i + c) * 64 + i alias the V' blocks (j, c) at (2 j + c) * 64
    const unsigned da_wave = lds_addr(Vs) + (unsigned)(wave * WR) * 4u;
    const unsigned da_rd = da_wave + (unsigned)((lane & 7) * 129 + (lane >> 3) * 8) * 4u;      // row r = lane & 7: block (r, 0) + r + wl * 8
    auto tf_p1w = [&](const int k, const int q, const int buf = 0) __attribute__((always_inline)) {
        if (ELIM & (1 | 64)) return;
        unsigned long long save;
        if (DA) {               // rows 2q, 2q+1, both channels: four lane-linear stores into the blocks (i, c) of this wave's V'[buf] region
            unsigned keep;
            const unsigned m0v = da_wave + (unsigned)buf * (VPB * 4u);
            if (q == 0) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %2 offset:256\n\t"
                                     "ds_write_addtid_b32 %3 offset:516\n\tds_write_addtid_b32 %4 offset:772\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(tr[k][0][0]), "v"(tr[k][0][1]), "v"(tr[k][1][0]), "v"(tr[k][1][1]), "s"(m0v) : "memory");
            else if (q == 1) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:1032\n\tds_write_addtid_b32 %2 offset:1288\n\t"
                                          "ds_write_addtid_b32 %3 offset:1548\n\tds_write_addtid_b32 %4 offset:1804\n\ts_mov_b32 m0, %0"
                                          : "=&s"(keep) : "v"(tr[k][2][0]), "v"(tr[k][2][1]), "v"(tr[k][3][0]), "v"(tr[k][3][1]), "s"(m0v) : "memory");
            else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:2064\n\tds_write_addtid_b32 %2 offset:2320\n\t"
                              "ds_write_addtid_b32 %3 offset:2580\n\tds_write_addtid_b32 %4 offset:2836\n\ts_mov_b32 m0, %0"
                              : "=&s"(keep) : "v"(tr[k][4][0]), "v"(tr[k][4][1]), "v"(tr[k][5][0]), "v"(tr[k][5][1]), "s"(m0v) : "memory");
            return;
        }
        if (VAR & 2) {          // four lane-linear 4-byte stores (address = M0 + offset + 4 * lane): no address register, no exec mask
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %2 offset:256\n\t"
                         "ds_write_addtid_b32 %3 offset:512\n\tds_write_addtid_b32 %4 offset:768\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(tr[k][2 * q][0]), "v"(tr[k][2 * q][1]), "v"(tr[k][2 * q + 1][0]), "v"(tr[k][2 * q + 1][1]),
                           "s"(addtid_m0 + (unsigned)q * 1024u) : "memory");
            return;
        }
        if (VAR & 64) {
            unsigned long long sv;
            asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\tds_write_b64 %1, %2 offset:%5\n\tds_write_b64 %1, %3 offset:%6\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "v"(scr_w[k]), "v"(tr[k][2 * q]), "v"(tr[k][2 * q + 1]), "s"(0x3F3F3F3F3F3F3F3Full), "n"(0), "n"(48) : "memory");
            return;
        }
        if (VAR & 8) {
            if (q == 0) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:6" :: "v"(scr_w[k]), "v"(tr[k][0]), "v"(tr[k][1]) : "memory");
            else if (q == 1) asm volatile("ds_write2_b64 %0, %1, %2 offset0:12 offset1:18" :: "v"(scr_w[k]), "v"(tr[k][2]), "v"(tr[k][3]) : "memory");
            else asm volatile("ds_write2_b64 %0, %1, %2 offset0:24 offset1:30" :: "v"(scr_w[k]), "v"(tr[k][4]), "v"(tr[k][5]) : "memory");
            return;
        }
        if (q == 0) asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\tds_write2_b64 %1, %2, %3 offset0:0 offset1:6\n\ts_mov_b64 exec, %0"
                                 : "=&s"(save) : "v"(scr_w[k]), "v"(tr[k][0]), "v"(tr[k][1]), "s"(0x3F3F3F3F3F3F3F3Full) : "memory");
        else if (q == 1) asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\tds_write2_b64 %1, %2, %3 offset0:12 offset1:18\n\ts_mov_b64 exec, %0"
                                      : "=&s"(save) : "v"(scr_w[k]), "v"(tr[k][2]), "v"(tr[k][3]), "s"(0x3F3F3F3F3F3F3F3Full) : "memory");
        else asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\tds_write2_b64 %1, %2, %3 offset0:24 offset1:30\n\ts_mov_b64 exec, %0"
                          : "=&s"(save) : "v"(scr_w[k]), "v"(tr[k][4]), "v"(tr[k][5]), "s"(0x3F3F3F3F3F3F3F3Full) : "memory");
    };
    typedef const __attribute__((address_space(3))) f32x4* lds_quad_ptr;
    auto tf_p2 = [&](const int k, const int buf = 0) __attribute__((always_inline)) {
        if (ELIM & (1 | 128)) return;
        if (DA) {               // row r of the window, both channels: six ds_read2_b32 (the channel blocks are 64 dwords apart)
            const unsigned a = da_rd + (unsigned)buf * (VPB * 4u);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                td[k][j] = f32x2{reinterpret_cast<const __attribute__((address_space(3))) float*>(a)[j], reinterpret_cast<const __attribute__((address_space(3))) float*>(a)[j + 64]};
            return;
        }
        if (VAR & 4) {          // the row as 2 x (16 + 8 bytes): what a lane-linear hand-over layout needs
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 a = reinterpret_cast<lds_quad_ptr>(scr_r[k] + c * 1024u)[0];
                const f32x2 b = reinterpret_cast<lds_pair_ptr>(scr_r[k] + c * 1024u)[2];
                td[k][0][c] = a[0]; td[k][1][c] = a[1]; td[k][2][c] = a[2]; td[k][3][c] = a[3]; td[k][4][c] = b[0]; td[k][5][c] = b[1];
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) td[k][j] = reinterpret_cast<lds_pair_ptr>(scr_r[k])[j];
    };
    auto tf_p2b = [&](const int k) __attribute__((always_inline)) { if (!(ELIM & (1 | 32))) bt6(td[k], tr[k], kc); };
    const unsigned addtid_v = lds_addr(Vs) + (unsigned)wave * 3072u;
    auto tf_p2w = [&](const int k, const int buf, const int q) __attribute__((always_inline)) {
        if (ELIM & (1 | 256)) return;
        if (DA) {               // columns 2q, 2q+1, both channels: blocks (j, c) at (2 j + c) * 64 dwords
            unsigned keep;
            const unsigned m0v = da_wave + (unsigned)buf * (VPB * 4u);
            if (q == 0) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %2 offset:256\n\t"
                                     "ds_write_addtid_b32 %3 offset:512\n\tds_write_addtid_b32 %4 offset:768\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(tr[k][0][0]), "v"(tr[k][0][1]), "v"(tr[k][1][0]), "v"(tr[k][1][1]), "s"(m0v) : "memory");
            else if (q == 1) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:1024\n\tds_write_addtid_b32 %2 offset:1280\n\t"
                                          "ds_write_addtid_b32 %3 offset:1536\n\tds_write_addtid_b32 %4 offset:1792\n\ts_mov_b32 m0, %0"
                                          : "=&s"(keep) : "v"(tr[k][2][0]), "v"(tr[k][2][1]), "v"(tr[k][3][0]), "v"(tr[k][3][1]), "s"(m0v) : "memory");
            else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:2048\n\tds_write_addtid_b32 %2 offset:2304\n\t"
                              "ds_write_addtid_b32 %3 offset:2560\n\tds_write_addtid_b32 %4 offset:2816\n\ts_mov_b32 m0, %0"
                              : "=&s"(keep) : "v"(tr[k][4][0]), "v"(tr[k][4][1]), "v"(tr[k][5][0]), "v"(tr[k][5][1]), "s"(m0v) : "memory");
            return;
        }
        if (VAR & 524288) {     // V2: the channel pair of a position is 8 contiguous bytes: two ds_write_b64 per slot
            unsigned long long sv;
            const unsigned a2 = p2v2[k] + (unsigned)buf * (unsigned)V2B + 16u * (unsigned)q;
            asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\tds_write_b64 %1, %2 offset:%5\n\tds_write_b64 %1, %3 offset:%6\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "v"(a2), "v"(tr[k][2 * q]), "v"(tr[k][2 * q + 1]), "s"(0x3F3F3F3F3F3F3F3Full), "n"(0), "n"(8) : "memory");
            return;
        }
        const unsigned a0 = p2_addr[k] + (unsigned)buf * (VB4 * 4u), a1 = a0 + 32u * 36u * 4u;
        const int j = 2 * q;
        unsigned long long save;
        if (VAR & 1) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %2 offset:256\n\t"
                         "ds_write_addtid_b32 %3 offset:512\n\tds_write_addtid_b32 %4 offset:768\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(tr[k][j][0]), "v"(tr[k][j][1]), "v"(tr[k][j + 1][0]), "v"(tr[k][j + 1][1]),
                           "s"(addtid_v + (unsigned)buf * (VB4 * 4u) + (unsigned)q * 1024u) : "memory");
            return;
        }
#define VST(O0, O1)                                                                                                             \
        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %7\n\tds_write2_b32 %1, %2, %3 offset0:" #O0 " offset1:" #O1    \
                     "\n\tds_write2_b32 %4, %5, %6 offset0:" #O0 " offset1:" #O1 "\n\ts_mov_b64 exec, %0"                         \
                     : "=&s"(save) : "v"(a0), "v"(tr[k][j][0]), "v"(tr[k][j + 1][0]), "v"(a1), "v"(tr[k][j][1]), "v"(tr[k][j + 1][1]), \
                       "s"((VAR & 8) ? ~0ull : 0x3F3F3F3F3F3F3F3Full) : "memory")
        if (q == 0) VST(0, 1); else if (q == 1) VST(2, 3); else VST(4, 5);
#undef VST
    };

    // prologue: fragments of groups 0, 1
    frag_read(0, 0, 0); frag_read(1, 0, 1);
    unsigned rd_byte = 0u, rt_byte = RB4 * 4u;
    // ---- HW (VAR & 1024, NW = 4): one lane = one WHOLE window (tile, channel pair) of ONE unit, rows {0,1,2} or {5,3,4} of the
    // transformed window: no hand-over.  Waves 0, 1 take the two halves of unit n+2, waves 2, 3 of unit n+3; the column pass runs in
    // the even unit (42 reads, 36 packed VALU, X = 18 pairs stay in registers), the row pass in the odd unit (36 VALU, 18 ds_write2_b32)
    constexpr bool HW = (VAR & 1024) != 0;
    // NW = 8 (VAR & 1024): the same half-window tasks on FOUR of the eight waves (waves 0-3, or -- VAR & 32768 -- the even waves);
    // waves ti = 0, 1 run the column pass in even units, ti = 2, 3 in odd units (staggered: every unit carries both passes)
    const int hw_ti = NW == 8 ? ((VAR & 32768) ? wave >> 1 : wave & 3) : wave;
    const int hw_tile = lane & 31, hw_cp = lane >> 5, hw_half = hw_ti & 1, hw_unit = hw_ti >> 1;
    const unsigned hw_rd = raw_lds + (unsigned)((((4 * (hw_tile / 8)) * PX + 4 * (hw_tile % 8)) * 4 + 2 * hw_cp) * 4) + (unsigned)hw_unit * (RB4 * 4u);
    const unsigned hw_rd_cf = raw_lds + (unsigned)lane * 8u + (unsigned)hw_unit * (RB4 * 4u);
    const unsigned hw_sh = hw_half ? (unsigned)PX * 16u : 0u;                     // half B reads e0, e2, e4 one row lower
    // rows of V this lane writes: half A 0, 1, 2; half B 5, 3, 4
    const unsigned hw_wr = lds_addr(Vs) + (unsigned)(((2 * hw_cp * 32 + hw_tile) * 36) * 4) + (unsigned)hw_unit * (VB4 * 4u);
    const unsigned hw_row0 = hw_wr + (hw_half ? 5u : 0u) * 24u, hw_row1 = hw_wr + (hw_half ? 3u : 1u) * 24u, hw_row2 = hw_wr + (hw_half ? 4u : 2u) * 24u;
    const unsigned long long hw_k1 = hw_half ? 0x3f0000003f800000ull : kc, hw_k2 = hw_k1 + 0x0010000000100000ull, hw_k3 = hw_k1 + 0x0020000000200000ull;
    f32x2 hx[3][6];           // X[row k of the half][column]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 6; ++c) hx[k][c] = f32x2{(float)tid, (float)c};
    f32x2 hd[2][7], ho[6];    // a column's 7 inputs (e0, e2, e4, m1..m4), double-buffered; a row's 6 outputs
#pragma unroll
    for (int i = 0; i < 7; ++i) { hd[0][i] = f32x2{1.f, 2.f}; hd[1][i] = f32x2{1.f, 2.f}; }
#pragma unroll
    for (int i = 0; i < 6; ++i) ho[i] = f32x2{(float)lane, 3.f};
    // real layouts (VAR & 65536: half per wave, lane = 2 tile + cp, P = 36; VAR & 131072: half per lane, P = 38)
    constexpr bool R1 = (VAR & 65536) != 0, R2 = (VAR & 131072) != 0;
    constexpr int RP = R2 ? 38 : 36;
    int r_tile, r_cp, r_half;
    if (R2) { const int l5 = lane & 31; r_cp = l5 & 1; r_half = (l5 >> 1) & 1; r_tile = 16 * (hw_ti & 1) + 4 * (2 * (lane >> 5) + (l5 >> 4)) + ((l5 >> 2) & 3); }
    else { r_tile = lane >> 1; r_cp = lane & 1; r_half = hw_ti & 1; }
    const int r_ty = (r_tile >> 2) & 3, r_tx = 4 * (r_tile >> 4) + (r_tile & 3);
    const unsigned r_rd = raw_lds + (unsigned)(((4 * r_ty) * RP + r_ty + 4 * r_tx) * 16 + 8 * r_cp) + (unsigned)hw_unit * (RB4 * 4u);
    const unsigned r_sh = r_half ? (unsigned)RP * 16u : 0u;
    const unsigned r_w0 = lds_addr(Vs) + (unsigned)(hw_unit * VB4 + (2 * r_cp * 32 + r_tile) * 36) * 4u;
    const unsigned r_wr0 = r_w0 + (r_half ? 5u : 0u) * 24u, r_wr1 = r_w0 + (r_half ? (R2 ? 4u : 3u) : 1u) * 24u, r_wr2 = r_w0 + (r_half ? (R2 ? 3u : 4u) : 2u) * 24u;
    auto hw_read = [&](const int c) __attribute__((always_inline)) {           // column c: rows e0/e2/e4 (shifted for half B) and m1..m4
        if (ELIM & (1 | 16)) return;
        if (R1 || R2) {
            const unsigned a = r_rd, b = r_rd + r_sh;
#pragma unroll
            for (int i = 0; i < 3; ++i) hd[0][i] = reinterpret_cast<lds_pair_ptr>(b)[2 * (2 * i * RP + ((2 * i) >> 2)) + 2 * c];
#pragma unroll
            for (int i = 1; i < 5; ++i) hd[0][2 + i] = reinterpret_cast<lds_pair_ptr>(a)[2 * (i * RP + (i >> 2)) + 2 * c];
            return;
        }
        const unsigned a = ((VAR & 2048) ? hw_rd_cf : hw_rd) + (unsigned)c * 16u, b = a + hw_sh;
#pragma unroll
        for (int i = 0; i < 3; ++i) hd[NW == 8 ? 0 : c & 1][i] = reinterpret_cast<lds_pair_ptr>(b)[2 * i * PX * 2];
#pragma unroll
        for (int i = 0; i < 4; ++i) hd[NW == 8 ? 0 : c & 1][3 + i] = reinterpret_cast<lds_pair_ptr>(a)[(1 + i) * PX * 2];
    };
    auto hw_col = [&](const int c) __attribute__((always_inline)) {            // 6 packed multiply-adds: three rows of X of column c
        if (ELIM & (1 | 32)) return;
        const f32x2* d = hd[NW == 8 ? 0 : c & 1];
        hx[0][c] = pk_fma_k<0>(d[0], hw_k2, pk_fnma_k<1>(d[1], hw_k2, d[2]));
        const f32x2 t0 = pk_fnma_k<1>(d[4], hw_k1, d[6]), t1 = pk_fnma_k<1>(d[3], hw_k1, d[5]);
        hx[1][c] = pk_fma_k<0>(t1, hw_k3, t0);
        hx[2][c] = pk_fnma_k<0>(t1, hw_k3, t0);
    };
    auto hw_row = [&](const int k) __attribute__((always_inline)) { if (!(ELIM & (1 | 32))) bt6(hx[k], ho, kc); };
    auto hw_st = [&](const int k, const int jj) __attribute__((always_inline)) {   // V[row][2 jj, 2 jj + 1] of both channels
        if (ELIM & (1 | 256)) return;
        const unsigned a0 = (VAR & 262144) ? (k == 0 ? r_wr0 : k == 1 ? r_wr1 : r_wr2) : (VAR & 2048) ? lds_addr(Vs) + (unsigned)lane * 8u + (unsigned)k * 1024u + (unsigned)hw_unit * (VB4 * 4u) : k == 0 ? hw_row0 : k == 1 ? hw_row1 : hw_row2;
        if (jj == 0) asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:1\n\tds_write2_b32 %3, %4, %5 offset0:0 offset1:1"
                                  :: "v"(a0), "v"(ho[0][0]), "v"(ho[1][0]), "v"(a0 + 4608u), "v"(ho[0][1]), "v"(ho[1][1]) : "memory");
        else if (jj == 1) asm volatile("ds_write2_b32 %0, %1, %2 offset0:2 offset1:3\n\tds_write2_b32 %3, %4, %5 offset0:2 offset1:3"
                                       :: "v"(a0), "v"(ho[2][0]), "v"(ho[3][0]), "v"(a0 + 4608u), "v"(ho[2][1]), "v"(ho[3][1]) : "memory");
        else asm volatile("ds_write2_b32 %0, %1, %2 offset0:4 offset1:5\n\tds_write2_b32 %3, %4, %5 offset0:4 offset1:5"
                          :: "v"(a0), "v"(ho[4][0]), "v"(ho[5][0]), "v"(a0 + 4608u), "v"(ho[4][1]), "v"(ho[5][1]) : "memory");
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto unit_body = [&](const int c, auto vb_tag, auto duty_tag) __attribute__((always_inline)) {
        constexpr int vb = decltype(vb_tag)::value;
        constexpr int duty = decltype(duty_tag)::value;      // 2: every wave one task (the shipped kernel), 1: two tasks, 0: none
        const float* const un2 = wsrc + (long long)((c + 2) & 15) * UB4;
        const unsigned third = 3u * RB4 * 4u - rd_byte - rt_byte;
#pragma unroll
        for (int g = 0; g < 9; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int s = 4 * g + e;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    if (NW == 8) acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g % 3][m][e], bf[g % 3][e], acc[s][m], 0, 0, 0);
                    else if (s * NM + m < 64) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[s][m]) : "v"(af[g % 3][m][e]), "v"(bf[g % 3][e]));
                    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[s][m]) : "v"(af[g % 3][m][e]), "v"(bf[g % 3][e]));
                    __builtin_amdgcn_sched_barrier(0);
                    // slot (g, e, m): what rides behind this MFMA
                    if (NW == 8) {
                        if (e == 0) frag_read((g + 2) % 3, g + 2 < 9 ? vb : vb ^ 1, g + 2 < 9 ? g + 2 : g - 7);
                        if (UG && e == 3) {      // global fragment two groups ahead
                            const int ga = g + 2;
                            ufrag_read(ga % 3, wsrc + (long long)((c + (ga >= 9)) & 15) * UB4, ga % 9);
                        }
                        if (e == 1) {
                            if (g == 7 && !UG) { u_dma(un2, vb, 0); u_dma(un2, vb, 1); u_dma(un2, vb, 2); }
                            else if (g == 8 && !UG) { u_dma(un2, vb, 3); u_dma(un2, vb, 4); }
                            else if (g == 0) raw_dma(rptr, rd_byte, 0);
                            else if (g == 1) raw_dma(rptr, rd_byte, 1);
                        }
                        if (duty == 2) {
                        if (g == 0 && e == 2) tf_p1(0, rt_byte);
                        else if (g == 2 && e == 2) { tf_p1b(0); tf_p1w(0, 0, vb ^ 1); }
                        else if (g == 2 && e == 3) tf_p1w(0, 1, vb ^ 1);
                        else if (g == 3 && e == 1) tf_p1w(0, 2, vb ^ 1);
                        else if (g == 3 && e == 2) tf_p2(0, vb ^ 1);
                        else if (g == 5 && e == 2) { tf_p2b(0); tf_p2w(0, vb ^ 1, 0); }
                        else if (g == 5 && e == 3) tf_p2w(0, vb ^ 1, 1);
                        else if (g == 6 && e == 1) tf_p2w(0, vb ^ 1, 2);
                        } else if (duty == 3 || duty == 4) {
                            if ((vb == 0) == (duty == 3)) {      // column pass: column g read in group g, evaluated in group g + 1
                                if (g >= 1 && g < 7 && e == 1) hw_col(g - 1);
                                if (g < 6 && e == 2) hw_read(g);
                            } else {                             // row pass: row k evaluated in group 2 k, three store pairs behind it
                                if ((g & 1) == 0 && g < 6 && e == 2) hw_row(g >> 1);
                                if ((g & 1) == 0 && g < 6 && e == 3) hw_st(g >> 1, 0);
                                if ((g & 1) == 1 && g < 6 && e == 1) hw_st(g >> 1, 1);
                                if ((g & 1) == 1 && g < 6 && e == 2) hw_st(g >> 1, 2);
                            }
                        } else if (duty == 1) {
                        if (g == 0 && e == 2) tf_p1(0, rt_byte);
                        else if (g == 0 && e == 3) tf_p1(TW - 1, rt_byte);
                        else if (g == 1 && e == 2) { tf_p1b(0); tf_p1w(0, 0); }
                        else if (g == 1 && e == 3) tf_p1w(0, 1);
                        else if (g == 2 && e == 1) tf_p1w(0, 2);
                        else if (g == 2 && e == 2) { tf_p1b(TW - 1); tf_p1w(TW - 1, 0); }
                        else if (g == 2 && e == 3) tf_p1w(TW - 1, 1);
                        else if (g == 3 && e == 1) tf_p1w(TW - 1, 2);
                        else if (g == 3 && e == 2) tf_p2(0);
                        else if (g == 3 && e == 3) tf_p2(TW - 1);
                        else if (g == 4 && e == 2) { tf_p2b(0); tf_p2w(0, vb ^ 1, 0); }
                        else if (g == 4 && e == 3) tf_p2w(0, vb ^ 1, 1);
                        else if (g == 5 && e == 1) tf_p2w(0, vb ^ 1, 2);
                        else if (g == 5 && e == 2) { tf_p2b(TW - 1); tf_p2w(TW - 1, vb ^ 1, 0); }
                        else if (g == 5 && e == 3) tf_p2w(TW - 1, vb ^ 1, 1);
                        else if (g == 6 && e == 1) tf_p2w(TW - 1, vb ^ 1, 2);
                        }
                        if (g == 6 && e == 3) {
                            if (!(ELIM & 8)) { if (UG) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); __syncthreads(); }
                        }
                    } else {
                        // one wave per SIMD: 8 slots per group; the same work spread over twice the slots
                        const int q = 2 * e + m;           // slot in the group, 0..7
                        if (q == 0) frag_read((g + 2) % 3, g + 2 < 9 ? vb : vb ^ 1, g + 2 < 9 ? g + 2 : g - 7);
                        if (VAR & 4096) {       // U(n+2) spread from group 7 of this unit to group 1 of the next (un1: the unit before's target)
                            const float* const un1 = wsrc + (long long)((c + 1) & 15) * UB4;
                            if (g == 7) { if (q == 1) u_dma(un2, vb, 0); if (q == 4) u_dma(un2, vb, 1); if (q == 7) u_dma(un2, vb, 2); }
                            else if (g == 8) { if (q == 2) u_dma(un2, vb, 3); if (q == 5) u_dma(un2, vb, 4); }
                            else if (g == 0) { if (q == 1) u_dma(un1, vb ^ 1, 5); if (q == 4) u_dma(un1, vb ^ 1, 6); if (q == 7) u_dma(un1, vb ^ 1, 7); }
                            else if (g == 1) { if (q == 2) u_dma(un1, vb ^ 1, 8); }
                            else if (g == 2) { if (q == 1) raw_dma(rptr, rd_byte, 0); if (q == 5) raw_dma(rptr, rd_byte, 1); }
                            else if (g == 3) { if (q == 1) raw_dma(rptr, rd_byte, 2); }
                        } else {
                        if (g == 7) { if (q >= 1 && q <= 5) u_dma(un2, vb, q - 1); }
                        else if (g == 8) { if (q >= 1 && q <= 4) u_dma(un2, vb, q + 4); }
                        else if (g == 0) { if (q == 1) raw_dma(rptr, rd_byte, 0); if (q == 3) raw_dma(rptr, rd_byte, 1); }
                        else if (g == 1) { if (q == 1) raw_dma(rptr, rd_byte, 2); }
                        }
                        if (HW) {
                            if (vb == 0) {              // column pass: column c read in group c, evaluated in group c + 1
                                if (g < 6 && q == 2) hw_read(g);
                                if (g >= 1 && g < 7 && q == 5) hw_col(g - 1);
                            } else {                    // row pass: row k evaluated in group 2 k, stored one ds_write2 pair per slot
                                if ((g & 1) == 0 && g < 6 && q == 3) hw_row(g >> 1);
                                if ((g & 1) == 0 && g < 6 && q >= 4 && q <= 6) hw_st(g >> 1, q - 4);
                            }
                        } else {
                        if (g == 0 && q == 4) tf_p1(0, rt_byte);
                        else if (g == 0 && q == 6) tf_p1(1, rt_byte);
                        else if (g == 2 && q == 2) { tf_p1b(0); tf_p1w(0, 0); }
                        else if (g == 2 && q == 3) tf_p1w(0, 1);
                        else if (g == 2 && q == 4) tf_p1w(0, 2);
                        else if (g == 2 && q == 6) { tf_p1b(1); tf_p1w(1, 0); }
                        else if (g == 2 && q == 7) tf_p1w(1, 1);
                        else if (g == 3 && q == 1) tf_p1w(1, 2);
                        else if (g == 3 && q == 3) tf_p2(0);
                        else if (g == 3 && q == 5) tf_p2(1);
                        else if (g == 5 && q == 2) { tf_p2b(0); tf_p2w(0, vb ^ 1, 0); }
                        else if (g == 5 && q == 3) tf_p2w(0, vb ^ 1, 1);
                        else if (g == 5 && q == 4) tf_p2w(0, vb ^ 1, 2);
                        else if (g == 5 && q == 6) { tf_p2b(1); tf_p2w(1, vb ^ 1, 0); }
                        else if (g == 5 && q == 7) tf_p2w(1, vb ^ 1, 1);
                        else if (g == 6 && q == 1) tf_p2w(1, vb ^ 1, 2);
                        }
                        if (g == 6 && q == 7) {
                            if (!(ELIM & 8)) { asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); __syncthreads(); }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        rptr += 2560;
        rd_byte = rt_byte; rt_byte = third;
    };
    using D0 = std::integral_constant<int, 0>; using D1 = std::integral_constant<int, 1>; using D2 = std::integral_constant<int, 2>;
    auto loop = [&](auto da, auto db) __attribute__((always_inline)) {
        for (int c = 0; c < units; c += 2) {
            unit_body(c, std::integral_constant<int, 0>{}, da);
            unit_body(c + 1, std::integral_constant<int, 1>{}, db);
            if ((c & 62) == 62) rptr = rsrc + ((long long)(blockIdx.x * 7919 + c) * 2560) % (rsrc_floats - 80 * 2560);
        }
    };
    using D3 = std::integral_constant<int, 3>; using D4 = std::integral_constant<int, 4>;
    if (HW && NW == 8) {    // three instantiations: column pass in even units / in odd units / no transform
        const bool tw = (VAR & 32768) ? (wave & 1) == 0 : wave < 4;
        if (!tw) loop(D0{}, D0{});
        else if (hw_unit == 0) loop(D3{}, D3{});
        else loop(D4{}, D4{});
    } else if (DUTY) {      // the choice is made ONCE per wave: two instantiations of the whole loop
        if (wave < 4) loop(D1{}, D0{}); else loop(D0{}, D1{});
    } else {
        loop(D2{}, D2{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < 36; ++s)
#pragma unroll
        for (int m = 0; m < NM; ++m) sum += acc[s][m][0] + acc[s][m][1] + acc[s][m][2] + acc[s][m][3];
    if (!HW) {
#pragma unroll
        for (int k = 0; k < TW; ++k) sum += tr[k][0][0] + tr[k][5][1];
    }
    if (HW && NW == 8) sum += hd[0][0][0] + hd[0][6][1];
    if (HW) sum += hx[0][0][0] + hx[2][5][1] + ho[0][0] + ho[5][1];
    if (sum == 123.456f) out[tid] = sum;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NW, int ELIM, int VAR = 0>
void run(const char* name, const float* w, const float* r, long long rf, float* out, unsigned long long* cyc)
{
    const int units = 4096;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    unit_probe<NW, ELIM, VAR><<<256, NW * 64>>>(w, r, rf, out, 64, cyc);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        unit_probe<NW, ELIM, VAR><<<256, NW * 64>>>(w, r, rf, out, units, cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double ns_unit = best * 1e6 / units;
    // issued MFMA FLOPs per unit and CU: 36 positions x 32 tiles x 64 couts x 4 channels x 2
    const double tf = 256.0 * 36 * 32 * 64 * 4 * 2 / ns_unit * 1e-3;
    printf("NW=%d %-28s %7.1f ns/unit  %7.0f memtime ticks/unit  %6.1f TF/s issued = %.3f of 157.3  (%s)\n", NW, name, ns_unit,
           (double)c / units, tf, tf / 157.3, hipGetErrorString(err));
}

int main()
{
    float *w, *r, *out; unsigned long long* cyc;
    const long long rf = 128ll << 20;          // 512 MiB of patch source
    (void)hipMalloc(&w, 16 * UB4 * 4); (void)hipMemset(w, 0, 16 * UB4 * 4);
    (void)hipMalloc(&r, rf * 4); (void)hipMemset(r, 0, rf * 4);
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 64);
    run<8, 0>("everything", w, r, rf, out, cyc);
    run<8, 1>("no input transform", w, r, rf, out, cyc);
    run<8, 256>("no V stores", w, r, rf, out, cyc);
    run<8, 0, 524288>("V as [cp][tile][pos][2]: b64 stores", w, r, rf, out, cyc);        // (two ds_read2_b32 per fragment: slower, 3425)
    run<8, 0, 1024 + 2048>("HW8 stand-in addresses", w, r, rf, out, cyc);
    run<8, 0, 1024 + 2048 + 65536 + 262144>("HW8 real reads+stores M1", w, r, rf, out, cyc);
    run<8, 0, 1024 + 2048 + 131072 + 262144>("HW8 real reads+stores M2", w, r, rf, out, cyc);
    run<4, 0, 1024 + 2048 + 4096>("NW4 HW cf, U DMAs spread", w, r, rf, out, cyc);
    return 0;
}

// 3x3 convolution by Winograd F(4x4, 3x3) on the fp32 matrix instruction, second generation (round 4): ONE wave per SIMD.
//   M[pos] (tiles x cout) += V[pos] (tiles x cin) * U[pos] (cin x cout),  pos = 0..35
// for the ReflectionPad -> Conv2d 3x3 -> ReLU -> BN [-> MaxPool] block (multipoint/models/MultiPoint.py:143-148, 62-65, 78-81).
// Same arithmetic as the first generation (conv_wino43.hip: U = G g G^T on the host in double, V = B^T d B and Y = A^T M A as
// fixed-order fp32 multiply-add chains, interpolation points {0, +-3/4, +-3/2, inf}); what changed is who does what, because the
// round-4 unit probe (tools/probes/unit_probe.hip) showed where the first generation's matrix pipe idled: its input transform
// (8 lanes per 6x6 window, a column pass and a row pass with a hand-over through LDS) cost ~860 of a unit's ~3300 cycles --
// 574 of them the LDS STORES (a store moves address + data VGPRs to the LDS at 2 cycles per dword and instruction: 51 dwords per
// lane and unit, 120 store instructions per CU and unit), 350 the 192 packed VALU instructions, 260 the reads.
//
//  * Workgroup = 256 threads = 4 waves, one per SIMD, 512 registers each.  Item = 32 tiles (4 x 8 or 8 x 4 tiles of 4x4 output
//    pixels) x 64 output channels; wave w multiplies tile block w & 1 (16 tiles) by channel block w >> 1 (32 couts = two MFMA
//    column blocks) for all 36 positions: 72 MFMAs per unit of 4 input channels on 288 accumulator registers (256 AGPRs + 32
//    VGPRs; the MFMAs are asm statements because hipcc otherwise forces all of them into AGPRs and shuttles the overflow).  One V
//    fragment feeds two MFMAs: 27 fragment reads per 72 MFMAs instead of 36.
//  * Input transform with NO hand-over: a lane owns a WHOLE window (tile, channel pair) of a unit and half of its rows -- the two
//    waves of a pair take rows {0, 1, 2} and {5, 3, 4} of the transformed window (the same instruction sequence with the roles of
//    the two interpolation points swapped and three reads one patch row lower).  Column pass in one unit (42 conflict-free
//    ds_read_b64, 36 packed VALU; X = 18 register pairs), row pass in the next (36 packed VALU, 36 ds_write_addtid_b32).  Per CU and
//    unit: 144 packed VALU (was 192), 36 store instructions moving 36 dwords per lane-slot (was 120 moving 408), no scratch.
//    The two pairs run one unit out of phase (pair 0: column pass in even units, pair 1 in odd units), so every unit carries one
//    column pass and one row pass, V stays double-buffered and the raw ring stays at three buffers.
//  * LDS: V' [buf][channel of the pair][pos][lane = 2 * tile + cp] (ds_write_addtid_b32 is lane-linear and needs no address
//    register; the MFMA side reads two positions per ds_read2st64_b32), at LDS offset 0 because the instruction takes its base
//    from M0[15:0]; U [buf][ch][cout][pos] as before; raw patches row by row with a row skew -- slot(y, x) = y * P + (y >> 2) + x --
//    and the item's tiles numbered so that 16 consecutive tiles are 4 tile rows x 4 tile columns: the 32 windows of a read group then
//    hit 32 distinct bank pairs while the DMA lanes still read runs of consecutive pixels.
//  * DMA order as in the first generation (weights and patches apart in time, one counted wait + barrier per unit), but the 9
//    weight DMAs of a wave are spread from behind the barrier of unit n to group 1 of unit n+1 (one per 3 MFMAs).
#include "mp_common.h"

#include <algorithm>
#include <type_traits>

#ifndef MPB_DMA_EARLY
#define MPB_DMA_EARLY 0
#endif
#ifndef MPQX
#define MPQX 0     // developer elimination switches (timing only, results WRONG): 1 no input transform, 2 no DMA, 8 no epilogue
#endif
namespace {

constexpr int UC4 = 4;                             // input channels per unit
constexpr int UB4 = UC4 * 64 * 36;                 // floats per U buffer  [ch][cout][pos]   (36 KiB)
constexpr int VR4 = 36 * 64 + 1;                   // floats per channel region of a V' buffer: [pos][lane], odd so that the two channels of a pair sit one bank apart
constexpr int VB4 = 2 * VR4 + 2;                   // floats per V' buffer (16-byte multiple)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int reflect_clamp_q(int v, int n)
{
    v = v < 0 ? -v : v;
    v = v >= n ? 2 * (n - 1) - v : v;
    v = v < 0 ? 0 : v;
    return v >= n ? n - 1 : v;
}
__device__ __forceinline__ float relu_q(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
// LDS-DMA (hazards in front of the statement are checked at build time: multipoint_amd/build.py)
__device__ __forceinline__ void dma16(const float* sbase, unsigned voff_bytes, unsigned lds_byte)
{
    if (MPQX & 2) return;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_byte) : "memory");
}
// ... with the instruction's immediate offset OFF: it moves the source AND the LDS destination (round-4 probe; docs/HISTORY.md 3.3), which
// is exactly what a block-for-block copy wants -- one scalar source base, one scalar destination base and one lane-offset register
// serve a wave's nine weight blocks (the unit bodies are short of scalar registers, and a lane-offset register per block would
// be spilled: the epilogue needs every register, and a reload inside a unit body waits with vmcnt(0) for ALL DMAs in flight).
// M0ADD is added to the destination only.
template <int OFF, int M0ADD>
__device__ __forceinline__ void dma16i(const float* sbase, unsigned voff_bytes, unsigned lds_base)
{
    if (MPQX & 2) return;
    unsigned keep;
    if (M0ADD == 0)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_base), "n"(OFF) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %5\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_base), "n"(OFF), "n"(M0ADD) : "memory", "scc");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }

// packed fp32 multiply-add with a coefficient from a scalar register pair (low / high half picked by op_sel)
constexpr unsigned long long pk_const2(double lo, double hi)
{
    return (unsigned long long)__builtin_bit_cast(unsigned, (float)lo) | ((unsigned long long)__builtin_bit_cast(unsigned, (float)hi) << 32);
}
constexpr double W43A = MP_W43_A, W43B = MP_W43_B;                  // interpolation points {0, +-a, +-b, inf} (mp_common.h)
constexpr unsigned long long K_AB = pk_const2(W43A, W43B), K_BA = pk_const2(W43B, W43A), K_A2B2 = pk_const2(W43A * W43A, W43B * W43B),
                             K_B2A2 = pk_const2(W43B * W43B, W43A * W43A),
                             K_PS = pk_const2(W43A * W43A * W43B * W43B, W43A * W43A + W43B * W43B);
static_assert((double)(float)(W43A * W43A * W43B * W43B) == W43A * W43A * W43B * W43B && (double)(float)(W43A * W43A + W43B * W43B) ==
              W43A * W43A + W43B * W43B, "the transform coefficients must be exact in fp32");
template <int HI>
__device__ __forceinline__ f32x2 pk_fma_k(f32x2 a, unsigned long long k, f32x2 c)      // a * k + c
{
    f32x2 d;
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}
template <int HI>
__device__ __forceinline__ f32x2 pk_fnma_k(f32x2 a, unsigned long long k, f32x2 c)     // c - a * k
{
    f32x2 d;
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}
// 1-D input transform B^T d (6 -> 6), packed over two channels: 12 multiply-adds (conv_wino43.hip's bt6, same order)
__device__ __forceinline__ void bt6(const f32x2 d[6], f32x2 r[6])
{
    const f32x2 t0 = pk_fnma_k<1>(d[2], K_A2B2, d[4]);      // d4 - b^2 d2
    const f32x2 t1 = pk_fnma_k<1>(d[1], K_A2B2, d[3]);      // d3 - b^2 d1
    const f32x2 t2 = pk_fnma_k<0>(d[2], K_A2B2, d[4]);      // d4 - a^2 d2
    const f32x2 t3 = pk_fnma_k<0>(d[1], K_A2B2, d[3]);      // d3 - a^2 d1
    r[0] = pk_fma_k<0>(d[0], K_PS, pk_fnma_k<1>(d[2], K_PS, d[4]));
    r[1] = pk_fma_k<0>(t1, K_AB, t0);
    r[2] = pk_fnma_k<0>(t1, K_AB, t0);
    r[3] = pk_fma_k<1>(t3, K_AB, t2);
    r[4] = pk_fnma_k<1>(t3, K_AB, t2);
    r[5] = pk_fma_k<0>(d[1], K_PS, pk_fnma_k<1>(d[3], K_PS, d[5]));
}
// (1-D output transform: at6s(), mp_common.h)

// accumulator s * 2 + m: the first 64 are the compiler's (MFMA builtin: hipcc keeps them in the 256 AGPRs), the last 8 are pinned to
// VGPRs by asm statements (288 > 256: with builtins only, hipcc still wants every MFMA result in an AGPR and shuttles the overflow
// through copies, 240 vector instructions per two units)
#define MPB_MFMA(ACC, A, B, IDX, FIRSTV)                                                                                   \
    do {                                                                                                                   \
        if ((IDX) < 64) ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, (FIRSTV) ? f32x4{0.f, 0.f, 0.f, 0.f} : ACC, 0, 0, 0);  \
        else if (FIRSTV) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(ACC) : "v"(A), "v"(B));                \
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B));                           \
    } while (0)

// an accumulator register for ordinary vector code: read from its AGPR WHERE it is used (left to itself hipcc copies every AGPR
// accumulator of a column block into VGPRs at the top of the epilogue -- 144 registers -- and spills the loop's state)
__device__ __forceinline__ float acc_read(float a)
{
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a));
    return x;
}
#define MPB_ACC_RD(DST, ACC, IDX, R)                                                                                       \
    do {                                                                                                                   \
        if ((IDX) < 64) DST = acc_read((ACC)[R]);                                                                          \
        else DST = (ACC)[R];                                                                                               \
    } while (0)

// SPLIT (single-pair latency, conv_wino43.hip's protocol): the launcher cut the input channels into 2^ks_shift ranges -- an item is
// (tile block, virtual slice = slice * ranges + range) and runs the units of its range only; its pre-bias output tiles go to
// p.split_scratch, where split_reduce_kernel (conv_split.hip, the next launch) sums the shares in range order and runs the epilogue.
template <bool POOL, bool BNF, int TC4, bool ZPAD, bool SPLIT = false>
__global__ __launch_bounds__(256, 1) void conv_wino43b_kernel(const ConvParams p)
{
    constexpr int TR4 = 32 / TC4;                      // tile rows x tile columns of an item
    constexpr int OY = 4 * TR4, OX = 4 * TC4;          // output pixels of an item
    constexpr int PY = OY + 2, PX = OX + 2;            // raw patch
    // raw patch in LDS: granule slot(y, x) = y * P + (y >> 2) + x -- rows stay contiguous (a DMA instruction's 64 lanes read runs of
    // consecutive pixels: 9-10 cache lines from a planar tensor), and the row skew y >> 2 makes the column pass conflict-free: a
    // 32-lane read group is 16 windows (4 tile rows x 4 tile columns, both channel pairs), whose pixels (4 ty + k, 4 tx + c) sit at
    // slots ty + 4 tx + const modulo 16 (P is a multiple of 4, 4 P a multiple of 16).  The first version swizzled the COLUMNS
    // ((x & 3) * 9 + (x >> 2)): equally conflict-free, but its DMA lanes read 16-byte pieces 64 bytes apart and the issue of such an
    // instruction stalls the wave (one wave per SIMD: nobody covers): conv2 3.57 vs 3.21 ms with linear reads (timing build).
    constexpr int P = TC4 == 8 ? 36 : 20;
    static_assert(P % 4 == 0 && P >= PX && (4 * P) % 16 == 0, "raw layout");
    constexpr int NG = (PY - 1) * P + ((PY - 1) >> 2) + PX;      // granules of a patch: 653 / 686
    constexpr int NRB = (NG + 63) / 64;                // DMA blocks of 64 granules: 11 / 12
    constexpr int RB4 = NRB * 256;                     // floats per raw buffer
    static_assert(NRB <= 12, "three raw DMA blocks per wave");

    // LDS: V' first (ds_write_addtid_b32 takes its base from M0[15:0]: the V' buffers must lie below 64 KiB)
    __shared__ __attribute__((aligned(16))) float smem[2 * VB4 + 2 * UB4 + 3 * RB4 + 256 + 3 * 64 + 4 * 3 * 64];
    float* const Vs = smem;
    float* const Us = smem + 2 * VB4;
    float* const raw = Us + 2 * UB4;                   // three buffers + one dummy block
    float* const prm = raw + 3 * RB4 + 256;
    // per-lane byte offsets of the patch gather (3 DMA blocks per wave): kept in LDS, not in registers -- they are loop-carried
    // state, and whatever lives in a register across the epilogue ends up spilled (a reload inside a unit body = vmcnt(0))

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = wave & 1, cbb = wave >> 1;          // tile block (16 tiles), channel block (32 couts) of this wave's GEMMs
    unsigned* const rvl = reinterpret_cast<unsigned*>(prm + 3 * 64) + wave * 192 + lane;
    const int half = wave & 1;                         // transform: rows {0, 1, 2} (0) or {5, 3, 4} (1) of the window; pair = wave >> 1
    const int NC = SPLIT ? (p.cin / UC4) >> p.ks_shift : p.cin / UC4;      // units per item (even)


    // ---- work items: (tile block, slice) of this XCD's contiguous eighth ----
    const XcdRange xr = xcd_range(p.nitems, p.xcd_shift);
    const int stride = xr.stride, item_end = xr.item_end;
    int item = xr.item;
    if (item >= item_end) return;

    auto udiv = [](unsigned n, unsigned magic, unsigned d) -> unsigned { return d == 1 ? n : __umulhi(n, magic); };
    struct Where { int slice, img, y0, x0; const float* in_base; };
    auto decode = [&](int it) __attribute__((always_inline)) -> Where {
        Where w{};
        const int tile = (int)udiv((unsigned)it, p.magic_slices, (unsigned)p.nslices);
        w.slice = it - tile * p.nslices;
        const int trow = (int)udiv((unsigned)tile, p.magic_tx, (unsigned)p.tiles_x);
        const int tx = tile - trow * p.tiles_x;
        const int bi = (int)udiv((unsigned)trow, p.magic_ty, (unsigned)p.tiles_y);
        const int ty = trow - bi * p.tiles_y;
        w.img = p.img_list ? p.img_list[bi] : bi;
        w.y0 = ty * OY; w.x0 = tx * OX;
        w.in_base = p.in_planar ? p.in + (long long)w.img * (p.cin / 4) * p.H * p.W * 4
                                : p.in + (long long)w.img * p.H * p.W * p.in_cstride + p.in_coff;
        if constexpr (SPLIT)      // first unit of the item's range of input channels
            w.in_base += (long long)((w.slice & ((1 << p.ks_shift) - 1)) * NC) * (p.in_planar ? (long long)p.H * p.W * 4 : UC4);
        return w;
    };

    // ---- raw patch staging by DMA: granule slot s = block * 64 + lane; this wave issues blocks wave, wave + 4, wave + 8 ----
    const int pix_stride = p.in_planar ? 4 : p.in_cstride;              // floats between horizontally adjacent pixels
    const long long unit_stride = p.in_planar ? (long long)p.H * p.W * 4 : UC4;     // floats between consecutive units
    bool roff_rel = false;        // rvl holds the item-invariant offsets of interior items
    auto slot_pixel = [&](const int j, int& py, int& px, const int ln) __attribute__((always_inline)) -> bool {
        const int s = (wave + 4 * j) * 64 + ln;
        py = s / P;
        if (py * P + (py >> 2) > s) --py;                 // (the skew moves a row's start by at most 4 slots)
        px = s - (py * P + (py >> 2));
        return py < PY && px < PX;
    };
    auto raw_offsets = [&](const Where& w) __attribute__((always_inline)) -> const float* {
        const bool interior = (w.y0 >= 1) && (w.y0 + OY < p.H) && (w.x0 >= 1) && (w.x0 + OX < p.W);
        int ln = lane;
        asm volatile("" : "+v"(ln));            // (opaque: the slot -> pixel map is recomputed per item instead of living -- spilled -- across the loop)
        if (interior) {
            if (!roff_rel) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    int py, px;
                    rvl[j * 64] = slot_pixel(j, py, px, ln) ? (unsigned)((py * p.W + px) * pix_stride) * 4u : 0u;
                }
                roff_rel = true;
            }
            return w.in_base + (long long)((w.y0 - 1) * p.W + (w.x0 - 1)) * pix_stride;
        }
        roff_rel = false;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int py, px;
            unsigned off = 0;
            if (slot_pixel(j, py, px, ln)) {
                if constexpr (ZPAD) {
                    // ZeroPad2d(1): pixels outside the frame are zeros -- their lanes sit out of the DMA and store a zero granule instead
                    const int gy = w.y0 + py - 1, gx = w.x0 + px - 1;
                    off = (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? (unsigned)((gy * p.W + gx) * pix_stride) * 4u : 0xFFFFFFFFu;
                } else {
                    const int gy = reflect_clamp_q(w.y0 + py - 1, p.H), gx = reflect_clamp_q(w.x0 + px - 1, p.W);
                    off = (unsigned)((gy * p.W + gx) * pix_stride) * 4u;
                }
            }
            rvl[j * 64] = off;
        }
        return w.in_base;
    };
    const unsigned raw_lds = lds_addr(raw), us_lds = lds_addr(Us), vs_lds = lds_addr(Vs);
    const unsigned raw_m0 = raw_lds + (unsigned)wave * 1024u;             // block `wave` of buffer 0
    const unsigned raw_dummy = raw_lds + 3u * RB4 * 4u;
    const f32x4 zero4v = {0.f, 0.f, 0.f, 0.f};
    auto raw_dma = [&](const float* src, unsigned boff_bytes, int j, unsigned voff) __attribute__((always_inline)) {
        const unsigned dst = (wave + 4 * j < NRB) ? raw_m0 + (unsigned)j * 4096u + boff_bytes : raw_dummy;
        if (MPQX & 16) return;                                            // (timing only)
        if (MPQX & 64) { dma16(src, (unsigned)lane * 16u, dst); return; } // (timing only: a linear 1 KiB instead of the patch gather)
        if (MPQX & 128) { dma16(p.in, voff, dst); return; }               // (timing only: the gather pattern on cache-hot addresses)
        if constexpr (ZPAD) {
            // lanes whose pixel lies outside the frame (offset ~0) are masked out of the DMA and write a zero granule into their slot
            // instead: every slot of the buffer is written each unit, by one or the other -- no branch in the unit body
            unsigned keep; unsigned long long save;
            const unsigned slot = dst + (unsigned)lane * 16u;
            asm volatile("v_cmp_ne_u32 vcc, -1, %2\n\ts_mov_b64 %1, exec\n\ts_and_b64 exec, exec, vcc\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 1\n\t"
                         "global_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0\n\ts_andn2_b64 exec, %1, vcc\n\tds_write_b128 %5, %6\n\ts_mov_b64 exec, %1"
                         : "=&s"(keep), "=&s"(save) : "v"(voff), "s"(src), "s"(dst), "v"(slot), "v"(zero4v) : "memory", "vcc", "scc");
            return;
        }
        dma16(src, voff, dst);
    };
    // weight blocks 9 wave + i (i = 0..8) of the unit whose weights start at `ub` -> U[buf]: source and destination bases point 4 KiB
    // into the wave's 9 KiB, blocks 0..7 are the immediates -4096 .. 3072, block 8 the immediate 3072 on bases 1 KiB further
    // lane constants of the unit loop: recomputed per item from an OPAQUE copy of the lane (lane_values(), below), so that their live
    // ranges end with the item's last unit -- values that live across the epilogue get spilled, and a reload inside a unit body is
    // an s_waitcnt vmcnt(0) on every DMA in flight
    unsigned lane16 = 0, tf_rd = 0, tf_eo = 0;
    int a_base = 0, b_base = 0;
    const unsigned us_w = us_lds + (unsigned)wave * 9216u + 4096u;
    auto u_dma = [&](const float* ubw, auto buf_tag, auto i_tag) __attribute__((always_inline)) {        // ubw = ub + wave * 2304 + 1024 floats
        constexpr int buf = decltype(buf_tag)::value, i = decltype(i_tag)::value;
        if (MPQX & 32) return;                                            // (timing only)
        if constexpr (i < 8) dma16i<(i - 4) * 1024, buf * UB4 * 4>(ubw, lane16, us_w);
        else dma16i<3072, buf * UB4 * 4 + 1024>(ubw, lane16 + 1024u, us_w);
    };
    auto u_ptr = [&](int slice) __attribute__((always_inline)) -> const float* { return p.wpack + (long long)slice * NC * UB4; };
    auto load_prm = [&](int vslice) __attribute__((always_inline)) {
        const int slice = SPLIT ? vslice >> p.ks_shift : vslice;
        if (tid < 64) {
            prm[tid] = p.bias[slice * 64 + tid]; prm[64 + tid] = p.scale[slice * 64 + tid]; prm[128 + tid] = p.shift[slice * 64 + tid];
        }
    };

    // ---- input transform: lane = 2 * tile + cp owns the window (tile, channel pair cp) and this wave's half of its rows ----
    typedef const __attribute__((address_space(3))) f32x2* lds_pair_ptr;
    // granule of pixel (4 ty + k, 4 tx + c) = lane base + k P + (k >> 2) + c; rows e0 / e2 / e4 of half 1 are one row lower
    // coefficients of this wave's half: rows 1, 2 use (b^2, a), rows 3, 4 use (a^2, b)
    const unsigned long long k_sq = half ? K_B2A2 : K_A2B2;           // .hi = the square the half's rows subtract: b^2 (half 0), a^2 (half 1)
    const unsigned long long k_pt = half ? K_BA : K_AB;               // .lo = the half's point: a (half 0), b (half 1)
    // V' rows this wave writes: 6 * row * 64 floats into a channel region: half 0 rows 0, 1, 2; half 1 rows 5, 3, 4
    const unsigned tf_row0 = (half ? 5u : 0u) * (6u * 64u * 4u), tf_row1 = (half ? 3u : 1u) * (6u * 64u * 4u), tf_row2 = (half ? 4u : 2u) * (6u * 64u * 4u);
    f32x2 hx[3][6];               // X[row k of the half][column]: the column pass's result, consumed by the row pass one unit later
    f32x2 hd[7];                  // a column's inputs e0, e2, e4, m1..m4
    f32x2 ho[6];
    unsigned tf_ra = 0, tf_re = 0;                  // ... of the raw buffer the column pass reads (set once per unit)
    auto tf_read = [&](const int c) __attribute__((always_inline)) {
        if (MPQX & 1) return;
        auto ro = [](int k) { return (k * P + (k >> 2)) * 2; };      // row k of the window, in 8-byte pairs
        const int o = c * 2;
#pragma unroll
        for (int i = 0; i < 3; ++i) hd[i] = reinterpret_cast<lds_pair_ptr>(tf_re)[o + ro(2 * i)];
#pragma unroll
        for (int i = 0; i < 4; ++i) hd[3 + i] = reinterpret_cast<lds_pair_ptr>(tf_ra)[o + ro(1 + i)];
    };
    auto tf_col = [&](const int c) __attribute__((always_inline)) {          // three rows of X of column c: 6 packed multiply-adds
        if (MPQX & 1) return;
        const f32x2* d = hd;
        hx[0][c] = pk_fma_k<0>(d[0], K_PS, pk_fnma_k<1>(d[1], K_PS, d[2]));      // a^2 b^2 e0 + (e4 - (a^2 + b^2) e2): row 0 / row 5
        const f32x2 t0 = pk_fnma_k<1>(d[4], k_sq, d[6]);                         // m4 - s m2
        const f32x2 t1 = pk_fnma_k<1>(d[3], k_sq, d[5]);                         // m3 - s m1
        hx[1][c] = pk_fma_k<0>(t1, k_pt, t0);                                    // t0 + p t1: row 1 / row 3
        hx[2][c] = pk_fnma_k<0>(t1, k_pt, t0);                                   // t0 - p t1: row 2 / row 4
    };
    auto tf_row = [&](const int k) __attribute__((always_inline)) { if (!(MPQX & 1)) bt6(hx[k], ho); };
    // V'[buf][c][6 row + j][lane], j = 2 jj, 2 jj + 1, both channels: four lane-linear stores (address = M0 + offset + 4 lane)
    const unsigned tf_m0[3] = {vs_lds + tf_row0, vs_lds + tf_row1, vs_lds + tf_row2};
    auto tf_store = [&](const int k, const int jj, auto vbuf_tag) __attribute__((always_inline)) {
        if (MPQX & 1) return;
        constexpr int VBB = decltype(vbuf_tag)::value * VB4 * 4;           // byte offset of the V' buffer
        unsigned keep;
#define MPB_VST(J)                                                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:%6\n\tds_write_addtid_b32 %2 offset:%7\n\t" \
                     "ds_write_addtid_b32 %3 offset:%8\n\tds_write_addtid_b32 %4 offset:%9\n\ts_mov_b32 m0, %0"                    \
                     : "=&s"(keep) : "v"(ho[J][0]), "v"(ho[J][1]), "v"(ho[J + 1][0]), "v"(ho[J + 1][1]), "s"(tf_m0[k]),             \
                       "n"(VBB + (J) * 256), "n"(VBB + VR4 * 4 + (J) * 256), "n"(VBB + ((J) + 1) * 256), "n"(VBB + VR4 * 4 + ((J) + 1) * 256) : "memory")
        if (jj == 0) MPB_VST(0); else if (jj == 1) MPB_VST(2); else MPB_VST(4);
#undef MPB_VST
    };

    // ---- GEMM operands ----
    // U[ch = lane >> 4][cout][pos]: a lane's fragments of 4 consecutive positions are one ds_read_b128 (lanes 144 bytes apart)
    // V'[c = ch & 1][pos][2 * tile + (ch >> 1)]: two positions per ds_read2st64_b32
    f32x4 af[3][2], bf[3];                                                     // rings over position groups of 4
    auto lane_values = [&]() __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                                           // (opaque: not hoisted out of the item loop)
        lane16 = (unsigned)ln * 16u;
        // tile number T of the item: 16 consecutive tiles = 4 tile rows x 4 tile columns (TC4 = 8: T = 16 (tx >> 2) + 4 ty + (tx & 3);
        // TC4 = 4: T = 4 ty + tx)
        const int w_tile = ln >> 1, w_cp = ln & 1;
        const int w_ty = TC4 == 8 ? (w_tile >> 2) & 3 : w_tile >> 2, w_tx = TC4 == 8 ? 4 * (w_tile >> 4) + (w_tile & 3) : w_tile & 3;
        tf_rd = raw_lds + (unsigned)((((4 * w_ty) * P + w_ty + 4 * w_tx) * 4 + 2 * w_cp) * 4);
        a_base = ((ln >> 4) * 64 + cbb * 32 + (ln & 15)) * 36;               // second column block: + 16 * 36
        b_base = ((ln >> 4) & 1) * VR4 + 2 * (tb * 16 + (ln & 15)) + (ln >> 5);
    };
    lane_values();
    tf_eo = half ? (unsigned)P * 16u : 0u;                                     // rows e0 / e2 / e4 of half 1 are one row lower (wave-uniform)

    // ---- prologue ----
    Where cur = decode(item);
    const float* rbase = raw_offsets(cur);
    const float* rsrc = rbase;               // the cursor's unit: rbase + ld_chunk * unit_stride
    int ld_chunk = 0;
    int ld_next_item = item + stride;
    auto ld_advance = [&]() __attribute__((always_inline)) {
        rsrc += unit_stride;
        if (++ld_chunk == NC) {
            ld_chunk = 0;
            if (ld_next_item < item_end) {
                rbase = raw_offsets(decode(ld_next_item));
                ld_next_item += stride;
            }
            rsrc = rbase;
        }
    };
    const float* up = u_ptr(cur.slice);
    // raw(k) lives in buffer k % 3: unit n's column pass reads raw(n+2) and its DMAs send raw(n+4) over raw(n+1)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const unsigned v0 = rvl[0], v1 = rvl[64], v2 = rvl[128];
        raw_dma(rsrc, (unsigned)b * (RB4 * 4u), 0, v0); raw_dma(rsrc, (unsigned)b * (RB4 * 4u), 1, v1); raw_dma(rsrc, (unsigned)b * (RB4 * 4u), 2, v2);
        ld_advance();                                                                                          // raw(0), raw(1), raw(2)
    }
    {
        const float* const u0 = up + wave * 2304 + 1024;
        using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>;
#define MPB_U9(PTR, BUF) u_dma(PTR, BUF{}, std::integral_constant<int, 0>{}); u_dma(PTR, BUF{}, std::integral_constant<int, 1>{}); \
        u_dma(PTR, BUF{}, std::integral_constant<int, 2>{}); u_dma(PTR, BUF{}, std::integral_constant<int, 3>{}); u_dma(PTR, BUF{}, std::integral_constant<int, 4>{}); \
        u_dma(PTR, BUF{}, std::integral_constant<int, 5>{}); u_dma(PTR, BUF{}, std::integral_constant<int, 6>{}); u_dma(PTR, BUF{}, std::integral_constant<int, 7>{}); \
        u_dma(PTR, BUF{}, std::integral_constant<int, 8>{})
        MPB_U9(u0, B0);                                                                 // U(0)
        MPB_U9(u0 + UB4, B1);                                                           // U(1)
#undef MPB_U9
    }
    load_prm(cur.slice);
    dma_wait();
    __syncthreads();
    // pair 0 transforms unit 0 (-> V'[0]), pair 1 unit 1 (-> V'[1])
    {
        const unsigned rb = cbb ? RB4 * 4u : 0u;
        tf_ra = tf_rd + rb; tf_re = tf_ra + tf_eo;
#pragma unroll
        for (int c = 0; c < 6; ++c) { tf_read(c); tf_col(c); }
        if (cbb == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { tf_row(k); tf_store(k, 0, std::integral_constant<int, 0>{}); tf_store(k, 1, std::integral_constant<int, 0>{}); tf_store(k, 2, std::integral_constant<int, 0>{}); }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { tf_row(k); tf_store(k, 0, std::integral_constant<int, 1>{}); tf_store(k, 1, std::integral_constant<int, 1>{}); tf_store(k, 2, std::integral_constant<int, 1>{}); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the V' stores are asm statements: hipcc does not count them
    __syncthreads();
    { const unsigned v0 = rvl[0], v1 = rvl[64], v2 = rvl[128]; raw_dma(rsrc, 0u, 0, v0); raw_dma(rsrc, 0u, 1, v1); raw_dma(rsrc, 0u, 2, v2); }
    ld_advance();                                                                      // raw(3) over raw(0)
    // ring state at unit n: r_rd = buffer of raw(n+2), r_nx = raw(n+3), r_dm = raw(n+4) = the buffer raw(n+1) leaves
    unsigned r_rd = 2u * RB4 * 4u, r_nx = 0u, r_dm = RB4 * 4u;
    auto frag_read = [&](const int slot, const int vb, const int g) __attribute__((always_inline)) {
        af[slot][0] = *reinterpret_cast<const f32x4*>(&Us[vb * UB4 + a_base + 4 * g]);
        af[slot][1] = *reinterpret_cast<const f32x4*>(&Us[vb * UB4 + a_base + 16 * 36 + 4 * g]);
        const float* const v = Vs + vb * VB4 + b_base + 4 * g * 64;
        bf[slot] = f32x4{v[0], v[64], v[128], v[192]};
    };
    frag_read(0, 0, 0); frag_read(1, 0, 1);

    // the whole item loop exists twice: pair 0 (waves 0, 1) runs the column pass in even units, pair 1 in odd units; the choice
    // is made ONCE per wave (two instantiations, no accumulator lives across the choice)
    auto item_loop = [&](auto pair_tag) __attribute__((always_inline)) {
    constexpr int PAIR = decltype(pair_tag)::value;
    for (;;) {
        f32x4 acc[36][2];
        const int item_next = item + stride;
        const bool has_next = item_next < item_end;
        const int next_slice = has_next ? (int)(item_next - (int)udiv((unsigned)item_next, p.magic_slices, (unsigned)p.nslices) * p.nslices)
                                        : cur.slice;
        const float* unext = u_ptr(next_slice);
        // one unit: 72 MFMAs and everything that rides in their shadow -- one basic block.  ROLE 0: column pass of unit n+2 (reads
        // raw(n+2)), ROLE 1: row pass of unit n+1 (writes V'[vb ^ 1]).  Slot (g, q): behind MFMA q = 2 e + m of group g.
        auto unit_body = [&](const int c, auto first_tag, auto role_tag, auto vb_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr int ROLE = decltype(role_tag)::value;
            constexpr int vb = decltype(vb_tag)::value;
            const float* const un2 = (c + 2 < NC ? up + (long long)(c + 2) * UB4 : unext + (long long)(c + 2 - NC) * UB4) + (wave * 2304 + 1024);    // U(n+2) -> U[vb]
            const float* const un1 = (c + 1 < NC ? up + (long long)(c + 1) * UB4 : unext + (long long)(c + 1 - NC) * UB4) + (wave * 2304 + 1024);    // U(n+1) -> U[vb ^ 1]: its last four blocks
            using VB = std::integral_constant<int, vb>; using VN = std::integral_constant<int, vb ^ 1>;
            if (ROLE == 0 || ROLE == 3) { tf_ra = tf_rd + r_rd; tf_re = tf_ra + tf_eo; }
            unsigned rv0 = 0, rv1 = 0, rv2 = 0;
#define MPB_I(N) std::integral_constant<int, N>{}
#pragma unroll
            for (int g = 0; g < 9; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int s = 4 * g + e, q = 2 * e + m;
                        // weights as the A operand: D[cout][tile] -- lane = tile, register quad = 4 consecutive output channels
                        MPB_MFMA(acc[s][m], af[g % 3][m][e], bf[g % 3][e], s * 2 + m, FIRST);
                        __builtin_amdgcn_sched_barrier(0);
                        // fragments two groups ahead; groups 7, 8 fetch groups 0, 1 of the NEXT unit (behind the unit barrier)
                        if (q == 0) frag_read((g + 2) % 3, g + 2 < 9 ? vb : vb ^ 1, g + 2 < 9 ? g + 2 : g - 7);
                        // DMAs: U(n+2) -> U[vb] from behind this unit's barrier (every fragment of U[vb] has been fetched) to group 1 of
                        // the next unit, one per three MFMAs; raw(n+4) in groups 2, 3 -- apart from the weights in time (the memory
                        // pipe returns in order: a weight DMA queued behind a patch DMA comes back at HBM latency)
#if MPB_DMA_EARLY
                        // (variant: all nine weight blocks behind the barrier, the patch DMAs two groups earlier)
                        if (g == 7) { if (q == 1) u_dma(un2, VB{}, MPB_I(0)); if (q == 2) u_dma(un2, VB{}, MPB_I(1)); if (q == 4) u_dma(un2, VB{}, MPB_I(2)); if (q == 5) u_dma(un2, VB{}, MPB_I(3)); if (q == 7) u_dma(un2, VB{}, MPB_I(4)); }
                        else if (g == 8) { if (q == 1) u_dma(un2, VB{}, MPB_I(5)); if (q == 3) u_dma(un2, VB{}, MPB_I(6)); if (q == 5) u_dma(un2, VB{}, MPB_I(7)); if (q == 7) u_dma(un2, VB{}, MPB_I(8)); }
                        else if (g == 0) { if (q == 2) raw_dma(rsrc, r_dm, 0, rv0); if (q == 5) raw_dma(rsrc, r_dm, 1, rv1); }
                        else if (g == 1) { if (q == 1) raw_dma(rsrc, r_dm, 2, rv2); }
                        if (g == 0 && q == 0) { rv0 = rvl[0]; rv1 = rvl[64]; rv2 = rvl[128]; }
#else
                        if (g == 7) { if (q == 1) u_dma(un2, VB{}, MPB_I(0)); if (q == 4) u_dma(un2, VB{}, MPB_I(1)); if (q == 7) u_dma(un2, VB{}, MPB_I(2)); }
                        else if (g == 8) { if (q == 2) u_dma(un2, VB{}, MPB_I(3)); if (q == 5) u_dma(un2, VB{}, MPB_I(4)); }
                        else if (g == 0) { if (q == 1) u_dma(un1, VN{}, MPB_I(5)); if (q == 4) u_dma(un1, VN{}, MPB_I(6)); if (q == 7) u_dma(un1, VN{}, MPB_I(7)); }
                        else if (g == 1) { if (q == 2) u_dma(un1, VN{}, MPB_I(8)); }
                        else if (g == 2) { if (q == 1) raw_dma(rsrc, r_dm, 0, rv0); if (q == 5) raw_dma(rsrc, r_dm, 1, rv1); }
                        else if (g == 3) { if (q == 1) raw_dma(rsrc, r_dm, 2, rv2); }
                        if (g == 1 && q == 5) { rv0 = rvl[0]; rv1 = rvl[64]; rv2 = rvl[128]; }      // the gather offsets of this unit's patch DMAs
#endif
                        // ROLE 0: column pass of unit n+2 (column c2 read and evaluated in group c2).  ROLE 1: row pass of unit n+1 -> V'[vb ^ 1]
                        // (row k evaluated in group 2 k, stored four dwords per slot).  ROLE 2: nothing.  ROLE 3 (pair 1, last unit of an item):
                        // the column pass of unit n+2 AND, behind the barrier, its row pass -> V'[vb], which every wave has finished reading
                        // by then -- so that no X lives across the epilogue (36 registers that hipcc would spill)
                        if (ROLE == 0 || ROLE == 3) {
                            if (g < 6 && q == 1) tf_read(g);
                            if (g < 6 && q == 6) tf_col(g);
                        }
                        if (ROLE == 1) {
                            if ((g & 1) == 0 && g < 6 && q == 3) tf_row(g >> 1);
                            if ((g & 1) == 0 && g < 6 && q == 5) tf_store(g >> 1, 0, VN{});
                            if ((g & 1) == 0 && g < 6 && q == 7) tf_store(g >> 1, 1, VN{});
                            if ((g & 1) == 1 && g < 6 && q == 2) tf_store(g >> 1, 2, VN{});
                        }
                        if (ROLE == 3 && g >= 7) {
                            const int t = (g - 7) * 8 + q;                                           // 12 actions in the 16 slots behind the barrier
                            if (t < 12) { if ((t & 3) == 0) tf_row(t >> 2); else tf_store(t >> 2, (t & 3) - 1, VB{}); }
                        }
                        if (g == 6 && q == 7) {
                            // unit barrier: every fragment of the unit has been fetched (two groups ahead), V'(n+1) is written, U(n+1)
                            // and raw(n+3) have landed; raw(n+4) (this unit's three DMAs) stays in flight
                            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
                            __syncthreads();
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#undef MPB_I
            ld_advance();
            const unsigned t = r_rd; r_rd = r_nx; r_nx = r_dm; r_dm = t;          // rotate the raw ring
        };
        using T = std::true_type; using F = std::false_type;
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>;
        using R2 = std::integral_constant<int, 2>; using R3 = std::integral_constant<int, 3>;
        if constexpr (PAIR == 0) {          // column pass in even units, row pass in odd units
            unit_body(0, T{}, R0{}, I0{});
            unit_body(1, F{}, R1{}, I1{});
            for (int c = 2; c < NC; c += 2) {
                unit_body(c, F{}, R0{}, I0{});
                unit_body(c + 1, F{}, R1{}, I1{});
            }
        } else {                            // one unit later: nothing in unit 0, column pass in odd units, row pass in even units, and the
            unit_body(0, T{}, R2{}, I0{});  // last unit's column pass followed by its row pass
            for (int c = 1; c < NC - 1; c += 2) {
                unit_body(c, F{}, R0{}, I1{});
                unit_body(c + 1, F{}, R1{}, I0{});
            }
            unit_body(NC - 1, F{}, R3{}, I1{});
        }

        // ---- output transform Y = A^T M A in registers, bias / ReLU / BN, [2x2 max-pool], store ----
        // lane = tile (lane & 15) of the wave's tile block, registers r = output channels 4 * (lane >> 4) + r of column block m
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");               // the last MFMAs' results (asm statements: no hazard tracking)
        if (MPQX & 8) {
            float sink = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 36; ++s2) sink += acc[s2][0][0] + acc[s2][1][3];
            if (sink == 123.456f) p.out[tid] = sink;
        } else {
            int eln = lane;
            asm volatile("" : "+v"(eln));                            // (opaque: the epilogue's lane constants are not hoisted out of the item loop)
            const int tl = tb * 16 + (eln & 15);                        // tile T of the item (numbering: lane_values())
            const int t_ty = TC4 == 8 ? (tl >> 2) & 3 : tl >> 2, t_tx = TC4 == 8 ? 4 * (tl >> 4) + (tl & 3) : tl & 3;
            const int oy = cur.y0 + 4 * t_ty, ox = cur.x0 + 4 * t_tx;
            const int cs = p.out_cstride;
            constexpr int NO = POOL ? 2 : 4;                            // output rows / columns per tile
            const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
            const int py0 = POOL ? oy >> 1 : oy, px0 = POOL ? ox >> 1 : ox;
            // MODE 0: the ordinary epilogue; 1 (SPLIT): this range's pre-bias output tiles -> split_scratch
            unsigned long long* const part = SPLIT ? reinterpret_cast<unsigned long long*>(p.split_scratch) + (long long)item * (64 * 256) + tid : nullptr;
            auto epilogue = [&](auto mode_tag) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
            const int cl = cbb * 32 + m * 16 + 4 * (eln >> 4);          // first of this lane's 4 output channels in the slice
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(&prm[cl]);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(&prm[64 + cl]);
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(&prm[128 + cl]);
            const int ch0 = (SPLIT ? cur.slice >> p.ks_shift : cur.slice) * 64 + cl;
            f32x2 keep[NO][NO];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 tcol[4][6];                                       // T[a][j] = sum_i A^T[a][i] M[i][j]
                {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x2 mm[6], y[4];
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        float lo, hi;
                        MPB_ACC_RD(lo, acc[6 * i + j][m], (6 * i + j) * 2 + m, 2 * h);
                        MPB_ACC_RD(hi, acc[6 * i + j][m], (6 * i + j) * 2 + m, 2 * h + 1);
                        mm[i] = f32x2{lo, hi};
                    }
                    at6s(mm, y);
#pragma unroll
                    for (int a = 0; a < 4; ++a) tcol[a][j] = y[a];
                    __builtin_amdgcn_sched_barrier(0);      // (register peak: the scheduler otherwise hoists every accumulator read)
                }
                }
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        f32x2 y[4];
                        at6s(tcol[a], y);
#pragma unroll
                        for (int b = 0; b < 4; ++b) part[((m * 2 + h) * 16 + a * 4 + b) * 256] = __builtin_bit_cast(unsigned long long, y[b]);
                    }
                } else {
                const f32x2 bb = {b4[2 * h], b4[2 * h + 1]}, ss = {s4[2 * h], s4[2 * h + 1]}, tt = {t4[2 * h], t4[2 * h + 1]};
                // row a of the 4x4 output tile, pre-bias
                auto out_row = [&](const int a, f32x2 (&y)[4]) __attribute__((always_inline)) { at6s(tcol[a], y); };
                auto act = [&](f32x2 v, const int r, const int c) __attribute__((always_inline)) -> f32x2 {
                    v = w43_add_bias(v, r, c, bb);
                    if (BNF) { v = v * ss + tt; return f32x2{relu_q(v[0]), relu_q(v[1])}; }
                    v = f32x2{relu_q(v[0]), relu_q(v[1])};
                    return v * ss + tt;
                };
                // a uniform per-image base + 32-bit byte offsets (an image's output is far below 4 GB).  NHWC: pixel stride
                // cs floats; planar [B][cout/4][Ho][Wo][4]: this lane's quad is plane ch0 / 4, pixel stride 16 bytes
                char* const img_base = reinterpret_cast<char*>(
                    p.out_planar ? p.out + (long long)cur.img * (p.cout / 4) * Ho * Wo * 4
                                 : p.out + (long long)cur.img * Ho * Wo * cs + p.out_coff);
                const unsigned ps = p.out_planar ? 16u : (unsigned)cs * 4u;                       // bytes per pixel step
                const unsigned rs = (unsigned)Wo * ps;                                            // bytes per row step
                const unsigned o0 = p.out_planar ? (unsigned)(((ch0 >> 2) * Ho + py0) * Wo + px0) * 16u
                                                 : (unsigned)((py0 * Wo + px0) * cs + ch0) * 4u;
                const bool inside = py0 + NO <= Ho && px0 + NO <= Wo;     // the whole tile lies inside the frame
                // one output row (pooled: one row pair) at a time: the register peak is what decides whether loop-carried
                // values survive the epilogue in registers (a reload inside a unit body stalls on every DMA in flight)
#pragma unroll
                for (int a = 0; a < NO; ++a) {
                    __builtin_amdgcn_sched_barrier(0);
                    f32x2 res[NO];
                    if constexpr (POOL) {
                        f32x2 y0[4], y1[4];
                        out_row(2 * a, y0); out_row(2 * a + 1, y1);
#pragma unroll
                        for (int b = 0; b < 4; ++b) { y0[b] = act(y0[b], 2 * a, b); y1[b] = act(y1[b], 2 * a + 1, b); }
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int r = 0; r < 2; ++r)
                                res[b][r] = fmaxf(fmaxf(y0[2 * b][r], y0[2 * b + 1][r]), fmaxf(y1[2 * b][r], y1[2 * b + 1][r]));
                    } else {
                        f32x2 y0[4];
                        out_row(a, y0);
#pragma unroll
                        for (int b = 0; b < 4; ++b) res[b] = act(y0[b], a, b);
                    }
                    if (h == 0) {
#pragma unroll
                        for (int b = 0; b < NO; ++b) keep[a][b] = res[b];
                    } else if (ch0 < p.cout) {
                        if (inside) {                                     // unconditional 16-byte stores
#pragma unroll
                            for (int b = 0; b < NO; ++b) {
                                const f32x4 v = {keep[a][b][0], keep[a][b][1], res[b][0], res[b][1]};
                                *reinterpret_cast<f32x4*>(img_base + (o0 + (unsigned)a * rs + (unsigned)b * ps)) = v;
                            }
                        } else {                                          // frame edge of a frame that is no multiple of the tile (or a phantom tile)
#pragma unroll
                            for (int b = 0; b < NO; ++b)
                                if (py0 + a < Ho && px0 + b < Wo) {
                                    const f32x4 v = {keep[a][b][0], keep[a][b][1], res[b][0], res[b][1]};
                                    *reinterpret_cast<f32x4*>(img_base + (o0 + (unsigned)a * rs + (unsigned)b * ps)) = v;
                                }
                        }
                    }
                }
                }
            }
            }
            };
            // (SPLIT: split_reduce_kernel, the next launch on the stream, sums the ranges' shares in range order and runs the rest)
            epilogue(std::integral_constant<int, SPLIT ? 1 : 0>{});
        }
        if (!has_next) { dma_wait(); return; }      // the prefetch DMAs still in flight write THIS workgroup's LDS: drain them
        if (next_slice != cur.slice) {
            __syncthreads();
            load_prm(next_slice);
        }
        item = item_next;
        cur = decode(item);
        up = unext;
        // the next item's first fragments: fetched again here rather than carried across the epilogue in 24 registers (the last unit
        // of an item prefetches them like every unit; those copies die at the loop's exit)
        lane_values();
        frag_read(0, 0, 0); frag_read(1, 0, 1);
    }
    };
    if (cbb == 0) item_loop(std::integral_constant<int, 0>{}); else item_loop(std::integral_constant<int, 1>{});
}

template <bool POOL, int TC4, bool ZPAD, bool SPLIT = false>
int launch_q(const ConvParams& p, hipStream_t s)
{
    constexpr int OY = 4 * (32 / TC4), OX = 4 * TC4;
    ConvParams q = p;
    if (SPLIT) q.nslices = p.nslices << p.ks_shift;         // virtual slices: (slice, range of input channels)
    q.tiles_x = (p.W + OX - 1) / OX; q.tiles_y = (p.H + OY - 1) / OY;
    const long long nitems = (long long)p.B * q.tiles_x * q.tiles_y * q.nslices;
    if (nitems <= 0) return 0;
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); };
    q.magic_slices = magic(q.nslices); q.magic_tx = magic(q.tiles_x); q.magic_ty = magic(q.tiles_y);
    const long long dmax = std::max(std::max(q.nslices, q.tiles_x), q.tiles_y);
    if (nitems * dmax >= 0x100000000ll) return 1;
    q.nitems = (int)nitems;
    const unsigned grid = persistent_grid(nitems, p.ncu, p.xcd_shift);
    const ConvParams& pp = q;
    if (p.bn_first) hipLaunchKernelGGL((conv_wino43b_kernel<POOL, true, TC4, ZPAD, SPLIT>), dim3(grid), dim3(256), 0, s, pp);
    else hipLaunchKernelGGL((conv_wino43b_kernel<POOL, false, TC4, ZPAD, SPLIT>), dim3(grid), dim3(256), 0, s, pp);
    return 0;
}

// the item shape that covers the frame with fewer items (16 x 32 pixels on a tie: longer contiguous patch rows)
template <bool POOL, bool ZPAD, bool SPLIT = false>
int launch_shape(const ConvParams& p, hipStream_t s)
{
    const long long wide = (long long)((p.W + 31) / 32) * ((p.H + 15) / 16), tall = (long long)((p.W + 15) / 16) * ((p.H + 31) / 32);
    return tall < wide ? launch_q<POOL, 4, ZPAD, SPLIT>(p, s) : launch_q<POOL, 8, ZPAD, SPLIT>(p, s);
}

}  // namespace

// true when launch_conv_wino43b handles this layer shape: input channels a multiple of 8 (units of 4, walked in pairs), output
// channels a multiple of 4 (16-byte stores); reflection OR zero padding; ANY frame of at least 2 x 2 pixels (tiles that stick out
// of the frame are masked in the epilogue)
bool conv_wino43b_supports(const ConvParams& p)
{
    return p.cin % 8 == 0 && p.cout % 4 == 0 && p.H >= 2 && p.W >= 2 &&
           p.in_cstride % 4 == 0 && p.in_coff % 4 == 0 && p.out_cstride % 4 == 0 && p.out_coff % 4 == 0;
}

int launch_conv_wino43b(const ConvParams& p, bool pool, hipStream_t s)
{
    if (p.ks_shift > 0) {        // api.hip: single-pair launches with fewer items than CUs
        const int ncs = (p.cin / UC4) >> p.ks_shift;
        if (ncs < 2 || (ncs & 1) || (ncs << p.ks_shift) * UC4 != p.cin || !p.split_scratch) return 2;
        const int rc = p.pad_zero ? (pool ? launch_shape<true, true, true>(p, s) : launch_shape<false, true, true>(p, s))
                                  : (pool ? launch_shape<true, false, true>(p, s) : launch_shape<false, false, true>(p, s));
        return rc ? rc : launch_split_reduce(p, 2, pool, s);
    }
    if (p.pad_zero) return pool ? launch_shape<true, true>(p, s) : launch_shape<false, true>(p, s);
    return pool ? launch_shape<true, false>(p, s) : launch_shape<false, false>(p, s);
}

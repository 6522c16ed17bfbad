// 3x3 convolution by Winograd F(4x4, 3x3) on the fp32 matrix instruction: 36 element-wise GEMMs
//   M[pos] (tiles x cout) += V[pos] (tiles x cin) * U[pos] (cin x cout),  pos = 0..35
// per 4x4 output pixels -- 2.25 multiplications per output against 4 for F(2x2,3x3) (conv_wino.hip) and 9 for the direct
// form: 1.78x fewer MFMAs than conv_wino.hip for the same ReflectionPad -> Conv2d -> ReLU -> BN [-> MaxPool] block
// (multipoint/models/MultiPoint.py:143-148).  fp32 throughout; U = G g G^T is computed once on the host (in double, rounded
// once), V = B^T d B and Y = A^T M A are short fixed-order multiply-add chains whose coefficients are exact binary fractions.
// Interpolation points {0, +-3/4, +-3/2, inf} instead of the textbook {0, +-1, +-2, inf}: same instruction count, 3.4x smaller
// maximum error (mp_common.h; measured on trained-like statistics: docs/HISTORY.md section 4).
//
// Structure (the round-2 lessons of conv_wino.hip apply unchanged: every operand through LDS, filled by LDS-DMA; ONE counted
// wait + barrier per unit; no control flow inside a unit; vector work clustered):
//  * Persistent workgroups, ONE per CU, 512 threads = 8 waves (two per SIMD).  Item = 32 tiles (4 rows x 8 columns of 4x4
//    pixels = 16 x 32 output pixels, or -- TC4 = 4 -- 8 x 4 = 32 x 16 pixels where that covers the frame with fewer items)
//    x 64 output channels.  Wave w multiplies tile block w&1 (16 tiles) by channel block w>>1
//    (16 couts) for ALL 36 positions on v_mfma_f32_16x16x4_f32 (32 cycles): 36 accumulators of 4 registers = 144, held in
//    ordinary VGPRs, so the output transform is in-register (a lane owns ONE tile and 4 consecutive output channels) and needs
//    neither an exchange between waves nor accumulator reads.
//  * K is walked in units of 4 input channels = one MFMA per position.  LDS per unit: V 18 KiB [ch][tile][pos] and U 36 KiB
//    [ch][cout][pos] (a lane's operands of 4 consecutive positions are ONE conflict-free ds_read_b128: lanes are 144 bytes
//    apart), both double-buffered; raw 10 KiB (18 x 34 patch x 4 channels, one 16-byte granule per pixel = LDS-DMA order) in a
//    ring of THREE; + a per-wave scratch for the input transform: 158 KiB.
//  * Input layout: NHWC, or channel-quad-planar [B][C/4][H][W][4] when the producer is conv_first.hip or a pooled launch of
//    this kernel (api.hip decides per tensor): a unit's patch rows are then contiguous, 9-11 cache lines per DMA instruction
//    instead of 64.
//  * DMA order: the memory pipe returns in order across the CU, so a weight DMA (L2 hit) queued behind a patch DMA (HBM miss)
//    of ANY wave comes back at HBM latency.  The 5 weight DMAs of unit n+2 therefore go out right behind the barrier of unit
//    n, the 2 patch DMAs of unit n+3 six MFMA groups ahead of the next weights, and the barrier waits with vmcnt(2) --
//    for everything but the patch DMAs, which have until the next barrier (62.6 k -> 58.9 k cycles per item against
//    patch-first order with a full wait).
//  * Input transform of unit n+1 while unit n is multiplied: 8 lanes per (tile, channel pair) window; lanes 0-5 transform
//    one COLUMN of the 6x6 window each (12 packed instructions, written as asm: hipcc scalarises a third of them), hand the
//    result over through the wave's own LDS scratch (LDS operations of a wave execute in order: no barrier), then transform
//    one ROW each and write V with ds_write2_b32 of the registers as they are.  Reads, arithmetic and stores of the two passes
//    are spread over groups 0-6 of the unit, the stores two per MFMA gap.
#include "mp_common.h"

#include <algorithm>
#include <type_traits>

#ifdef MP_TIMING
// developer instrumentation (tools/conv_timing_wino.py with MP_TIMING_KERNEL=43): per-workgroup cycle sums per phase, wave 0
__device__ unsigned long long g_timing_q[256 * 8];
__device__ int g_timing_q_sel = 480;
extern "C" int mp_debug_select_height_wino43(int h) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_timing_q_sel), &h, sizeof(int)); }
extern "C" int mp_debug_read_timing_wino43(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timing_q), sizeof(unsigned long long) * n);
}
#define MPQ_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define MPQ_ADD(slot, a, b) do { tsum[slot] += (b) - (a); } while (0)
#else
#define MPQ_T(var) do { } while (0)
#define MPQ_ADD(slot, a, b) do { } while (0)
#endif

#ifndef MPQX
#define MPQX 0     // developer elimination switches (timing only, results WRONG): 1 no input transform, 2 no DMA, 4 no fragment reads, 8 no epilogue
#endif
namespace {

// an item's 32 tiles are 4 rows x 8 columns (16 x 32 output pixels, raw patch 18 x 34) or -- TC4 = 4 -- 8 rows x 4 columns (32 x 16
// pixels, patch 34 x 18): launch_q picks the shape with fewer phantom tiles (60 x 80 layers: 8 instead of 12 items per image)
constexpr int NPIX = 18 * 34;                      // 612 patch pixels = 16-byte granules (4 channels each), either shape
constexpr int UC4 = 4;                             // input channels per unit
constexpr int VB4 = UC4 * 32 * 36;                 // floats per V buffer  [ch][tile][pos]   (18 KiB)
constexpr int UB4 = UC4 * 64 * 36;                 // floats per U buffer  [ch][cout][pos]   (36 KiB)
constexpr int NRB = (NPIX + 63) / 64;              // raw DMA blocks of 64 granules (10)
constexpr int RB4 = NRB * 64 * 4;                  // floats per raw buffer: 10 blocks (10 KiB); one dummy block behind the three
constexpr int SW4 = 8 * 36 * 2;                    // floats of a wave's transform scratch: 8 windows x 36 x (2 channels)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int reflect_clamp_q(int v, int n)
{
    v = v < 0 ? -v : v;
    v = v >= n ? 2 * (n - 1) - v : v;
    v = v < 0 ? 0 : v;
    return v >= n ? n - 1 : v;
}
__device__ __forceinline__ float relu_q(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
// LDS-DMA, see conv_wino.hip (hazards in front of the statement are checked at build time: multipoint_amd/build.py)
__device__ __forceinline__ void dma16(const float* sbase, unsigned voff_bytes, unsigned lds_byte)
{
    if (MPQX & 2) return;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_byte) : "memory");
}
// the same behind a wave-uniform branch: the five wait states an SGPR base needs behind a VALU write must lie INSIDE the branch's
// own block (multipoint_amd/build.py checks it), so the statement opens with two more
__device__ __forceinline__ void dma16b(const float* sbase, unsigned voff_bytes, unsigned lds_byte)
{
    if (MPQX & 2) return;
    unsigned keep;
    asm volatile("s_nop 1\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_byte) : "memory");
}
// 4 bytes per lane: 64 lanes x 4 bytes from (uniform base + per-lane byte offset) to LDS [lds_byte + 4 * lane, + 4)
__device__ __forceinline__ void dma4(const float* sbase, unsigned voff_bytes, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }

// Packed fp32 arithmetic as explicit instructions: hipcc scalarises a third of the transform's packed multiply-adds (4 v_fma_f32
// for 2 v_pk_fma_f32 per pass), and next to an MFMA stream every vector instruction costs matrix-pipe time (docs/HISTORY.md A.3).
// The transform coefficients come in scalar register pairs (VOP3P takes no literal on gfx950).
// Two coefficients share one scalar register pair (low / high half, picked by op_sel: the selected half feeds both lanes), so the
// six coefficients of the input transform occupy three pairs instead of six (SGPRs are what the fused-first-block instantiation
// is shortest of).
constexpr unsigned long long pk_const2(double lo, double hi)
{
    return (unsigned long long)__builtin_bit_cast(unsigned, (float)lo) | ((unsigned long long)__builtin_bit_cast(unsigned, (float)hi) << 32);
}
constexpr double W43A = MP_W43_A, W43B = MP_W43_B;                  // interpolation points {0, +-a, +-b, inf} (mp_common.h)
constexpr unsigned long long K_AB = pk_const2(W43A, W43B), K_A2B2 = pk_const2(W43A * W43A, W43B * W43B),
                             K_PS = pk_const2(W43A * W43A * W43B * W43B, W43A * W43A + W43B * W43B);
static_assert((double)(float)(W43A * W43A * W43B * W43B) == W43A * W43A * W43B * W43B && (double)(float)(W43A * W43A + W43B * W43B) ==
              W43A * W43A + W43B * W43B, "the transform coefficients must be exact in fp32");
// HI = 0: the low half of k, 1: the high half
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
// 1-D input transform B^T d (6 -> 6), packed over two channels: 12 multiply-adds.  Row of point p = the coefficients of
// x (x^2 - a^2)(x^2 - b^2) / (x - p), last row the polynomial itself:
//   B^T = [a^2 b^2, 0, -(a^2+b^2), 0, 1, 0;  0, -+a b^2, -b^2, +-a, 1, 0 (p = +-a);  0, -+b a^2, -a^2, +-b, 1, 0 (p = +-b);
//          0, a^2 b^2, 0, -(a^2+b^2), 0, 1]
__device__ __forceinline__ void bt6(const f32x2 d[6], f32x2 r[6])
{
#if defined(MPQX) && (MPQX & 131072)
    for (int i = 0; i < 6; ++i) r[i] = d[i];        // timing only: no arithmetic
    return;
#endif
    const f32x2 t0 = pk_fnma_k<1>(d[2], K_A2B2, d[4]);      // d4 - b^2 d2      (even part of the +-a rows)
    const f32x2 t1 = pk_fnma_k<1>(d[1], K_A2B2, d[3]);      // d3 - b^2 d1      (odd part / a)
    const f32x2 t2 = pk_fnma_k<0>(d[2], K_A2B2, d[4]);      // d4 - a^2 d2      (+-b rows)
    const f32x2 t3 = pk_fnma_k<0>(d[1], K_A2B2, d[3]);      // d3 - a^2 d1
    r[0] = pk_fma_k<0>(d[0], K_PS, pk_fnma_k<1>(d[2], K_PS, d[4]));      // a^2 b^2 d0 + (d4 - (a^2+b^2) d2)
    r[1] = pk_fma_k<0>(t1, K_AB, t0);                       // t0 + a t1
    r[2] = pk_fnma_k<0>(t1, K_AB, t0);                      // t0 - a t1
    r[3] = pk_fma_k<1>(t3, K_AB, t2);                       // t2 + b t3
    r[4] = pk_fnma_k<1>(t3, K_AB, t2);                      // t2 - b t3
    r[5] = pk_fma_k<0>(d[1], K_PS, pk_fnma_k<1>(d[3], K_PS, d[5]));      // a^2 b^2 d1 + (d5 - (a^2+b^2) d3)
}
// (1-D output transform: at6s(), mp_common.h)

// F1: the layer's input is the first encoder block (Cin = 1 -> 64, conv_first.hip's arithmetic) of p.img, computed by this
// kernel itself, PER UNIT and straight into the raw LDS ring (round 3; round 2 evaluated it per item into a global scratch that
// the unit loop DMA'd back: 12 GB of fabric traffic per launch and a phase of 11 k cycles per item with the matrix pipe idle --
// an L2-resident scratch would have been 7 % faster, the phase-free bound 22 %).  While unit n is multiplied, the 4 channels of
// unit n + 2 are produced for the item's 18 x 34 patch by v_mfma_f32_4x4x1_16b_f32: 16 blocks of D[4 channels][4 pixels] +=
// W[4][1] X[1][4] per instruction, nine taps = nine instructions per 64 pixels, the bias as the accumulator's initial value.  A
// lane supplies ITS pixel's tap value (gathered from a 20 x 36 fp32 image patch in LDS whose rows and columns are staged
// already reflected) and the weight of channel lane & 3, and receives the 4 channels of its own pixel -- exactly one raw
// granule, written after ReLU / BatchNorm with one ds_write_b128.  Nothing of the block ever leaves the CU; the image patch of
// item k + 2 arrives by 4-byte LDS-DMA during the epilogue of item k.
// VIN (round 5: layers with MANY output slices -- the 3x3 head convolutions, 512 couts = 8 slices): the input arrives already
// TRANSFORMED.  Every 64-cout slice of a tile block needs the same V = B^T d B, and the in-kernel transform is what the matrix pipe
// waits for (docs/HISTORY.md Appendix B: 2470 cycles per unit without it, 3300-3650 with); wino43_vprod_kernel below writes V once per (tile block,
// unit) into p.vglobal in exactly the LDS order [unit][ch 4][tile 32][pos 36] (18 KiB, the same bits the in-kernel transform
// produces), and this instantiation DMAs it straight into a ring of FOUR V buffers three units ahead: no raw patches, no
// scratch, no vector work in the unit body at all.  The 8 slices of a tile block are consecutive items = 8 workgroups of one XCD at
// the same time: one of them misses to HBM, the others hit the XCD's L2.
template <bool POOL, bool BNF, int TC4, bool F1, bool SPLIT = false, bool VIN = false>
__global__ __launch_bounds__(512, 2) void conv_wino43_kernel(const ConvParams p)
{
    static_assert(!(F1 && SPLIT), "the fused first block is never split");
    static_assert(!(VIN && (F1 || SPLIT)), "pre-transformed input: plain launches only");
    constexpr int TR4 = 32 / TC4;                      // tile rows x tile columns of an item
    constexpr int OY = 4 * TR4, OX = 4 * TC4;          // output pixels of an item
    constexpr int PY = OY + 2, PX = OX + 2;            // raw patch
    static_assert(PY * PX == NPIX, "item shape");
    __shared__ __attribute__((aligned(16))) float Vs[(VIN ? 4 : 2) * VB4];
    __shared__ __attribute__((aligned(16))) float Us[2 * UB4];
    __shared__ __attribute__((aligned(16))) float raw[VIN ? 256 : 3 * RB4 + 256];        // VIN: only the dummy DMA block
    __shared__ __attribute__((aligned(16))) float scr[VIN ? 4 : 8 * SW4];
    // bias | BatchNorm scale | shift per output channel of the slice; POOL && !SPLIT (the epilogue pools before it activates): | sign of the
    // scale (+-1) | integer clamp bounds of the ReLU on the sign-folded value, and the scale slot holds |scale|
    __shared__ __attribute__((aligned(16))) float prm[(POOL && !SPLIT ? 6 : 3) * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = wave & 1, cb = wave >> 1;             // tile block (16 tiles), channel block (16 couts) of this wave's GEMMs
    // units per item (even: cin is a multiple of 8).  SPLIT: the launcher cut the input channels into 2^ks_shift ranges -- an
    // item is (tile block, virtual slice = slice * ranges + range) and runs the units of its range only
    const int NC = SPLIT ? (p.cin / UC4) >> p.ks_shift : p.cin / UC4;

    // ---- work items: (tile block, slice) of this XCD's contiguous eighth ----
    const XcdRange xr = xcd_range(p.nitems, p.xcd_shift);
    const int stride = xr.stride, item_end = xr.item_end;
    int item = xr.item;
    if (item >= item_end) return;

    auto udiv = [](unsigned n, unsigned magic, unsigned d) -> unsigned { return d == 1 ? n : __umulhi(n, magic); };
    struct Where { int slice, img, y0, x0, tile; const float* in_base; };
    auto decode = [&](int it) __attribute__((always_inline)) -> Where {
        Where w{};
        const int tile = (int)udiv((unsigned)it, p.magic_slices, (unsigned)p.nslices);
        w.slice = it - tile * p.nslices;
        w.tile = tile;
        const int trow = (int)udiv((unsigned)tile, p.magic_tx, (unsigned)p.tiles_x);
        const int tx = tile - trow * p.tiles_x;
        const int bi = (int)udiv((unsigned)trow, p.magic_ty, (unsigned)p.tiles_y);
        const int ty = trow - bi * p.tiles_y;
        w.img = p.img_list ? p.img_list[bi] : bi;
        w.y0 = ty * OY; w.x0 = tx * OX;
        // planar input [B][cin/4][H][W][4]: base of the image's first plane; a unit's plane is chunk * H * W * 4 floats on
        w.in_base = p.in_planar ? p.in + (long long)w.img * (p.cin / 4) * p.H * p.W * 4
                                : p.in + (long long)w.img * p.H * p.W * p.in_cstride + p.in_coff;
        if constexpr (SPLIT)      // first unit of the item's range of input channels
            w.in_base += (long long)((w.slice & ((1 << p.ks_shift) - 1)) * NC) * (p.in_planar ? (long long)p.H * p.W * 4 : UC4);
        return w;
    };

    // ---- raw patch staging by DMA: granule f = block * 64 + lane = patch pixel f; this wave issues blocks wave, wave + 8 ----
    const int pix_stride = p.in_planar ? 4 : p.in_cstride;              // floats between horizontally adjacent pixels
    const long long unit_stride = p.in_planar ? (long long)p.H * p.W * 4 : UC4;     // floats between consecutive units
    unsigned rvoff[2];            // byte offset of the granule's source pixel (channel 0 of the unit)
    bool roff_rel = false;        // rvoff holds the item-invariant offsets of interior items
    auto raw_offsets = [&](const Where& w) __attribute__((always_inline)) -> const float* {
        const bool interior = (w.y0 >= 1) && (w.y0 + OY < p.H) && (w.x0 >= 1) && (w.x0 + OX < p.W);
        if (interior) {
            if (!roff_rel) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int f = (wave + 8 * j) * 64 + lane;
                    const int py = f / PX, px = f - py * PX;
                    rvoff[j] = (f < NPIX) ? (unsigned)((py * p.W + px) * pix_stride) * 4u : 0u;
                }
                roff_rel = true;
            }
            return w.in_base + (long long)((w.y0 - 1) * p.W + (w.x0 - 1)) * pix_stride;
        }
        roff_rel = false;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = (wave + 8 * j) * 64 + lane;
            const int py = f / PX, px = f - py * PX;
            unsigned off = 0;
            if (f < NPIX) {
                const int gy = reflect_clamp_q(w.y0 + py - 1, p.H), gx = reflect_clamp_q(w.x0 + px - 1, p.W);
                off = (unsigned)((gy * p.W + gx) * pix_stride) * 4u;
            }
            rvoff[j] = off;
        }
        return w.in_base;
    };
    const unsigned raw_lds = lds_addr(raw), us_lds = lds_addr(Us);
    // raw block j (0, 1) of this wave for channel unit `chunk` of the cursor's item -> the raw buffer at float offset `boff`;
    // waves 2..7 have no second block: their DMA lands in the dummy block behind the three buffers (every wave issues two, so
    // that one s_waitcnt vmcnt(2) leaves exactly the unit's raw DMAs in flight in all of them)
    // (src: the unit's first channel at the patch origin, kept as a running pointer by the cursor; boff_bytes: the raw buffer)
    const unsigned raw_m0 = raw_lds + (unsigned)wave * 1024u;             // block `wave` of buffer 0
    const unsigned raw_dummy = raw_lds + 3u * RB4 * 4u;
    auto raw_dma = [&](const float* src, unsigned boff_bytes, int j) __attribute__((always_inline)) {
        const unsigned dst = j == 0 ? raw_m0 + boff_bytes : (wave + 8 < NRB ? raw_m0 + 8192u + boff_bytes : raw_dummy);
        if (MPQX & 16) return;
        if (MPQX & 128) { dma16(p.in, (unsigned)lane * 16u, dst); return; }   // L1-hot source
        dma16(src, (MPQX & 64) ? (unsigned)lane * 16u : rvoff[j], dst);
    };
    // the unit's 36 weight blocks of 1 KiB over the waves: wave w issues blocks [u_first, u_first + u_cnt), up to three per site (two
    // sites per unit).  No dummy loads: the unit barrier's vmcnt(2) / vmcnt(3) leaves the NEWEST two / three loads in flight -- the raw /
    // V loads every wave issues in equal number behind its weight loads -- however many weight loads precede them.  Plain launches:
    // 5 blocks on waves 0-3, 4 on waves 4-7.  F1: waves 0, 1 also produce the second raw block of every unit (612 patch pixels are 10
    // blocks of 64 for 8 waves), so SIMDs 0, 1 (waves 0, 4 / 1, 5) issue 3 + 4 weight loads and SIMDs 2, 3 (waves 2, 6 / 3, 7) 6 + 5:
    // an LDS-DMA costs ~50 cycles of its SIMD's issue next to the MFMA stream, the extra raw block ~170.
    int u_first, u_cnt;
    if constexpr (F1 && !(MPQX & 33554432)) {
#if defined(MP_DMA_BAL) && MP_DMA_BAL == 0          // (developer A/B) no dummies, not balanced: 5 on waves 0-3, 4 on waves 4-7
        constexpr int cnt_[8] = {5, 5, 5, 5, 4, 4, 4, 4}, first_[8] = {0, 5, 10, 15, 20, 24, 28, 32};
#elif defined(MP_DMA_BAL) && MP_DMA_BAL == 3        // (developer A/B) SIMDs 0, 1: 8, SIMDs 2, 3: 10
        constexpr int cnt_[8] = {4, 4, 5, 5, 4, 4, 5, 5}, first_[8] = {0, 4, 8, 13, 18, 22, 26, 31};
#else
        constexpr int cnt_[8] = {3, 3, 6, 6, 4, 4, 5, 5}, first_[8] = {0, 3, 14, 20, 6, 10, 26, 31};
#endif
        u_cnt = 3; u_first = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { u_cnt = wave == w ? cnt_[w] : u_cnt; u_first = wave == w ? first_[w] : u_first; }
    } else {
        u_cnt = wave < 4 ? 5 : 4; u_first = wave < 4 ? 5 * wave : 20 + 4 * (wave - 4);
    }
    // site 0: blocks k = 0..2 of this wave, site 1: k = 3..5 (wave-uniform branches)
    auto u_dma_site = [&](const float* ub, int buf, int site) __attribute__((always_inline)) {
        if (MPQX & 32) return;
        if constexpr (VIN) {
            // (the pre-transformed-input instantiation sits at 252 registers: the branch-free form with four dummy re-loads of block 35
            // per unit -- five loads on every wave, blocks wave + 8 i -- keeps it free of scratch)
#pragma unroll
            for (int i = 3 * site; i < (site ? 5 : 3); ++i) {
                int b = wave + 8 * i;
                b = b < 36 ? b : 35;
                dma16((MPQX & 256) ? p.wpack : ub + b * 256, (unsigned)lane * 16u, us_lds + (unsigned)(buf * UB4 + b * 256) * 4u);
            }
            return;
        }
#pragma unroll
        for (int k = 3 * site; k < 3 * site + 3; ++k)
            if (k < u_cnt) {
                const int b = u_first + k;
                dma16b((MPQX & 256) ? p.wpack : ub + b * 256, (unsigned)lane * 16u, us_lds + (unsigned)(buf * UB4 + b * 256) * 4u);
            }
    };
    auto u_ptr = [&](int slice) __attribute__((always_inline)) -> const float* {      // unit 0 of a slice
        return p.wpack + (long long)slice * NC * UB4;
    };
    auto load_prm = [&](int vslice) __attribute__((always_inline)) {
        const int slice = SPLIT ? vslice >> p.ks_shift : vslice;
        if (tid < 64) {
            const float sc = p.scale[slice * 64 + tid];
            prm[tid] = p.bias[slice * 64 + tid]; prm[128 + tid] = p.shift[slice * 64 + tid];
            if constexpr (POOL && !SPLIT) {
                const bool neg = sc < 0.f;
                prm[64 + tid] = fabsf(sc); prm[192 + tid] = neg ? -1.f : 1.f;
                prm[256 + tid] = __int_as_float(neg ? (int)0x80000000 : 0); prm[320 + tid] = __int_as_float(neg ? 0 : 0x7fffffff);
            } else {
                prm[64 + tid] = sc;
            }
        }
    };

    // ---- input transform V = B^T d B of one unit: 8 lanes per window (tile, channel pair); lanes 0-5 work ----
    const int win = tid >> 3, sub = tid & 7;             // window 0..63 of the unit: tile = win >> 1, channel pair = win & 1
    const int w_tile = win >> 1, w_cp = win & 1;
    const int w_ty = w_tile / TC4, w_tx = w_tile % TC4;  // tile row / column inside the item
    const int sub6 = sub < 6 ? sub : 5;                  // lanes 6, 7 repeat lane 5's work (results identical, harmless)
    // pass 1: column `sub6` of the window: patch pixels (4*w_ty + i, 4*w_tx + sub6), i = 0..5, channels 2*w_cp, 2*w_cp+1
    const int p1_read = ((4 * w_ty) * PX + 4 * w_tx + sub6) * 4 + 2 * w_cp;             // floats into raw[]
    float* const myscr = scr + wave * SW4 + (win & 7) * 72;                             // this window's 36 pairs
    const unsigned scr_w = lds_addr(myscr + sub6 * 2);                                  // pass 1 stores pair (i, sub6) at + 48 i bytes
    // pass 2: row `sub6`: scratch pairs [sub6][0..5]; V[ch][tile][pos = 6*sub6 + j]
    const int p2_write = (2 * w_cp * 32 + w_tile) * 36 + 6 * sub6;
    f32x2 td[6], tr[6];
    f32x2 tq[6], tq2[6];                                 // MPQX & 65536 only
    // the lane's read position in the raw buffer the NEXT transform reads (two registers: rows 0-3 and rows 4-5 are
    // within a ds_read2_b64's offset range of them); advanced inside the VALU cluster of pass 1b, where an add is cheap
    // (absolute LDS byte addresses, so that the reads take the registers as they are)
    typedef const __attribute__((address_space(3))) f32x2* lds_pair_ptr;
    const unsigned p1_base = lds_addr(raw) + (unsigned)p1_read * 4u;
    unsigned p1_a = p1_base, p1_b = p1_base + 4u * PX * 16u;
    auto tf_pass1 = [&]() __attribute__((always_inline)) {
        if (MPQX & (1 | 16384)) return;
        if (MPQX & 65536) {               // timing only: the reads are issued, nothing depends on them until the unit's end
#pragma unroll
            for (int i = 0; i < 4; ++i) tq[i] = reinterpret_cast<lds_pair_ptr>(p1_a)[i * PX * 2];
#pragma unroll
            for (int i = 4; i < 6; ++i) tq[i] = reinterpret_cast<lds_pair_ptr>(p1_b)[(i - 4) * PX * 2];
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) td[i] = reinterpret_cast<lds_pair_ptr>(p1_a)[i * PX * 2];
#pragma unroll
        for (int i = 4; i < 6; ++i) td[i] = reinterpret_cast<lds_pair_ptr>(p1_b)[(i - 4) * PX * 2];
    };
    auto tf_pass1b = [&](unsigned next_byte) __attribute__((always_inline)) {   // next_byte: raw buffer of the next transform
        if (MPQX & 1) return;
        bt6(td, tr);                                      // tr[i'] = (B^T d)[i'][column sub6]
        p1_a = p1_base + next_byte; p1_b = p1_base + 4u * PX * 16u + next_byte;
    };
    // the LDS write path takes two 8-byte stores per MFMA gap for free and saturates beyond (docs/HISTORY.md A.3): the transform's
    // stores go out in pairs, one pair per gap
    auto tf_pass1w = [&](int k) __attribute__((always_inline)) {                // rows 2k, 2k+1 of the scratch
        if (MPQX & 1) return;
        if (MPQX & 32768) { asm volatile("" :: "v"(tr[2 * k]), "v"(tr[2 * k + 1])); return; }     // timing only: no store
        // lanes 6, 7 of a window sit out: three lanes storing to one address are a 3-way bank conflict on every store
        unsigned long long save;
        if (k == 0)
            asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\tds_write2_b64 %1, %2, %3 offset0:0 offset1:6\n\ts_mov_b64 exec, %0"
                         : "=&s"(save) : "v"(scr_w), "v"(tr[0]), "v"(tr[1]), "s"(0x3F3F3F3F3F3F3F3Full) : "memory");
        else if (k == 1)
            asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\tds_write2_b64 %1, %2, %3 offset0:12 offset1:18\n\ts_mov_b64 exec, %0"
                         : "=&s"(save) : "v"(scr_w), "v"(tr[2]), "v"(tr[3]), "s"(0x3F3F3F3F3F3F3F3Full) : "memory");
        else
            asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\tds_write2_b64 %1, %2, %3 offset0:24 offset1:30\n\ts_mov_b64 exec, %0"
                         : "=&s"(save) : "v"(scr_w), "v"(tr[4]), "v"(tr[5]), "s"(0x3F3F3F3F3F3F3F3Full) : "memory");
    };
    auto tf_pass2 = [&]() __attribute__((always_inline)) {
        if (MPQX & (1 | 16384)) return;
        if (MPQX & 65536) {
#pragma unroll
            for (int j = 0; j < 6; ++j) tq2[j] = *reinterpret_cast<const f32x2*>(&myscr[(sub6 * 6 + j) * 2]);
            return;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) td[j] = *reinterpret_cast<const f32x2*>(&myscr[(sub6 * 6 + j) * 2]);
    };
    // V[ch][tile][6 * row + j], j = 0..5, of the lane's two channels: ds_write2_b32 takes two unrelated registers, so the
    // packed results go out as they are (hipcc pairs them into ds_write_b64 through nine v_mov)
    const unsigned p2_addr = lds_addr(Vs) + (unsigned)p2_write * 4u;
    auto tf_pass2b = [&]() __attribute__((always_inline)) {
        if (MPQX & 1) return;
        bt6(td, tr);                                      // tr[j'] = V[row sub6][j']
    };
    auto tf_pass2w = [&](int buf, int k) __attribute__((always_inline)) {       // positions 2k, 2k+1 of both channels
        if (MPQX & 1) return;
        if (MPQX & 262144) { asm volatile("" :: "v"(tr[2 * k]), "v"(tr[2 * k + 1])); return; }    // timing only: no store
        const unsigned a0 = p2_addr + (unsigned)buf * (VB4 * 4u), a1 = a0 + 32u * 36u * 4u;
        const int j = 2 * k;
        unsigned long long save;
#define MPQ_VST(O0, O1)                                                                                                        \
        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %7\n\tds_write2_b32 %1, %2, %3 offset0:" #O0 " offset1:" #O1   \
                     "\n\tds_write2_b32 %4, %5, %6 offset0:" #O0 " offset1:" #O1 "\n\ts_mov_b64 exec, %0"                        \
                     : "=&s"(save) : "v"(a0), "v"(tr[j][0]), "v"(tr[j + 1][0]), "v"(a1), "v"(tr[j][1]), "v"(tr[j + 1][1]),       \
                       "s"(0x3F3F3F3F3F3F3F3Full) : "memory")
        if (k == 0) MPQ_VST(0, 1);
        else if (k == 1) MPQ_VST(2, 3);
        else MPQ_VST(4, 5);
#undef MPQ_VST
    };

    // ---- GEMM operands: a lane's fragments of 4 consecutive positions are one ds_read_b128 ----
    const int a_base = ((lane >> 4) * 64 + cb * 16 + (lane & 15)) * 36;    // U[ch = lane>>4][cout][pos]
    const int b_base = ((lane >> 4) * 32 + tb * 16 + (lane & 15)) * 36;    // V[ch = lane>>4][tile][pos]
    f32x4 af[3], bf[3];                                                     // rings over position groups of 4

    // ---- F1: the first encoder block, produced per unit into the raw ring ----
    // LDS: the raw ring needs only TWO buffers here (nothing is in flight from memory); the third buffer's 10 KiB hold two image
    // patches (item k's and item k+1's: 20 x 36 fp32, rows / columns already reflected) and the block's bias per unit (its BatchNorm is folded away at load time);
    // w1s holds the weights as [unit][tap][channel of the unit].
    constexpr int IW1 = PX + 2, IH1 = PY + 2;                 // image patch: the receptive field of the 18 x 34 raw patch
    constexpr int IPB = 768;                                  // floats per patch buffer (IH1 * IW1 = 720, rounded up to whole 64-lane DMA blocks)
    static_assert(!F1 || (IH1 * IW1 <= IPB && 2 * IPB + 16 * 4 + 16 * 4 * 12 <= RB4), "F1: patches + parameters + weights must fit the third raw buffer");
    const unsigned ip_lds = raw_lds + 2u * RB4 * 4u;          // ipatch[2][IPB]
    const unsigned bst_lds = ip_lds + 2u * IPB * 4u;          // bst[16][bias4] (the block's BatchNorm is folded away at load time)
    const unsigned w1_lds = bst_lds + 16u * 4u * 4u;         // w1u[16 units][4 channels][12: taps 0..8, 3 unused]
    typedef const __attribute__((address_space(3))) float* lds_f32_ptr;
    typedef const __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;
    typedef __attribute__((address_space(3))) f32x4* lds_f32x4_wptr;
    // this lane's pixels: granule f = (wave + 8 j) * 64 + lane (j = 1: waves 0, 1 only).  pg_x: absolute LDS byte address of the
    // pixel's window origin in the CURSOR item's patch buffer; pg_w: byte offset of the granule in a raw buffer (granules 612..639
    // of the last block land in the buffer's slack, never read)
    unsigned pg_x[2] = {0u, 0u}, pg_w[2] = {0u, 0u};
    bool pg_rel = false;
    int pc_par = 0;                                           // patch buffer of the cursor's item
    auto prod_offsets = [&](const Where& w) __attribute__((always_inline)) {
        const bool interior = (w.y0 >= 1) && (w.y0 + OY < p.H) && (w.x0 >= 1) && (w.x0 + OX < p.W);     // no reflection anywhere in the patch
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = (wave + 8 * j) * 64 + lane;
            const int fc = f < NPIX ? f : NPIX - 1;
            const int py = fc / PX, px = fc - py * PX;
            int oy = py, ox = px;
            if (!interior) {
                // the raw pixel is the block's output at the frame position reflected by THIS layer's padding; its window starts
                // one row / column before it, and staged row r holds image row reflect(y0 - 2 + r)
                oy = reflect_clamp_q(w.y0 + py - 1, p.H) - w.y0 + 1; ox = reflect_clamp_q(w.x0 + px - 1, p.W) - w.x0 + 1;
                oy = min(max(oy, 0), IH1 - 3); ox = min(max(ox, 0), IW1 - 3);      // (only pixels of phantom outputs are clamped)
            }
            pg_x[j] = ip_lds + (unsigned)(pc_par * IPB + oy * IW1 + ox) * 4u;
            pg_w[j] = (unsigned)f * 16u;
        }
        pg_rel = interior;
    };
    // image patch of item w -> patch buffer par: 720 pixels, one 4-byte DMA granule each (waves 0..3 issue two blocks)
    auto patch_dma = [&](const Where& w, int par) __attribute__((always_inline)) {
        const float* const im = p.img + (long long)w.img * p.H * p.W;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (wave * 64 + 512 * j >= IH1 * IW1) continue;       // (wave-uniform)
            const int q = min(tid + 512 * j, IH1 * IW1 - 1);
            const int r = q / IW1, c = q - r * IW1;
            const unsigned off = (unsigned)(reflect_clamp_q(w.y0 - 2 + r, p.H) * p.W + reflect_clamp_q(w.x0 - 2 + c, p.W)) * 4u;
            dma4(im, off, ip_lds + (unsigned)(par * IPB + wave * 64 + 512 * j) * 4u);
        }
    };
    // the cursor: unit pc_unit of item pc_item is what the next produce() makes; LDS byte addresses of its weights / parameters
    int pc_unit = 0, pc_next_item = item + stride;
    unsigned pc_wv = w1_lds + (unsigned)(lane & 3) * 48u;     // w1u[unit][lane & 3][0..11]
    unsigned pc_bv = bst_lds;                                 // bst[unit * 4]
    // 64 pixels x 4 channels of the cursor's unit -> raw buffer at byte wbuf: nine rank-1 MFMAs on the lane's own pixel.  In steps, so
    // that the unit body can thread them through the gaps of its own MFMA stream (the nine small MFMAs depend on each other):
    //   step 0: issue the reads (9 taps of the pixel, 9 weights, the bias as the accumulator's initial value)
    //   steps 1..9: one v_mfma_f32_4x4x1_16b_f32 each; step 8 also fetches the BatchNorm terms
    //   step 10: ReLU / BatchNorm and the 16-byte store
    // The nine tap values of a pixel are the same for all 16 units of an item: block 0's (every wave) stay in registers (px_, loaded
    // when the cursor enters an item); the second block of waves 0, 1 gathers them per unit.
    float px_[9];
    f32x4 pw_[3], pd_;
    auto gather_x = [&](float (&x)[9], const int j) __attribute__((always_inline)) {
        const lds_f32_ptr xp = reinterpret_cast<lds_f32_ptr>(pg_x[j]);
#pragma unroll
        for (int t = 0; t < 9; ++t) x[t] = (MPQX & 16777216) ? 1.f : xp[(t / 3) * IW1 + (t % 3)];
    };
    auto prod_step = [&](const int k, const unsigned wbuf, const int j, const float (&x)[9]) __attribute__((always_inline)) {
        const lds_f32x4_ptr wp = reinterpret_cast<lds_f32x4_ptr>(pc_wv);
        const lds_f32x4_ptr bp = reinterpret_cast<lds_f32x4_ptr>(pc_bv);
        if (k == 0) {
            pd_ = bp[0];
            pw_[0] = wp[0]; pw_[1] = wp[1]; pw_[2] = wp[2];
        } else if (k <= 9) {
            if (MPQX & 8388608) asm volatile("" :: "v"(pw_[(k - 1) >> 2][(k - 1) & 3]), "v"(x[k - 1]));       // (timing only: no small MFMAs)
            else pd_ = __builtin_amdgcn_mfma_f32_4x4x1f32(pw_[(k - 1) >> 2][(k - 1) & 3], x[k - 1], pd_, 0, 0, 0);
        } else {
            // ReLU only: the block's BatchNorm was folded at load time -- into p.w1 / p.b1 for conv -> BN -> ReLU models, into this
            // layer's U and bias otherwise (api.hip build_encoder) -- which takes two packed multiply-adds and two LDS reads per
            // 64 pixels and unit out of the unit body
            const f32x4 v = {relu_q(pd_[0]), relu_q(pd_[1]), relu_q(pd_[2]), relu_q(pd_[3])};
            *reinterpret_cast<lds_f32x4_wptr>(raw_lds + wbuf + pg_w[j]) = v;
        }
    };
    auto produce = [&](const unsigned wbuf, const int j) __attribute__((always_inline)) {       // all steps at once
        if (j == 0) {
#pragma unroll
            for (int k = 0; k <= 10; ++k) prod_step(k, wbuf, 0, px_);
        } else {
            float qx[9];
            gather_x(qx, 1);
#pragma unroll
            for (int k = 0; k <= 10; ++k) prod_step(k, wbuf, 1, qx);
        }
    };
    auto prod_advance = [&]() __attribute__((always_inline)) {
        pc_wv += 4u * 12u * 4u; pc_bv += 4u * 4u;
        if (++pc_unit == NC) {
            pc_unit = 0;
            pc_wv -= (unsigned)NC * 4u * 12u * 4u; pc_bv -= (unsigned)NC * 4u * 4u;
            if (pc_next_item < item_end) {
                const Where w = decode(pc_next_item);
                pc_par ^= 1;
                if (!(pg_rel && (w.y0 >= 1) && (w.y0 + OY < p.H) && (w.x0 >= 1) && (w.x0 + OX < p.W))) prod_offsets(w);
                else {                                            // interior -> interior: only the patch buffer changes
                    const unsigned flip = pc_par ? (unsigned)IPB * 4u : 0u - (unsigned)IPB * 4u;
                    pg_x[0] += flip; pg_x[1] += flip;
                }
                gather_x(px_, 0);                                 // (the patch landed and was fenced by a unit barrier long ago)
                pc_next_item += stride;
            }
        }
    };

    // ---- VIN: the pre-transformed input of unit n + 3 by DMA into V ring slot (n + 3) & 3 ----
    // 18 blocks of 1 KiB per unit: wave w issues blocks w, w + 8, w + 16; waves 2..7 have no third block -- theirs re-reads block 17
    // into the dummy block (every wave issues three, so that one s_waitcnt vmcnt(3) leaves exactly a unit's V DMAs in flight)
    const unsigned vs_lds = lds_addr(Vs);
    auto v_dma = [&](const float* vsrc, int slot, int i) __attribute__((always_inline)) {
        const int b = wave + 8 * i;
        const unsigned dst = b < 18 ? vs_lds + (unsigned)(slot * VB4 + b * 256) * 4u : lds_addr(raw);
        dma16(vsrc + (b < 18 ? b : 17) * 256, (unsigned)lane * 16u, dst);
    };
    const float* vcur = nullptr;              // V of the cursor's unit in p.vglobal
    int v_chunk = 0, v_next_item = item + stride;
    auto v_advance = [&]() __attribute__((always_inline)) {
        vcur += VB4;
        if (++v_chunk == NC) {
            v_chunk = 0;
            if (v_next_item < item_end) {
                vcur = p.vglobal + (long long)decode(v_next_item).tile * NC * VB4;
                v_next_item += stride;
            } else {
                vcur -= (long long)NC * VB4;      // no next item: dummy re-reads of this one
            }
        }
    };

    // ---- prologue ----
    Where cur = decode(item);
    if constexpr (F1) {
        for (int f = tid; f < 16 * 4 * 12; f += 512) {            // w1u[unit][channel of the unit][tap] <- p.w1 [tap][64]
            const int ch = f / 12, t = f - ch * 12;
            (raw + 2 * RB4 + 2 * IPB + 16 * 4)[f] = t < 9 ? p.w1[t * 64 + ch] : 0.f;
        }
        if (tid < 64) (raw + 2 * RB4 + 2 * IPB)[tid] = p.b1[tid];       // bst[unit][4]: the bias = the accumulators' initial value
        patch_dma(cur, 0);
        if (item + stride < item_end) patch_dma(decode(item + stride), 1);
        prod_offsets(cur);
        dma_wait();
        __syncthreads();                                          // patches, weights and parameters visible
        gather_x(px_, 0);
        produce(0u, 0); if (wave + 8 < NRB) produce(0u, 1);       // raw(0) -> buffer 0
        prod_advance();
        produce(RB4 * 4u, 0); if (wave + 8 < NRB) produce(RB4 * 4u, 1);       // raw(1) -> buffer 1
        prod_advance();
    }
    const float* rbase = (F1 || VIN) ? p.in : raw_offsets(cur);
    const float* rsrc = rbase;               // the cursor's unit: rbase + ld_chunk * unit_stride
    Where ld_item = cur;
    int ld_chunk = 0;
    int ld_next_item = item + stride;
    auto ld_advance = [&]() __attribute__((always_inline)) {
        rsrc += unit_stride;
        if (++ld_chunk == NC) {
            ld_chunk = 0;
            if (ld_next_item < item_end) {
                ld_item = decode(ld_next_item);
                rbase = raw_offsets(ld_item);
                ld_next_item += stride;
            }
            rsrc = rbase;
        }
    };
    const float* up = u_ptr(cur.slice);
    // raw(k) lives in raw buffer k % 3: unit n transforms raw(n+1) and sends raw(n+3) over raw(n)
    if constexpr (VIN) {
        vcur = p.vglobal + (long long)cur.tile * NC * VB4;
#pragma unroll
        for (int u = 0; u < 3; ++u) {                                                                   // V(0), V(1), V(2)
            v_dma(vcur, u, 0); v_dma(vcur, u, 1); v_dma(vcur, u, 2);
            v_advance();
        }
    } else if constexpr (!F1) {
        raw_dma(rsrc, 0u, 0); raw_dma(rsrc, 0u, 1); ld_advance();                                       // raw(0)
        raw_dma(rsrc, RB4 * 4u, 0); raw_dma(rsrc, RB4 * 4u, 1); ld_advance();                           // raw(1)
        raw_dma(rsrc, 2u * RB4 * 4u, 0); raw_dma(rsrc, 2u * RB4 * 4u, 1); ld_advance();                 // raw(2)
    }
    unsigned rd_byte = 0u, rt_byte = RB4 * 4u;           // byte offsets of the raw buffer unit n DMAs into / transforms from
    u_dma_site(up, 0, 0); u_dma_site(up, 0, 1);                                        // U(0)
    u_dma_site(up + UB4, 1, 0); u_dma_site(up + UB4, 1, 1);                            // U(1)
    load_prm(cur.slice);
    dma_wait();
    __syncthreads();
    if constexpr (!VIN) {
        tf_pass1(); tf_pass1b(RB4 * 4u); tf_pass1w(0); tf_pass1w(1); tf_pass1w(2); tf_pass2(); tf_pass2b();
        tf_pass2w(0, 0); tf_pass2w(0, 1); tf_pass2w(0, 2);                            // V(0); unit 0 transforms raw(1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the V stores are asm statements: hipcc does not count them
        __syncthreads();
    }
    af[0] = *reinterpret_cast<const f32x4*>(&Us[a_base]);
    bf[0] = *reinterpret_cast<const f32x4*>(&Vs[b_base]);
    af[1] = *reinterpret_cast<const f32x4*>(&Us[a_base + 4]);
    bf[1] = *reinterpret_cast<const f32x4*>(&Vs[b_base + 4]);

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int cur_par = 0;                  // F1: patch buffer of the current item
#ifdef MP_TIMING
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool t_on = ((POOL ? p.H : -p.H) == g_timing_q_sel);
#endif
    for (;;) {
        MPQ_T(t_item);
        f32x4 acc[36];
        const int item_next = item + stride;
        const bool has_next = item_next < item_end;
        const int next_slice = has_next ? (int)(item_next - (int)udiv((unsigned)item_next, p.magic_slices, (unsigned)p.nslices) * p.nslices)
                                        : cur.slice;
        const float* unext = u_ptr(next_slice);
        // schedule of the input transform inside a unit (group g, slot e behind the e-th MFMA of the group)
        auto tf_at = [&](const int g, const int e, const int vb) __attribute__((always_inline)) {
#ifndef MP_TF_SCHED
#define MP_TF_SCHED -1
#endif
            // slots (group, MFMA of the group) of: raw reads | column pass + hand-over stores 0 | 1 | 2 | scratch reads | row pass + V stores 0 | 1 | 2.
            // Row 0: the schedule of rounds 3-5, kept for the fused launch (its production steps fill the neighbouring slots) and the
            // 32 x 16-pixel items.  Row 1 (round 6): the row pass one slot earlier and its stores one per group instead of back to back --
            // conv4 -2.3 %, conv6 -1.6 %, conv3 -0.8 % on one box (eight schedules tried: profiles/r06_transform_schedules.txt); the
            // same schedule costs conv7 / conv8 1 % and leaves the fused launch where it is.
            constexpr int S[2][8][2] = {
                {{0, 2}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {5, 2}, {5, 3}, {6, 1}},
                {{0, 2}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {4, 3}, {5, 1}, {6, 1}}};
            constexpr int TFS = MP_TF_SCHED >= 0 ? MP_TF_SCHED : ((TC4 == 8 && !F1) ? 1 : 0);
            auto at = [&](const int k) { return g == S[TFS][k][0] && e == S[TFS][k][1]; };
            if (at(0)) tf_pass1();
            if (at(1)) { tf_pass1b(F1 ? (unsigned)vb * (RB4 * 4u) : 3u * RB4 * 4u - rd_byte - rt_byte); tf_pass1w(0); }   // unit n+1 transforms raw(n+2): ring of 3 (F1: of 2, buffer n & 1)
            if (at(2)) tf_pass1w(1);
            if (at(3)) tf_pass1w(2);
            if (at(4)) tf_pass2();
            if (at(5)) { tf_pass2b(); tf_pass2w(vb ^ 1, 0); }
            if (at(6)) tf_pass2w(vb ^ 1, 1);
            if (at(7)) tf_pass2w(vb ^ 1, 2);
        };
        // F1: the production of raw(n+2) (prod_step) in the slots the transform leaves free, one small MFMA per slot; the second
        // block of waves 0, 1 (granules 512..611) as one piece in front of them
        auto prod_at = [&](const int g, const int e, const int vb) __attribute__((always_inline)) {
            const unsigned wbuf = (unsigned)vb * (RB4 * 4u);
            // (round 6: the second block at group 1 / 3 / 5 instead of 0: +-0 / +1.7 % / +1.6 %; one production step per group or all of them
            // early: +-0 / +0.9 % -- profiles/r06_transform_schedules.txt)
            constexpr int slot_g[11] = {0, 0, 1, 1, 1, 2, 3, 4, 4, 4, 5};
            constexpr int slot_e[11] = {1, 3, 1, 2, 3, 1, 3, 1, 2, 3, 1};
            if (MPQX & 4194304) return;                            // (timing only: no production at all)
            if (g == 0 && e == 0 && !(MPQX & 2097152)) { if (wave + 8 < NRB) produce(wbuf, 1); }      // (2097152, timing only: not the second block)
#pragma unroll
            for (int k = 0; k <= 10; ++k)
                if (g == slot_g[k] && e == slot_e[k]) prod_step(k, wbuf, 0, px_);
        };
        // the 36 MFMAs of a unit and everything that rides in their shadow: one basic block
        auto unit_body = [&](const int c, auto first_tag, auto vb_tag, auto vs_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr int vb = decltype(vb_tag)::value;
            constexpr int vs = VIN ? decltype(vs_tag)::value : vb;        // V buffer of this unit (VIN: ring of four)
            constexpr int vsn = VIN ? (vs + 1) & 3 : vb ^ 1;             // ... of the next one
            const float* const ur = Us + vb * UB4 + a_base;
            const float* const vr = Vs + vs * VB4 + b_base;
            const float* const urn = Us + (vb ^ 1) * UB4 + a_base;
            const float* const vrn = Vs + vsn * VB4 + b_base;
            const float* const un2 = c + 2 < NC ? up + (long long)(c + 2) * UB4 : unext + (long long)(c + 2 - NC) * UB4;
            MPQ_T(t_u0);
#ifdef MP_TIMING
            unsigned long long t_b0 = 0, t_b1 = 0;
#endif
#pragma unroll
            for (int g = 0; g < 9; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int s = 4 * g + e;
                    // weights as the A operand: D[cout][tile] -- lane = tile, register quad = 4 consecutive output channels
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g % 3][e], bf[g % 3][e], FIRST ? zero4 : acc[s], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!VIN) { if (e != 2) tf_at(g, e, vb); }
                    if constexpr (F1) prod_at(g, e, vb);
                    if (e == 0 && !(MPQX & 4)) {
                        // fragments two groups ahead; groups 7, 8 fetch groups 0, 1 of the NEXT unit (behind the unit barrier)
                        af[(g + 2) % 3] = *reinterpret_cast<const f32x4*>(g + 2 < 9 ? &ur[4 * (g + 2)] : &urn[4 * (g - 7)]);
                        bf[(g + 2) % 3] = *reinterpret_cast<const f32x4*>(g + 2 < 9 ? &vr[4 * (g + 2)] : &vrn[4 * (g - 7)]);
                    } else if (e == 1) {
                        // The memory pipe returns in order across the whole CU: a weight DMA (L2 hit) queued behind a patch
                        // DMA (HBM miss) of ANY wave comes back at HBM latency.  So the two kinds are kept apart in time:
                        // U(n+2) -> U[vb] right behind this unit's barrier (groups 7, 8: every wave has fetched its last
                        // fragment of U[vb] by then), raw(n+3) in groups 0, 1 of the next body -- six groups of MFMAs
                        // before the next weights queue up behind it.  The barrier waits with vmcnt(2): for everything
                        // but the two patch DMAs, which have until the NEXT barrier.
                        if (g == 7) u_dma_site(un2, vb, 0);
                        else if (g == 8) u_dma_site(un2, vb, 1);
                        else if (g == 0) { if constexpr (VIN) v_dma(vcur, (vs + 3) & 3, 0); else if constexpr (!F1) raw_dma(rsrc, rd_byte, 0); }
                        else if (g == 1) { if constexpr (VIN) v_dma(vcur, (vs + 3) & 3, 1); else if constexpr (!F1) raw_dma(rsrc, rd_byte, 1); }
                        else if (g == 2) { if constexpr (VIN) v_dma(vcur, (vs + 3) & 3, 2); }
                    } else if (e == 2) {
                        // input transform of unit n+1: raw[vb^1] -> V[vb^1]
                        // input transform of unit n+1, raw -> V[vb^1]: reads, arithmetic and stores of the two passes spread
                        // over groups 0-6 (the schedule table is tf_at())
                        if constexpr (!VIN) tf_at(g, 2, vb);
                    } else {
                        if (g == 6) {
                            // unit barrier: every fragment of the unit has been fetched (two groups ahead), V(n+1) is
                            // written, the DMAs of the unit were issued before group 7
#ifdef MP_TIMING
                            t_b0 = __builtin_amdgcn_s_memtime();
#endif
                            // U(n+1), raw(n+2) (and an item's output stores) have landed; raw(n+3) stays in flight
                            // (lgkmcnt: the V stores are asm statements hipcc does not count)
                            if constexpr (F1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (no raw DMA in flight: raw(n+2) was produced above)
                            else if constexpr (VIN) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");   // U(n+1), V(n+2) landed; V(n+3) stays in flight
                            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                            if (MPQX & 65536) {
#pragma unroll
                                for (int q = 0; q < 6; ++q) asm volatile("" :: "v"(tq[q]), "v"(tq2[q]));
                            }
#ifdef MP_TIMING
                            t_b1 = __builtin_amdgcn_s_memtime();
#endif
                            __syncthreads();
#ifdef MP_TIMING
                            tsum[4] += __builtin_amdgcn_s_memtime() - t_b1;
#endif
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            MPQ_T(t_u1);
            MPQ_ADD(0, t_u0, t_u1);                                   // a unit incl. its barrier
            MPQ_ADD(1, t_b0, t_b1);                                   // the DMA wait alone (slot 4: the barrier behind it)
        };
        auto unit = [&](const int c, auto first_tag, auto vb_tag, auto vs_tag) __attribute__((always_inline)) {
            unit_body(c, first_tag, vb_tag, vs_tag);
            if constexpr (VIN) {
                v_advance();
            } else if constexpr (F1) {
                prod_advance();
            } else {
                ld_advance();
                const unsigned ro = 3u * RB4 * 4u - rd_byte - rt_byte;     // rotate the raw ring: (rd, rt) <- (rt, third)
                rd_byte = rt_byte; rt_byte = ro;
            }
        };
        using VB0 = std::integral_constant<int, 0>;
        using VB1 = std::integral_constant<int, 1>;
        using VS2 = std::integral_constant<int, 2>;
        using VS3 = std::integral_constant<int, 3>;
        if constexpr (VIN) {          // units per item a multiple of 4 (cin % 16 == 0): unit c lives in U[c & 1], V[c & 3]
            unit(0, std::true_type{}, VB0{}, VB0{});
            for (int c = 1; c + 3 < NC; c += 4) {
                unit(c, std::false_type{}, VB1{}, VB1{});
                unit(c + 1, std::false_type{}, VB0{}, VS2{});
                unit(c + 2, std::false_type{}, VB1{}, VS3{});
                unit(c + 3, std::false_type{}, VB0{}, VB0{});
            }
            unit(NC - 3, std::false_type{}, VB1{}, VB1{});
            unit(NC - 2, std::false_type{}, VB0{}, VS2{});
            unit(NC - 1, std::false_type{}, VB1{}, VS3{});
        } else {
        unit(0, std::true_type{}, VB0{}, VB0{});
        for (int c = 1; c + 1 < NC; c += 2) {
            unit(c, std::false_type{}, VB1{}, VB1{});
            unit(c + 1, std::false_type{}, VB0{}, VB0{});
        }
        unit(NC - 1, std::false_type{}, VB1{}, VB1{});
        }

        MPQ_T(t_e0);
        MPQ_ADD(3, t_item, t_e0);                                      // whole unit loop of the item
        // F1: the image patch of item k+2 -> this item's patch buffer (last read while unit 13 was multiplied); in flight across the epilogue
        if constexpr (F1) { if (item_next + stride < item_end) patch_dma(decode(item_next + stride), cur_par); }
        // ---- output transform Y = A^T M A in registers, bias / ReLU / BN, [2x2 max-pool], store ----
        // lane = tile (lane & 15) of the wave's tile block, registers r = output channels 4 * (lane >> 4) + r of its channel block
        if (MPQX & 8) {
            float sink = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 36; ++s2) sink += acc[s2][0] + acc[s2][1] + acc[s2][2] + acc[s2][3];
            if (sink == 123.456f) p.out[tid] = sink;
        } else {
            const int tl = tb * 16 + (lane & 15);                       // tile of the item: row tl / TC4, column tl % TC4
            const int cl = cb * 16 + 4 * (lane >> 4);                   // first of this lane's 4 output channels in the slice
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(&prm[cl]);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(&prm[64 + cl]);
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(&prm[128 + cl]);
            const int oy = cur.y0 + 4 * (tl / TC4), ox = cur.x0 + 4 * (tl % TC4);
            const int ch0 = (SPLIT ? cur.slice >> p.ks_shift : cur.slice) * 64 + cl;
            const int cs = p.out_cstride;
            // per channel pair h (registers 2h, 2h+1): transform, activation, [pool] -- the 72 accumulator registers of a
            // finished pair are dead before the next one starts (register budget: 256 per lane); the first pair's results
            // wait in registers so that every pixel is ONE 16-byte store of the lane's 4 channels
            constexpr int NO = POOL ? 2 : 4;                            // output rows / columns per tile
            f32x2 keep[NO][NO];
            // SPLIT: this item's share of the sum over input channels leaves as pre-bias output tiles; split_reduce_kernel (conv_split.hip),
            // the next launch on the stream, adds the ranges' shares up in range order (deterministic) and runs the rest of the epilogue
            if constexpr (SPLIT) {
                unsigned long long* const part = reinterpret_cast<unsigned long long*>(p.split_scratch) + (long long)item * (2 * 16 * 512) + tid;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 tcol[4][6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        f32x2 m[6], y[4];
#pragma unroll
                        for (int i = 0; i < 6; ++i) m[i] = f32x2{acc[6 * i + j][2 * h], acc[6 * i + j][2 * h + 1]};
                        at6s(m, y);
#pragma unroll
                        for (int a = 0; a < 4; ++a) tcol[a][j] = y[a];
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        f32x2 y[4];
                        at6s(tcol[a], y);
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            part[(h * 16 + a * 4 + b) * 512] = __builtin_bit_cast(unsigned long long, y[b]);
                    }
                }
            }
            if constexpr (!SPLIT) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 tcol[4][6];                                       // T[a][j] = sum_i A^T[a][i] M[i][j]
                if constexpr (!SPLIT) {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x2 m[6], y[4];
#pragma unroll
                    for (int i = 0; i < 6; ++i) m[i] = f32x2{acc[6 * i + j][2 * h], acc[6 * i + j][2 * h + 1]};
                    at6s(m, y);
#pragma unroll
                    for (int a = 0; a < 4; ++a) tcol[a][j] = y[a];
                }
                }
                const f32x2 bb = {b4[2 * h], b4[2 * h + 1]}, ss = {s4[2 * h], s4[2 * h + 1]}, tt = {t4[2 * h], t4[2 * h + 1]};
                f32x2 res[NO][NO];
                if constexpr (POOL) {
                    // Pool BEFORE the activation: one activation per pooled value instead of four (-17 % of the epilogue's vector
                    // instructions, and next to the fp32 MFMA stream every vector instruction is matrix-pipe time).  Bit for bit the
                    // same result: bias add, ReLU and the BatchNorm affine are monotonic per channel -- non-decreasing where the
                    // scale is >= 0, non-increasing where it is negative -- so max(f(x_i)) = f(max x_i), or f(min x_i) for a negative
                    // scale.  The sign g = +-1 of the channel's scale is folded into the multiply-add that scales and adds the bias
                    // anyway (x' = g x exactly: an fma is odd in its product and addend), one max-pool serves both signs
                    // (max g x = g * max / min x), and the activation of the sign-folded value needs no multiplication back:
                    //   conv -> ReLU -> BN:  s relu(x) + t = |s| clamp(x') + t,  clamp = max(x', 0) for g = 1, min(x', 0) for g = -1
                    //                        (one v_med3_i32 on the float bits, as the ReLU is one v_max_i32)
                    //   conv -> BN -> ReLU:  relu(s x + t) = relu(|s| x' + t)
                    // (tests/test_gpu_parity.py::test_pooled_epilogue_pools_before_the_activation compares against the direct kernels.)
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(&prm[192 + cl]);
                    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(&prm[256 + cl]), hi4 = *reinterpret_cast<const f32x4*>(&prm[320 + cl]);
                    const f32x2 gg = {g4[2 * h], g4[2 * h + 1]};
                    const f32x2 gb = bb * gg;                                                  // g * bias (exact)
                    constexpr float A1 = (float)MP_W43_A, A2 = A1 * A1, A3 = A2 * A1, A4 = A2 * A2;
                    const f32x2 gk[5] = {gg, gg * f32x2{A1, A1}, gg * f32x2{A2, A2}, gg * f32x2{A3, A3}, gg * f32x2{A4, A4}};      // g * sigma_r sigma_c
                    f32x2 xv[4][4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        f32x2 y[4];
                        at6s(tcol[a], y);
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int e = (a == 1 ? 1 : a == 2 ? 2 : 0) + (b == 1 ? 1 : b == 2 ? 2 : 0);      // w43_out_scale(a, b) = A1^e
                            xv[a][b] = __builtin_elementwise_fma(y[b], gk[e], gb);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < NO; ++a)
#pragma unroll
                        for (int b = 0; b < NO; ++b) {
                            f32x2 m;
#pragma unroll
                            for (int r = 0; r < 2; ++r)
                                m[r] = fmaxf(fmaxf(xv[2 * a][2 * b][r], xv[2 * a][2 * b + 1][r]), fmaxf(xv[2 * a + 1][2 * b][r], xv[2 * a + 1][2 * b + 1][r]));
                            if (BNF) {
                                m = __builtin_elementwise_fma(m, ss, tt);
                                m = f32x2{relu_q(m[0]), relu_q(m[1])};
                            } else {
#pragma unroll
                                for (int r = 0; r < 2; ++r)
                                    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(m[r]) : "v"(m[r]), "v"(lo4[2 * h + r]), "v"(hi4[2 * h + r]));
                                m = __builtin_elementwise_fma(m, ss, tt);
                            }
                            res[a][b] = m;
                        }
                } else {
                f32x2 yv[4][4];                                         // [row a][col b] -> this pair's 2 channels
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f32x2 y[4];
                    at6s(tcol[a], y);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        f32x2 v = w43_add_bias(y[b], a, b, bb);
                        if (BNF) { v = v * ss + tt; v = f32x2{relu_q(v[0]), relu_q(v[1])}; }
                        else { v = f32x2{relu_q(v[0]), relu_q(v[1])}; v = v * ss + tt; }
                        yv[a][b] = v;
                    }
                }
#pragma unroll
                for (int a = 0; a < NO; ++a)
#pragma unroll
                    for (int b = 0; b < NO; ++b) res[a][b] = yv[a][b];
                }
                if (h == 0) {
#pragma unroll
                    for (int a = 0; a < NO; ++a)
#pragma unroll
                        for (int b = 0; b < NO; ++b) keep[a][b] = res[a][b];
                } else {
                    const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
                    const int py0 = POOL ? oy >> 1 : oy, px0 = POOL ? ox >> 1 : ox;
                    // a uniform per-image base + 32-bit byte offsets (an image's output is far below 4 GB).  NHWC: pixel stride
                    // cs floats; planar [B][cout/4][Ho][Wo][4]: this lane's quad is plane ch0 / 4, pixel stride 16 bytes
                    char* const img_base = reinterpret_cast<char*>(
                        p.out_planar ? p.out + (long long)cur.img * (p.cout / 4) * Ho * Wo * 4
                                     : p.out + (long long)cur.img * Ho * Wo * cs + p.out_coff);
                    const unsigned ps = p.out_planar ? 16u : (unsigned)cs * 4u;                       // bytes per pixel step
                    const unsigned rs = (unsigned)Wo * ps;                                            // bytes per row step
                    const unsigned o0 = p.out_planar ? (unsigned)(((ch0 >> 2) * Ho + py0) * Wo + px0) * 16u
                                                     : (unsigned)((py0 * Wo + px0) * cs + ch0) * 4u;
                    // H and W are multiples of 4 (conv_wino43_supports) and tiles are 4-aligned: a tile is inside the image or
                    // outside as a whole, and cout is a multiple of 4 -- ONE test, then NO x NO unconditional 16-byte stores
                    if (oy < p.H && ox < p.W && ch0 < p.cout) {
#pragma unroll
                        for (int a = 0; a < NO; ++a)
#pragma unroll
                            for (int b = 0; b < NO; ++b) {
                                const f32x4 v = {keep[a][b][0], keep[a][b][1], res[a][b][0], res[a][b][1]};
                                // (non-temporal stores, measured: un-pooled NHWC outputs +10 % -- the 64-byte pieces of a pixel's line written
                                // by four waves no longer merge in L2 --, pooled planar ones -1 % for the launch and +1 % for its consumer)
                                *reinterpret_cast<f32x4*>(img_base + (o0 + (unsigned)a * rs + (unsigned)b * ps)) = v;
                            }
                    }
                }
            }
            }
        }
        MPQ_T(t_e1);
        MPQ_ADD(2, t_e0, t_e1);                                        // epilogue
#ifdef MP_TIMING
        tsum[7] += 1;
        if (!has_next && tid == 0 && t_on)
            for (int i = 0; i < 8; ++i) g_timing_q[blockIdx.x * 8 + i] = tsum[i];
#endif
        if (!has_next) { dma_wait(); return; }      // the prefetch DMAs still in flight write THIS workgroup's LDS: drain them
        if (next_slice != cur.slice) {
            __syncthreads();
            load_prm(next_slice);
        }
        item = item_next;
        cur = decode(item);
        cur_par ^= 1;
        up = unext;
    }
}

// The input transform as a pass of its own (VIN launches): V = B^T d B of every 6x6 window of the layer's NHWC input, written to
// p.vglobal as [tile block][unit = cin / 4][channel of the unit 4][tile 32][pos 36] -- per (tile block, unit) exactly the 18 KiB the
// VIN kernel DMAs into a V buffer.  One thread per (tile, channel pair): the column pass and the row pass are the SAME bt6() chains
// the in-kernel transform runs (column pass first), so V has the same bits.  HBM/L2 streaming: reads 2.25x the input (the windows
// of adjacent tiles overlap) coalesced over the channel pairs of a pixel, writes 2.25x the input as 144-byte runs.
template <int TC4>
__global__ __launch_bounds__(256) void wino43_vprod_kernel(const ConvParams p, long long ntb)
{
    // A block = TL consecutive tiles of a tile block x every channel pair (256 threads = TL x cin / 2).  The reads stay coalesced over
    // the channel pairs of a pixel; the results are transposed through LDS so that they leave as 16-byte pieces of TL x 144-byte runs
    // (the first version stored each lane's 144-byte run directly: 64 pieces 9 KiB apart per store instruction, 0.18 ms).
    constexpr int TR4 = 32 / TC4, OY = 4 * TR4, OX = 4 * TC4;
    __shared__ __attribute__((aligned(16))) float st[19456];      // cin x (TL x 36 + 4) floats: 18432 + 4 cin, cin <= 256
    const int ncp = p.cin >> 1, NC = p.cin / UC4;
    const int TL = 256 / ncp;                            // tiles per block (4 for cin = 128)
    const int RS = TL * 36 + 4;                          // floats per (unit, channel) row: padded, lanes 2 rows apart hit 4 bank quads
    const int groups = 32 / TL;
    const int tid = threadIdx.x;
    const int cp = tid % ncp, tlc = tid / ncp;
    const long long tb = (long long)blockIdx.x / groups;
    const int t0 = ((int)((long long)blockIdx.x % groups)) * TL, t = t0 + tlc;
    (void)ntb;
    const int tx = (int)(tb % p.tiles_x);
    const long long trow = tb / p.tiles_x;
    const int ty = (int)(trow % p.tiles_y);
    const int bi = (int)(trow / p.tiles_y);
    const int img = p.img_list ? p.img_list[bi] : bi;
    const int wy = ty * OY + 4 * (t / TC4) - 1, wx = tx * OX + 4 * (t % TC4) - 1;      // frame position of the window's corner
    const float* const base = p.in + (long long)img * p.H * p.W * p.in_cstride + p.in_coff + 2 * cp;
    int gx[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) gx[j] = reflect_clamp_q(wx + j, p.W);
    f32x2 r[6][6];                                       // r[i'][j] = (B^T d)[i'][column j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x2 d[6], o[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
            d[i] = *reinterpret_cast<const f32x2*>(base + ((long long)reflect_clamp_q(wy + i, p.H) * p.W + gx[j]) * p.in_cstride);
        bt6(d, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = o[i];
    }
    float* const dst0 = st + (2 * cp) * RS + tlc * 36;   // row = unit * 4 + channel of the unit = 2 cp (+ 1)
    float* const dst1 = dst0 + RS;
#pragma unroll
    for (int i = 0; i < 6; i += 2) {                     // two rows = 12 positions = three 16-byte stores per channel
        f32x2 a[6], b[6];
        bt6(r[i], a);
        bt6(r[i + 1], b);
        *reinterpret_cast<f32x4*>(dst0 + 6 * i) = f32x4{a[0][0], a[1][0], a[2][0], a[3][0]};
        *reinterpret_cast<f32x4*>(dst0 + 6 * i + 4) = f32x4{a[4][0], a[5][0], b[0][0], b[1][0]};
        *reinterpret_cast<f32x4*>(dst0 + 6 * i + 8) = f32x4{b[2][0], b[3][0], b[4][0], b[5][0]};
        *reinterpret_cast<f32x4*>(dst1 + 6 * i) = f32x4{a[0][1], a[1][1], a[2][1], a[3][1]};
        *reinterpret_cast<f32x4*>(dst1 + 6 * i + 4) = f32x4{a[4][1], a[5][1], b[0][1], b[1][1]};
        *reinterpret_cast<f32x4*>(dst1 + 6 * i + 8) = f32x4{b[2][1], b[3][1], b[4][1], b[5][1]};
    }
    __syncthreads();
    // [unit][ch 4][tile 32][pos 36] in p.vglobal: row `row` = (unit, ch) holds this block's TL tiles as one run of TL x 144 bytes
    const int q4 = TL * 9, total4 = p.cin * q4;
    float* const gbase = p.vglobal + (tb * NC * 4 * 32 + t0) * 36;
    for (int i = tid; i < total4; i += 256) {
        const int row = i / q4, w = i - row * q4;
        *reinterpret_cast<f32x4*>(gbase + (long long)row * (32 * 36) + w * 4) = *reinterpret_cast<const f32x4*>(st + row * RS + w * 4);
    }
}

template <bool POOL, int TC4, bool F1 = false, bool SPLIT = false>
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
    if (p.bn_first) hipLaunchKernelGGL((conv_wino43_kernel<POOL, true, TC4, F1, SPLIT>), dim3(grid), dim3(512), 0, s, pp);
    else hipLaunchKernelGGL((conv_wino43_kernel<POOL, false, TC4, F1, SPLIT>), dim3(grid), dim3(512), 0, s, pp);
    return 0;
}

// two passes: the input transform into p.vglobal, then the GEMMs + output transform per (tile block, slice) with V by DMA
template <int TC4>
int launch_vin(const ConvParams& p, hipStream_t s)
{
    constexpr int OY = 4 * (32 / TC4), OX = 4 * TC4;
    ConvParams q = p;
    q.tiles_x = (p.W + OX - 1) / OX; q.tiles_y = (p.H + OY - 1) / OY;
    const long long ntb = (long long)p.B * q.tiles_x * q.tiles_y, nitems = ntb * q.nslices;
    if (nitems <= 0) return 0;
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); };
    q.magic_slices = magic(q.nslices); q.magic_tx = magic(q.tiles_x); q.magic_ty = magic(q.tiles_y);
    const long long dmax = std::max(std::max(q.nslices, q.tiles_x), q.tiles_y);
    if (nitems * dmax >= 0x100000000ll) return 1;
    q.nitems = (int)nitems;
    const ConvParams& pp = q;
    const int tl = 256 / (p.cin / 2);                    // tiles per producer block (launch_conv_wino43 checked the divisibility)
    hipLaunchKernelGGL((wino43_vprod_kernel<TC4>), dim3((unsigned)(ntb * (32 / tl))), dim3(256), 0, s, pp, ntb);
    const unsigned grid = persistent_grid(nitems, p.ncu, p.xcd_shift);
    if (p.bn_first) hipLaunchKernelGGL((conv_wino43_kernel<false, true, TC4, false, false, true>), dim3(grid), dim3(512), 0, s, pp);
    else hipLaunchKernelGGL((conv_wino43_kernel<false, false, TC4, false, false, true>), dim3(grid), dim3(512), 0, s, pp);
    return 0;
}

// the item shape that covers the frame with fewer items (16 x 32 pixels on a tie: longer contiguous patch rows)
template <bool POOL, bool SPLIT = false>
int launch_shape(const ConvParams& p, hipStream_t s)
{
    const long long wide = (long long)((p.W + 31) / 32) * ((p.H + 15) / 16), tall = (long long)((p.W + 15) / 16) * ((p.H + 31) / 32);
    return tall < wide ? launch_q<POOL, 4, false, SPLIT>(p, s) : launch_q<POOL, 8, false, SPLIT>(p, s);
}

}  // namespace

// true when launch_conv_wino43 handles this layer shape: reflection padding (zero-padding models use conv_wino.hip), input
// channels a multiple of 8 (units of 4, unrolled in pairs), spatial size a multiple of the 4x4 tile
bool conv_wino43_supports(const ConvParams& p)
{
    // (the patch DMAs and the f32x4 stores move 16-byte granules: channel offsets and strides must be multiples of 4 floats)
    return !p.pad_zero && p.cin % 8 == 0 && p.cout % 4 == 0 && p.H % 4 == 0 && p.W % 4 == 0 && p.H >= 4 && p.W >= 4 &&
           p.in_cstride % 4 == 0 && p.in_coff % 4 == 0 && p.out_cstride % 4 == 0 && p.out_coff % 4 == 0;
}

// work items of a launch (the larger of the two item shapes' counts is never chosen): what api.hip sizes the split by
// floats of p.vglobal a pre-transformed launch of this layer needs: tile blocks x (cin / 4) units x 18 KiB
long long conv_wino43_vglobal_floats(const ConvParams& p)
{
    const long long wide = (long long)((p.W + 31) / 32) * ((p.H + 15) / 16), tall = (long long)((p.W + 15) / 16) * ((p.H + 31) / 32);
    return (long long)p.B * std::min(wide, tall) * (p.cin / UC4) * VB4;
}

long long conv_wino43_items(const ConvParams& p)
{
    const long long wide = (long long)((p.W + 31) / 32) * ((p.H + 15) / 16), tall = (long long)((p.W + 15) / 16) * ((p.H + 31) / 32);
    return (long long)p.B * std::min(wide, tall) * p.nslices;
}

// p.wpack must point at the F(4x4,3x3) weights packed by pack_wino43_weights() (api.hip).  fuse_first: the input is the
// first encoder block of p.img (p.w1 / b1 / s1 / t1, 64 channels), evaluated inside the kernel; the layer must be the pooled
// 64 -> 64 one (enc.conv2).  p.ks_shift > 0 (small launches; (cin / 4) >> ks_shift even and >= 4): the input
// channels run as 2^ks_shift items per (tile block, slice) that meet in p.split_scratch
int launch_conv_wino43(const ConvParams& p, bool pool, hipStream_t s, bool fuse_first)
{
    if (fuse_first) return pool && p.cin == 64 ? launch_q<true, 8, true>(p, s) : 2;          // 2: shape not covered
    if (p.vglobal) {       // pre-transformed input (api.hip: un-pooled layers with >= 4 output slices; cin a multiple of 16, NHWC)
        if (pool || p.ks_shift > 0 || p.cin % 16 != 0 || p.in_planar || p.cin > 256 || 256 % (p.cin / 2) != 0 || (p.cin / 2) < 8) return 2;
        const long long wide = (long long)((p.W + 31) / 32) * ((p.H + 15) / 16), tall = (long long)((p.W + 15) / 16) * ((p.H + 31) / 32);
        return tall < wide ? launch_vin<4>(p, s) : launch_vin<8>(p, s);
    }
    if (p.ks_shift > 0) {
        const int ncs = (p.cin / 4) >> p.ks_shift;
        if (ncs < 4 || (ncs & 1) || (ncs << p.ks_shift) * 4 != p.cin || !p.split_scratch) return 2;
        const int rc = pool ? launch_shape<true, true>(p, s) : launch_shape<false, true>(p, s);
        return rc ? rc : launch_split_reduce(p, 1, pool, s);
    }
    return pool ? launch_shape<true>(p, s) : launch_shape<false>(p, s);
}

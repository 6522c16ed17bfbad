// ===== NOT BUILT, NOT SHIPPED (round 5): kept with its measurements, like conv_wino43_half_window.patch. =====
// This kernel is CORRECT -- wired in behind the fused first block + conv2 launch it was bit-identical to conv_f16_res.hip's on every
// shape of tests/test_gpu_f16.py (borders, partial tiles, image lists, 1024x1280) -- and SLOWER: 2.23 ms against 1.82 ms for 16
// frames 1024x1280.  Phase timers (tools/probes/conv_timing_f16_one.py, -DMP_TIMING; ticks per item, 4608 cycles of MFMA):
//   everything                      chunk 0 steps 6891, chunk 1 steps 5393, barriers 185, decode 467     sum 12960
//   without the epilogue pieces     5590 / 4310                                                            10621
//   without the production pieces   4156 / 4224                                                             8840
//   bare MFMA + operand stream      2491 / 3115                                                             6101
// One wave per SIMD streams its MFMAs at 75-92 % of the pipe, and every vector instruction threaded between them costs 7-10 cycles
// of it (the tap gathers of chunk 0 another ~1.9 k): the "shared" prices of mfma_valu_share_f16_probe.hip (1.5-4 cycles) hold for
// independent register-only instructions, not for the dependent, LDS-fed chains of a real epilogue.  With nothing else on the SIMD
// to issue in those gaps the launch is again MFMA time + vector time, now without a second wave's slack.  To build it: copy to
// multipoint_amd/csrc/, add to build.py SOURCES, declare conv_f16_one_supports / launch_conv_f16_one in mp_common.h and call it
// from run_conv_h (api.hip) in front of launch_conv_f16_res.
// fp16 path, the fused first block + enc.conv2 launch (64 -> 64 @ full resolution + ReLU + BN + 2x2 max-pool; MultiPoint.py:99-104,
// 143-148 under autocast) as ONE wave per SIMD that interleaves ALL of its vector work with its own MFMAs.
//
// Why a third fp16 convolution kernel (round 5; DESIGN.md 3.5, profiles/r05_conv_f16_res_phases.txt, r05_mfma_valu_share_f16.txt).
// conv_f16_res.hip runs two groups of four waves per CU in anti-phase, on the premise that one group's epilogue and tile production
// hide under the other group's MFMAs.  Measured, they do not: next to another wave's MFMA stream a vector instruction costs the SIMD
// ~8 cycles ("split" rows of the probe), alone it costs ~16 cycles of latency (a wave issues in order), and the launch is the SUM of
// its MFMA steps and its vector phases (1.81 ms for 0.85 ms of MFMA) in every arrangement of the phases that was tried.  The one
// cheap arrangement is the "shared" one -- a wave that spreads its vector instructions between its OWN MFMAs pays 1.5-4 cycles for
// each -- and it needs what two groups per CU cannot have: a second accumulator set (the epilogue of item k runs during the steps
// of item k + 1) and a ring of tile buffers (the tile of item k + 1 is produced during the steps of item k).  One workgroup of FOUR
// waves per CU has both: 512 registers per wave, and 72 KiB of resident weights + three 27 KiB chunk buffers in LDS.
//
// Per item (8 x 32 output pixels x 64 channels, 2 chunks of 32 input channels x 18 steps x 4 MFMAs per wave):
//   chunk 0 steps  read buffer b0;  in their shadow: the taps of tile k+1 are gathered, its chunk 0 is produced into b2 (one MFMA per
//                  M-block, then the activation of 8 channel pairs stage by stage, 8 vector instructions per slot), and the first
//                  half of item k-1's epilogue runs from the accumulator copy;
//   barrier        (b0 consumed, b2 complete)
//   chunk 1 steps  read b1;  in their shadow: chunk 1 of tile k+1 into b0, the second half of the epilogue, the image patch of
//                  item k+2 (registers -> LDS) and of item k+3 (global -> registers);
//   accumulators -> copy;  barrier;  (b0, b1, b2) <- (b2, b0, b1).
// Every LDS read is issued at least four MFMAs before its first use; nothing in a step waits.  Same arithmetic and rounding points
// as conv_f16_res.hip's fused launch (bit-identical outputs: tests/test_gpu_f16.py).
#include "mp_common.h"

#include <algorithm>
#include <type_traits>

#ifdef MP_TIMING
__device__ unsigned long long g_timing_o[256 * 8];
extern "C" int mp_debug_read_timing_f16_one(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timing_o), sizeof(unsigned long long) * n);
}
#define MPO_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define MPO_ADD(slot, a, b) do { tsum[slot] += (b) - (a); } while (0)
#else
#define MPO_T(var) do { } while (0)
#define MPO_ADD(slot, a, b) do { } while (0)
#endif
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CK1 = 32;            // input channels per LDS chunk
constexpr int PS1 = CK1 + 8;       // LDS pixel stride in halfs (80 B: conv_f16_res.hip's conflict-free stride)
constexpr int WRES1 = 36 * 2 * 64 * 8;      // halfs of the resident weights (72 KiB)
constexpr int TW1 = 32, TH1 = 8, LW1 = TW1 + 2, LH1 = TH1 + 2, NPIX1 = LW1 * LH1;      // 8 x 32 tile + halo: 340 pixels
constexpr int STEPS1 = 9 * (CK1 / 16);
constexpr int IW1 = LW1 + 2, IH1 = LH1 + 2, NIP1 = IW1 * IH1;                           // image patch 12 x 36
constexpr int NIPB1 = ((NIP1 + 2 * IW1 + 3 + 7) / 8) * 8;
constexpr int NIPR1 = (NIP1 + 255) / 256;
constexpr int NMB1 = (NPIX1 + 31) / 32;     // 11 M-blocks of 32 tile pixels
constexpr int NJ1 = (NMB1 + 3) / 4;         // 3 per wave

__device__ __forceinline__ int reflect_clamp_o(int v, int n)
{
    v = v < 0 ? -v : v;
    v = v >= n ? 2 * (n - 1) - v : v;
    v = v < 0 ? 0 : v;
    return v >= n ? n - 1 : v;
}

__device__ __forceinline__ void wg_barrier()      // the four waves of the workgroup; LDS traffic only (no vmcnt drain)
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool BNF>
__global__ __launch_bounds__(256, 1) void conv_f16_one_kernel(const ConvParamsH p)
{
    __shared__ __attribute__((aligned(16))) _Float16 wl[WRES1];
    __shared__ __attribute__((aligned(16))) _Float16 tiles[3 * NPIX1 * PS1];
    __shared__ __attribute__((aligned(16))) _Float16 ipatch[2 * NIPB1];
    __shared__ __attribute__((aligned(16))) float prm[3 * 64];
    __shared__ __attribute__((aligned(16))) float prm1[3 * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int li = lane & 31;

    {
        const h8* const src = reinterpret_cast<const h8*>(p.wpack);
        for (int f = tid; f < WRES1 / 8; f += 256) reinterpret_cast<h8*>(wl)[f] = src[f];
        if (tid < 64) {
            prm[tid] = p.bias[tid]; prm[64 + tid] = p.scale[tid]; prm[128 + tid] = p.shift[tid];
            prm1[64 + tid] = p.s1[tid]; prm1[128 + tid] = p.t1[tid];
        }
        // patch tails: zeros, and a 1.0 where the lanes that hold taps 8..15 read "tap 9" (the bias rides in the GEMM)
        for (int f = tid; f < 2 * NIPB1; f += 256) ipatch[f] = (f % NIPB1 == NIP1 + 1) ? (_Float16)1.f : (_Float16)0.f;
    }
    __syncthreads();

    // ---- work items: the workgroup's share of its XCD's contiguous eighth ----
    const int nxcd = 1 << p.xcd_shift;
    const int per_xcd = (p.nitems + nxcd - 1) >> p.xcd_shift;
    const int xcd = (int)blockIdx.x & (nxcd - 1);
    const int stride = (int)gridDim.x >> p.xcd_shift;
    const int item_end = min((xcd + 1) * per_xcd, p.nitems);
    int item = xcd * per_xcd + ((int)blockIdx.x >> p.xcd_shift);
    if (item >= item_end) return;

    auto udiv = [](unsigned n, unsigned magic, unsigned d) -> unsigned { return d == 1 ? n : __umulhi(n, magic); };
    struct Where { int img, y0, x0; };
    auto decode = [&](int tile) __attribute__((always_inline)) -> Where {
        Where w{};
        const int trow = (int)udiv((unsigned)tile, p.magic_tx, (unsigned)p.tiles_x);
        const int tx = tile - trow * p.tiles_x;
        const int bi = (int)udiv((unsigned)trow, p.magic_ty, (unsigned)p.tiles_y);
        const int ty = trow - bi * p.tiles_y;
        w.img = p.img_list ? p.img_list[bi] : bi;
        w.y0 = ty * TH1; w.x0 = tx * TW1;
        return w;
    };

    // ---- the first encoder block (conv_f16_res.hip's F1): image patch, gather offsets, weights as the MFMA's A operand ----
    float ipx[NIPR1];
    auto patch_load = [&](const Where& w) __attribute__((always_inline)) {
        const float* const im = p.img + (long long)w.img * p.H * p.W;
#pragma unroll
        for (int i = 0; i < NIPR1; ++i) {
            const int f = min(tid + i * 256, NIP1 - 1);
            const int r = f / IW1, c = f - r * IW1;
            ipx[i] = im[reflect_clamp_o(w.y0 - 2 + r, p.H) * p.W + reflect_clamp_o(w.x0 - 2 + c, p.W)];
        }
    };
    auto patch_write = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIPR1; ++i) {
            const int f = tid + i * 256;
            if (f < NIP1) ipatch[par * NIPB1 + f] = (_Float16)ipx[i];
        }
    };
    int va1[NJ1], vb1[NJ1], wa1[NJ1];
    bool f1_rel = false;
    auto f1_offsets = [&](const Where& w) __attribute__((always_inline)) {
        const bool interior = (w.y0 >= 1) && (w.y0 + TH1 < p.H) && (w.x0 >= 1) && (w.x0 + TW1 < p.W);
        if (interior && f1_rel) return;
        f1_rel = interior;
#pragma unroll
        for (int j = 0; j < NJ1; ++j) {
            const int pix = (wave + 4 * j) * 32 + li;
            const int pc = min(pix, NPIX1 - 1);
            const int ly = pc / LW1, lx = pc - ly * LW1;
            int oy = ly, ox = lx;
            if (!interior) {
                oy = reflect_clamp_o(w.y0 + ly - 1, p.H) - w.y0 + 1;
                ox = reflect_clamp_o(w.x0 + lx - 1, p.W) - w.x0 + 1;
                oy = min(max(oy, 0), IH1 - 3); ox = min(max(ox, 0), IW1 - 3);
            }
            const int base = oy * IW1 + ox;
            va1[j] = half ? base + 2 * IW1 + 2 : base;
            vb1[j] = half ? NIP1 : base;
            wa1[j] = pix < NPIX1 ? pc * PS1 + 4 * half : -1;
        }
    };
    h8 w1f[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * half + e;
            w1f[c][e] = k < 9 ? (_Float16)p.w1[k * 64 + 32 * c + li] : k == 9 ? (_Float16)p.b1[32 * c + li] : (_Float16)0.f;
        }

    // ---- production state: taps, the MFMA result being activated, the stage registers of 8 channel pairs ----
    h8 gx[NJ1];
    f32x16 pd[2];
    f32x4 ps_[4], pt_[4];
    f32x2 qt[8];
    h2 qh[8];
    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const h2 hzero = {0, 0};
    auto gather = [&](const int j, const _Float16* const ip) __attribute__((always_inline)) {
        gx[j][0] = ip[va1[j]];
        gx[j][1] = ip[vb1[j] + 1]; gx[j][2] = ip[vb1[j] + 2];
        gx[j][3] = ip[vb1[j] + IW1]; gx[j][4] = ip[vb1[j] + IW1 + 1]; gx[j][5] = ip[vb1[j] + IW1 + 2];
        gx[j][6] = ip[vb1[j] + 2 * IW1]; gx[j][7] = ip[vb1[j] + 2 * IW1 + 1];
    };
    auto prod_params = [&](const int c) __attribute__((always_inline)) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            ps_[rg] = *reinterpret_cast<const f32x4*>(&prm1[64 + 32 * c + rg * 8 + half * 4]);
            pt_[rg] = *reinterpret_cast<const f32x4*>(&prm1[128 + 32 * c + rg * 8 + half * 4]);
        }
    };
    // piece i (0 .. 3 + 7 * NJ1 - 1) of the production of chunk c of the next tile into the buffer at `dst` (halfs):
    //   pieces 0, 1, 2 = M(0), M(1) and -- behind A(0) -- M(2) are folded into the list below:
    //   list: M0, M1, A0s0..A0s5, M2, A1s0..A1s5, A2s0..A2s5      (As5 also stores the four 8-byte granules)
    constexpr int NPP = 3 + 6 * NJ1;                               // 21
    auto prod_piece = [&](const int i, const int c, _Float16* const dst) __attribute__((always_inline)) {
        if (i < 0 || i >= NPP) return;
        // decode the list position
        int kind, j, st;          // kind 0: MFMA of block j; 1: activation stage st of block j
        if (i == 0) { kind = 0; j = 0; st = 0; }
        else if (i == 1) { kind = 0; j = 1; st = 0; }
        else if (i < 8) { kind = 1; j = 0; st = i - 2; }
        else if (i == 8) { kind = 0; j = 2; st = 0; }
        else if (i < 15) { kind = 1; j = 1; st = i - 9; }
        else { kind = 1; j = 2; st = i - 15; }
        if (kind == 0) {
            pd[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[c], gx[j], z16, 0, 0, 0);
            return;
        }
        const f32x16& d = pd[j & 1];
        if (st == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) qh[q] = __builtin_convertvector(f32x2{d[2 * q], d[2 * q + 1]}, h2);      // conv output (bias included) -> fp16
        } else if (st == 1) {
            if (!BNF) {
#pragma unroll
                for (int q = 0; q < 8; ++q) qh[q] = __builtin_elementwise_max(qh[q], hzero);
            }
        } else if (st == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) qt[q] = __builtin_convertvector(qh[q], f32x2);
        } else if (st == 3) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 sv = ps_[q >> 1], tv = pt_[q >> 1];
                qt[q] = qt[q] * ((q & 1) ? f32x2{sv[2], sv[3]} : f32x2{sv[0], sv[1]}) + ((q & 1) ? f32x2{tv[2], tv[3]} : f32x2{tv[0], tv[1]});
            }
        } else if (st == 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                qh[q] = __builtin_convertvector(qt[q], h2);
                if (BNF) qh[q] = __builtin_elementwise_max(qh[q], hzero);
            }
        } else {
            // lane = pixel, pair q = channels 32c + 8 (q >> 1) + 4 half + 2 (q & 1) + {0, 1}: a granule = 4 consecutive channels
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const h4 v = h4{qh[2 * rg][0], qh[2 * rg][1], qh[2 * rg + 1][0], qh[2 * rg + 1][1]};
                if (wa1[j] >= 0) *reinterpret_cast<h4*>(&dst[wa1[j] + rg * 8]) = v;
            }
        }
    };

    // ---- epilogue state: the previous item's accumulators, its position, the stage registers of 8 pixel pairs ----
    f32x16 accP[2][2];
    Where prev{};
    bool have_prev = false;      // false during the first item: its epilogue pieces run on garbage with their stores masked
    f32x2 bia[2], scl[2], sft[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const float b = prm[nb * 32 + li], sc = prm[64 + nb * 32 + li], sh = prm[128 + nb * 32 + li];
        bia[nb] = f32x2{b, b}; scl[nb] = f32x2{sc, sc}; sft[nb] = f32x2{sh, sh};
    }
    f32x2 et[8];
    h2 eh[8];
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int cs = p.out_cstride;
    // piece i (0 .. 27) of the epilogue of the previous item: batch = i / 7 = (nb, register half), stage = i % 7.
    // lane = channel li of block nb; register r of M-block mb = pixel (r&3) + 8 (r>>2) + 4 half of tile row 2 wave + mb; the batch
    // holds registers r0 .. r0+7 (four horizontal pairs) of both M-blocks: pairs 0..3 = row 2 wave, 4..7 = row 2 wave + 1
    constexpr int NEP = 28;
    auto epi_piece = [&](const int i) __attribute__((always_inline)) {
        if (i < 0 || i >= NEP) return;
        const int b = i / 7, st = i % 7;
        const int nb = b >> 1, r0 = (b & 1) * 8;
        if (st == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                et[q] = f32x2{accP[0][nb][r0 + 2 * q], accP[0][nb][r0 + 2 * q + 1]} + bia[nb];
                et[4 + q] = f32x2{accP[1][nb][r0 + 2 * q], accP[1][nb][r0 + 2 * q + 1]} + bia[nb];
            }
        } else if (st == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                eh[q] = __builtin_convertvector(et[q], h2);
                if (!BNF) eh[q] = __builtin_elementwise_max(eh[q], hzero);
            }
        } else if (st == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) et[q] = __builtin_convertvector(eh[q], f32x2);
        } else if (st == 3) {
#pragma unroll
            for (int q = 0; q < 8; ++q) et[q] = et[q] * scl[nb] + sft[nb];
        } else if (st == 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                eh[q] = __builtin_convertvector(et[q], h2);
                if (BNF) eh[q] = __builtin_elementwise_max(eh[q], hzero);
            }
        } else if (st == 5) {
#pragma unroll
            for (int q = 0; q < 4; ++q) eh[q] = __builtin_elementwise_max(eh[q], eh[4 + q]);       // the 2x2 window's two rows
        } else {
            const int oy = (prev.y0 + 2 * wave) >> 1;
            _Float16* const obase = p.out + ((long long)prev.img * Ho * Wo) * cs + p.out_coff + (long long)oy * Wo * cs + nb * 32 + li;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 2 * q;
                const int iu = (r & 3) + 8 * (r >> 2);                                   // pixel column of the pair inside the tile (+ 4 half)
                const int ox = (prev.x0 + iu) >> 1;
                const _Float16 v = eh[q][0] > eh[q][1] ? eh[q][0] : eh[q][1];
                const bool ok = have_prev & (oy < Ho) & (ox + 2 * half < Wo);
                _Float16* const dstp = ok ? obase + (long long)(ox + 2 * half) * cs : p.dummy + lane;
                *dstp = v;
            }
        }
    };

    // ---- operands of the main GEMM ----
    const int a_base = (((2 * wave) + li / TW1) * LW1 + (li % TW1)) * PS1 + half * 8;       // MBH = 1: M-block = one tile row
    constexpr int A_MB = LW1 * PS1;
    constexpr int RA = 4;                                  // operand rings: fragments are fetched three steps ahead
    h8 af[RA][2], bf[RA][2];
    auto a_off = [](int s) -> int {
        const int tap = s >> 1, gg = s & 1;
        return ((tap / 3) * LW1 + tap % 3) * PS1 + gg * 16;
    };

    // ---- prologue: the first tile in phase form, the patches of the next two items ----
    Where cur = decode(item);
    int b0 = 0, b1 = NPIX1 * PS1, b2 = 2 * NPIX1 * PS1;    // chunk-0 buffer, chunk-1 buffer, spare (halfs into tiles[])
    int par = 0;
    patch_load(cur);
    patch_write(0);
    f1_offsets(cur);
    wg_barrier();
    {
#pragma unroll
        for (int j = 0; j < NJ1; ++j) gather(j, ipatch);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            prod_params(c);
#pragma unroll
            for (int i = 0; i < NPP; ++i) prod_piece(i, c, tiles + (c ? b1 : b0));
        }
    }
    if (item + stride < item_end) {
        const Where n1 = decode(item + stride);
        patch_load(n1);
        patch_write(1);
        f1_offsets(n1);                                    // the gathers inside item k are the NEXT item's
    }
    if (item + 2 * stride < item_end) patch_load(decode(item + 2 * stride));
    wg_barrier();

    const f32x16 zero16 = z16;
#ifdef MP_TIMING
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        MPO_T(t0);
        f32x16 acc[2][2];
        const int item_next = item + stride;
        const bool has_next = item_next < item_end;
        const _Float16* const ipn = ipatch + (par ^ 1) * NIPB1;             // the next item's image patch

        auto chunk_body = [&](auto c_tag) __attribute__((always_inline)) {
            constexpr int C = decltype(c_tag)::value;
            const _Float16* const tb = tiles + (C ? b1 : b0) + a_base;
            _Float16* const pdst = tiles + (C ? b0 : b2);                  // where the next tile's chunk C is produced
            auto w_off = [](int s) -> int { return (((s >> 1) * 4 + 2 * C + (s & 1)) * 2) * 512; };
#pragma unroll
            for (int s = 0; s < RA - 1; ++s) {
                af[s][0] = *reinterpret_cast<const h8*>(&tb[a_off(s)]);
                af[s][1] = *reinterpret_cast<const h8*>(&tb[A_MB + a_off(s)]);
                bf[s][0] = *reinterpret_cast<const h8*>(&wl[w_off(s) + lane * 8]);
                bf[s][1] = *reinterpret_cast<const h8*>(&wl[w_off(s) + 512 + lane * 8]);
            }
            prod_params(C);
            if (C == 0) {
#pragma unroll
                for (int j = 0; j < NJ1; ++j) gather(j, ipn);
            }
            // slots q = 4 s + m: production pieces on the even slots from slot P0 on (the gathers of chunk 0 have landed by then),
            // epilogue pieces on every fourth odd slot
            constexpr int P0 = (C == 0) ? 12 : 2;
#pragma unroll
            for (int s = 0; s < STEPS1; ++s) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    constexpr bool Z = (C == 0);
                    const int q = 4 * s + m;
                    f32x16& a = acc[m >> 1][m & 1];
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s % RA][m >> 1], bf[s % RA][m & 1], (Z && s == 0) ? zero16 : a, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m == 0 && s + RA - 1 < STEPS1) {
                        const int sn = s + RA - 1;
                        bf[sn % RA][0] = *reinterpret_cast<const h8*>(&wl[w_off(sn) + lane * 8]);
                        bf[sn % RA][1] = *reinterpret_cast<const h8*>(&wl[w_off(sn) + 512 + lane * 8]);
                    }
                    if (m == 3 && s + RA - 1 < STEPS1) {
                        const int sn = s + RA - 1;
                        af[sn % RA][0] = *reinterpret_cast<const h8*>(&tb[a_off(sn)]);
                        af[sn % RA][1] = *reinterpret_cast<const h8*>(&tb[A_MB + a_off(sn)]);
                    }
#ifndef MPOX
#define MPOX 0
#endif
                    if (!(MPOX & 2) && (q & 1) == 0 && q >= P0) prod_piece((q - P0) >> 1, C, pdst);
                    if (!(MPOX & 1) && (q & 3) == 1 && (q >> 2) < 14) epi_piece((C ? 14 : 0) + (q >> 2));      // (first item: garbage in, stores masked)
                    if (C == 1 && q == 59 && item_next + stride < item_end) patch_write(par);          // patch of item k+2 (its buffer was read while tile k was produced)
                    if (C == 1 && q == 63 && item_next + 2 * stride < item_end) patch_load(decode(item_next + 2 * stride));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        chunk_body(std::integral_constant<int, 0>{});
        MPO_T(t1);
        wg_barrier();                                          // b0 consumed by every wave, chunk 0 of the next tile complete in b2
        MPO_T(t2);
        chunk_body(std::integral_constant<int, 1>{});
        MPO_T(t3);
        MPO_ADD(0, t0, t1); MPO_ADD(1, t1, t2); MPO_ADD(2, t2, t3);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) accP[a][b] = acc[a][b];
        prev = cur;
        have_prev = true;
        MPO_T(t4);
        MPO_ADD(3, t3, t4);
#ifdef MP_TIMING
        tsum[7] += 1;
        if (!has_next && tid == 0) for (int i = 0; i < 8; ++i) g_timing_o[blockIdx.x * 8 + i] = tsum[i];
#endif
        if (!has_next) break;
        item = item_next;
        cur = decode(item);
        if (item + stride < item_end) f1_offsets(decode(item + stride));
        par ^= 1;
        MPO_T(t5);
        wg_barrier();                                          // b1 consumed, chunk 1 of the next tile complete in b0, the patch of item k+2 visible
        MPO_T(t6);
        MPO_ADD(4, t4, t5); MPO_ADD(5, t5, t6);
        { const int t = b0; b0 = b2; b2 = b1; b1 = t; }        // (chunk 0, chunk 1, spare) <- (b2, b0, b1)
    }
    // the last item's epilogue, in phase form
#pragma unroll
    for (int i = 0; i < NEP; ++i) epi_piece(i);
}

}  // namespace

// the fused first block + pooled 64 -> 64 layer at the 8 x 32 tile shape, reflection padding (everything else: conv_f16_res.hip)
bool conv_f16_one_supports(const ConvParamsH& p, int mbw, bool pool)
{
    return p.img != nullptr && pool && !p.pad_zero && mbw == 32 && p.cin == 64 && p.cout == 64 && p.nslices == 1 && p.out_cstride % 1 == 0;
}

int launch_conv_f16_one(const ConvParamsH& p, hipStream_t s)
{
    const long long nitems = (long long)p.B * p.tiles_x * p.tiles_y;
    if (nitems <= 0) return 0;
    ConvParamsH q = p;
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); };
    q.magic_tx = magic(p.tiles_x); q.magic_ty = magic(p.tiles_y);
    const long long dmax = std::max(p.tiles_x, p.tiles_y);
    if (nitems * dmax >= 0x100000000ll) return 1;
    q.nitems = (int)nitems;
    const unsigned grid = persistent_grid(nitems, p.ncu, p.xcd_shift);
    const ConvParamsH& pp = q;
    if (p.bn_first) hipLaunchKernelGGL((conv_f16_one_kernel<true>), dim3(grid), dim3(256), 0, s, pp);
    else hipLaunchKernelGGL((conv_f16_one_kernel<false>), dim3(grid), dim3(256), 0, s, pp);
    return 0;
}

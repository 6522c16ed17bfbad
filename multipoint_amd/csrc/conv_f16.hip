// fp16 convolution path (model config `mixed_precision: true`, reference MultiPoint.py:99-103: the forward
// runs under torch.cuda.amp.autocast, i.e. Conv2d on fp16 inputs/weights with fp32 accumulation and fp16
// outputs, BatchNorm evaluated in fp32 on fp16 activations, fp16 results).  BASELINE config 5.
//
// Same implicit GEMM as conv_mfma.hip with the byte geometry kept: an LDS pixel is 64 fp16 channels (128 B,
// stride 144 B -> conflict-free ds_read_b128), a step is (tap, 16-channel group) and one 16-byte operand per
// lane feeds ONE v_mfma_f32_32x32x16_f16 (32 cycles, 16x the fp32 instruction's rate), so the operand
// streams have to be fetched much further ahead: weights 5 steps (ring of 6), activations 2 steps (ring of 3).
// Rounding points (all round-to-nearest-even, as autocast produces them):
//   conv: fp32 accumulate (+ fp16 bias) -> fp16;  ReLU;  BN: fp32 affine on the fp16 value -> fp16;  max-pool.
#include "mp_common.h"

#include <algorithm>
#include <type_traits>

#ifdef MP_TIMING
// developer instrumentation: per-workgroup cycle sums per phase (wave 0), see tools/conv_timing_f16.py
__device__ unsigned long long g_timing_h[512 * 8];
__device__ int g_timing_h_sel = 1024;       // only launches whose input height matches are recorded
extern "C" int mp_debug_select_height_f16(int h) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_timing_h_sel), &h, sizeof(int)); }
extern "C" int mp_debug_read_timing_f16(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timing_h), sizeof(unsigned long long) * n);
}
#define MPH_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define MPH_ADD(slot, a, b) do { tsum[slot] += (b) - (a); } while (0)      // wave-uniform register accumulators
#else
#define MPH_T(var) do { } while (0)
#define MPH_ADD(slot, a, b) do { } while (0)
#endif

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int CKH = 64;        // input channels per LDS chunk
constexpr int PSH = CKH + 8;   // LDS pixel stride in halfs (144 B)

template <int TAPS, int MBW>
struct GeoH {
    static constexpr int MBH = 32 / MBW;
    static constexpr int TW = MBW;
    static constexpr int TH = 256 / MBW;
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int LW = TW + 2 * HALO;
    static constexpr int LH = TH + 2 * HALO;
    static constexpr int NPIX = LW * LH;
    static constexpr int NV = NPIX * (CKH / 8);          // 16-byte vectors per chunk tile
    static constexpr int NITER = (NV + 255) / 256;       // staging vectors per thread
    static constexpr int STEPS = TAPS * (CKH / 16);      // k16-steps per chunk
};

__device__ __forceinline__ int reflect_clamp_h(int v, int n)
{
    v = v < 0 ? -v : v;
    v = v >= n ? 2 * (n - 1) - v : v;
    v = v < 0 ? 0 : v;
    return v >= n ? n - 1 : v;
}

__device__ __forceinline__ float round_h(float v) { return (float)(_Float16)v; }

// conv result (fp32 accumulator) -> activation value as autocast produces it (still held in fp32)
template <bool RELU, bool BNF>
__device__ __forceinline__ float act_h(float acc, float bias, float scale, float shift)
{
    float v = round_h(acc + bias);
    if (BNF) {
        v = round_h(v * scale + shift);
        if (RELU) v = fmaxf(v, 0.f);
    } else {
        if (RELU) v = fmaxf(v, 0.f);
        v = round_h(v * scale + shift);
    }
    return v;
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// the same on a pair, written so that hipcc emits packed instructions (v_pk_add_f32, v_cvt_pk_f16_f32,
// v_pk_max_f16, v_pk_fma_f32): ReLU commutes with the rounding, so it runs on the packed halves
template <bool RELU, bool BNF>
__device__ __forceinline__ h2 act_h2(float a0, float a1, f32x2 bias, f32x2 scale, f32x2 shift)
{
    const f32x2 x = f32x2{a0, a1} + bias;
    h2 h = __builtin_convertvector(x, h2);
    const h2 zero = {0, 0};
    if (RELU && !BNF) h = __builtin_elementwise_max(h, zero);
    f32x2 y = __builtin_convertvector(h, f32x2) * scale + shift;
    asm volatile("" : "+v"(y));      // y exists as an fp32 pair (autocast: BatchNorm result in fp32, THEN fp16): no v_fma_mixlo_f16, which rounds once
    h2 o = __builtin_convertvector(y, h2);
    if (RELU && BNF) o = __builtin_elementwise_max(o, zero);
    return o;
}

template <int TAPS, int MBW, bool POOL, bool BNF>
__global__ __launch_bounds__(256, 2) void conv_f16_kernel(const ConvParamsH p)
{
    using G = GeoH<TAPS, MBW>;
    constexpr bool RELU = (TAPS == 9);
    constexpr bool SWAP = !POOL;      // weights as the MFMA A operand -> lane = pixel, register quad = 4 channels
    __shared__ __attribute__((aligned(16))) _Float16 lds[G::NPIX * PSH];
    // bias | BN scale | BN shift of the current 64-channel slice: fetched from global memory only when the slice
    // changes (a global load in every epilogue costs its L2 latency per item)
    __shared__ __attribute__((aligned(16))) float prm[3 * 64];
    // un-pooled layers: a wave's output block (32 pixels x 64 channels, 4 KiB) passes through LDS so that a lane stores 16 bytes and eight
    // lanes a pixel's whole 128-byte line (conv_f16_res.hip's epilogue); 16-byte granules XOR-swizzled by the pixel's low bits
    __shared__ __attribute__((aligned(16))) _Float16 ostage[POOL ? 8 : 4 * 2048];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int li = lane & 31;

    // ---- persistent workgroup: walks work items (tile, slice) of its XCD's contiguous eighth of the item space,
    //      gridDim.x/8 items apart, and fetches the NEXT item's activation tile while multiplying the current one
    //      (an fp16 tile is only ~4.6 k cycles of MFMA: without the prefetch the HBM latency of every tile is exposed)
    const XcdRange xr = xcd_range(p.nitems, p.xcd_shift);
    const int stride = xr.stride, item_end = xr.item_end;
    int item = xr.item;

    auto udiv = [](unsigned n, unsigned magic, unsigned d) -> unsigned { return d == 1 ? n : __umulhi(n, magic); };
    struct Where { int slice, img, y0, x0; long long px0; const _Float16* in_base; };
    auto decode = [&](int it) __attribute__((always_inline)) -> Where {
        Where w{};
        const int tile = (int)udiv((unsigned)it, p.magic_slices, (unsigned)p.nslices);
        w.slice = it - tile * p.nslices;
        if constexpr (TAPS == 9) {
            const int trow = (int)udiv((unsigned)tile, p.magic_tx, (unsigned)p.tiles_x);
            const int tx = tile - trow * p.tiles_x;
            const int bi = (int)udiv((unsigned)trow, p.magic_ty, (unsigned)p.tiles_y);
            const int ty = trow - bi * p.tiles_y;
            w.img = p.img_list ? p.img_list[bi] : bi;
            w.y0 = ty * G::TH; w.x0 = tx * G::TW;
            w.in_base = p.in + (long long)w.img * p.H * p.W * p.in_cstride + p.in_coff;
        } else {
            w.px0 = (long long)tile * 256;
            w.in_base = p.in + w.px0 * p.in_cstride + p.in_coff;
        }
        return w;
    };
    // per-thread staging offsets in halfs.  Interior items (every halo pixel inside the image; all pixels valid in
    // flat mode): offsets relative to the item's first halo pixel, IDENTICAL for every such item, so they are
    // computed once and only the wave-uniform base pointer moves (no VALU per item: a wave that is not streaming
    // MFMAs issues its vector instructions very slowly next to one that is).  Boundary items: absolute offsets from
    // in_base with -1 marking padding slots that are zero-filled at the LDS write.
    int goff[G::NITER];
    bool goff_rel = false;          // goff currently holds the item-invariant relative offsets
    bool cur_pad = false;           // the LDS image being written has padding slots
    auto is_interior = [&](const Where& w) __attribute__((always_inline)) -> bool {
        if constexpr (TAPS == 9) return (w.y0 >= 1) && (w.y0 + G::TH < p.H) && (w.x0 >= 1) && (w.x0 + G::TW < p.W);
        else return w.px0 + 256 <= p.total_px;
    };
    // returns the base pointer the staging loads of item w add goff to
    auto offsets = [&](const Where& w) __attribute__((always_inline)) -> const _Float16* {
        if (is_interior(w)) {
            if (!goff_rel) {
#pragma unroll
                for (int j = 0; j < G::NITER; ++j) {
                    const int f = tid + j * 256;
                    const int lp = f >> 3;
                    int off = 0;
                    if (f < G::NV) {
                        if constexpr (TAPS == 9) {
                            const int ly = lp / G::LW, lx = lp - ly * G::LW;
                            off = (ly * p.W + lx) * p.in_cstride + (f & 7) * 8;
                        } else {
                            off = lp * p.in_cstride + (f & 7) * 8;
                        }
                    }
                    goff[j] = off;
                }
                goff_rel = true;
            }
            if constexpr (TAPS == 9) return w.in_base + (long long)((w.y0 - 1) * p.W + (w.x0 - 1)) * p.in_cstride;
            else return w.in_base;
        }
        goff_rel = false;
#pragma unroll
        for (int j = 0; j < G::NITER; ++j) {
            const int f = tid + j * 256;
            const int lp = f >> 3, c8 = f & 7;
            int off = -1;
            if (f < G::NV) {
                if constexpr (TAPS == 9) {
                    const int ly = lp / G::LW, lx = lp - ly * G::LW;
                    int gy = w.y0 + ly - 1, gx = w.x0 + lx - 1;
                    bool zero = false;
                    if (p.pad_zero) {
                        zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
                        gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
                    } else {
                        gy = reflect_clamp_h(gy, p.H); gx = reflect_clamp_h(gx, p.W);
                    }
                    if (!zero) off = (gy * p.W + gx) * p.in_cstride + c8 * 8;
                } else {
                    if (w.px0 + lp < p.total_px) off = lp * p.in_cstride + c8 * 8;
                }
            }
            goff[j] = off;
        }
        return w.in_base;
    };

    const int a_base = (((2 * wave) * G::MBH + li / MBW) * G::LW + (li % MBW)) * PSH + half * 8;
    constexpr int A_MB = G::MBH * G::LW * PSH;
    const int nchunks = p.cin / CKH;

    constexpr int RB = (G::STEPS % 6 == 0) ? 6 : 4;      // weight ring (divides STEPS: stays aligned across chunks)
    constexpr int PF = RB - 1;                           // steps of weight prefetch
    constexpr int RA = 3;                                // activation-fragment ring (restarted per chunk)
    static_assert(G::STEPS % RB == 0, "weight ring must stay aligned across chunks");
    h8 af[RA][2], bf[RB][2], stg[G::NITER];
    constexpr int S0 = (TAPS == 9) ? 4 : 0;               // first step that issues a staging load
    constexpr int PER_STEP = (TAPS == 9) ? 1 : 2;
    auto a_off = [](int s) -> int {
        const int tap = s >> 2, gg = s & 3;
        const int kh = (TAPS == 9) ? tap / 3 : 0, kw = (TAPS == 9) ? tap % 3 : 0;
        return (kh * G::LW + kw) * PSH + gg * 16;
    };
    auto mma = [](const h8& a, const h8& b, const f32x16& cc) -> f32x16 {
        return SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, cc, 0, 0, 0)
                    : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, cc, 0, 0, 0);
    };

    if (item >= item_end) return;
    Where cur = decode(item);
    const _Float16* src = offsets(cur);                    // staging base of the item being loaded
    cur_pad = !goff_rel;
#pragma unroll
    for (int j = 0; j < G::NITER; ++j)
        stg[j] = *reinterpret_cast<const h8*>(src + (goff[j] >= 0 ? goff[j] : 0));

    auto lds_write = [&]() __attribute__((always_inline)) {
        if (!cur_pad) {
#pragma unroll
            for (int j = 0; j < G::NITER; ++j) {
                const int f = tid + j * 256;
                if (f < G::NV) *reinterpret_cast<h8*>(&lds[(f >> 3) * PSH + (f & 7) * 8]) = stg[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < G::NITER; ++j) {
                const int f = tid + j * 256;
                if (f < G::NV) {
                    h8 v = stg[j];
                    if (goff[j] < 0) v = h8{0, 0, 0, 0, 0, 0, 0, 0};
                    *reinterpret_cast<h8*>(&lds[(f >> 3) * PSH + (f & 7) * 8]) = v;
                }
            }
        }
    };
    lds_write();
    auto load_prm = [&](int slice) __attribute__((always_inline)) {
        if (tid < 64) {
            prm[tid] = p.bias[slice * 64 + tid]; prm[64 + tid] = p.scale[slice * 64 + tid]; prm[128 + tid] = p.shift[slice * 64 + tid];
        }
    };
    load_prm(cur.slice);
    // weights: [slice][chunk][step][nb][lane][8 halfs]; ONE linear stream per item, and the tail of an item's
    // prefetch already reads the head of the next item's stream, so it is never restarted after this point
    const h8* wp = reinterpret_cast<const h8*>(p.wpack) + ((long long)cur.slice * nchunks) * (G::STEPS * 128) + lane;
#pragma unroll
    for (int s = 0; s < PF; ++s) { bf[s][0] = wp[s * 128]; bf[s][1] = wp[s * 128 + 64]; }
    __syncthreads();

#ifdef MP_TIMING
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool t_on = (p.H == g_timing_h_sel);      // read once: a global load inside the loop would add a vmcnt(0) wait
#endif
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (;;) {
        MPH_T(t_item);
        f32x16 acc[2][2];      // not zero-initialised: the first MFMAs of the item take a literal-zero C operand
        const int item_next = item + stride;
        const bool has_next = item_next < item_end;
        Where nxt = cur;
        const h8* wnext = wp;
        const _Float16* cur_src = src;
        bool nxt_pad = cur_pad;

        auto chunk_body = [&](const int c, auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            // source of the staging loads issued during this chunk: the next chunk of this item, or chunk 0 of the
            // next item (whose offsets replace goff: the LDS image of the current chunk is already written)
            const bool last = c + 1 == nchunks;
            const _Float16* in_next;
            if (!last) {
                in_next = cur_src + (c + 1) * CKH;
            } else {
                if (has_next) {
                    nxt = decode(item_next);
                    src = offsets(nxt);
                    nxt_pad = !goff_rel;
                    // (from an opaque copy of the lane id: `p.wpack + lane` is item-invariant, and hipcc kept it as a 64-bit register pair
                    // through the MFMA loop -- spilled, and its reload here was an s_waitcnt vmcnt(0) on the loads in flight)
                    int lw = lane;
                    asm volatile("" : "+v"(lw));
                    wnext = reinterpret_cast<const h8*>(p.wpack) + ((long long)nxt.slice * nchunks) * (G::STEPS * 128) + lw;
                }
                in_next = src;                                  // !has_next: dummy re-read of the current tile
            }
            const h8* wc = wp + (long long)c * (G::STEPS * 128);
            const h8* wt = last ? wnext : wc + G::STEPS * 128;   // where the prefetch continues after this chunk
            MPH_T(t_steps0);
            if (c == 0) MPH_ADD(0, t_item, t_steps0);          // item start -> first step (decode, offsets)
#pragma unroll
            for (int s = 0; s < RA - 1; ++s) {
                af[s][0] = *reinterpret_cast<const h8*>(&lds[a_base + a_off(s)]);
                af[s][1] = *reinterpret_cast<const h8*>(&lds[a_base + A_MB + a_off(s)]);
            }
#pragma unroll
            for (int s = 0; s < G::STEPS; ++s) {
                // ONE memory instruction behind each MFMA: issued as a burst in front of the four MFMAs of a step, the same
                // loads cost ~45 cycles of matrix-pipe time each instead of ~10 (round-2 probe; docs/HISTORY.md A.3)
                const h8* const wsrc = (s + PF < G::STEPS) ? wc + (s + PF) * 128 : wt + (s + PF - G::STEPS) * 128;
                acc[0][0] = mma(af[s % RA][0], bf[s % RB][0], (FIRST && s == 0) ? zero16 : acc[0][0]);
                __builtin_amdgcn_sched_barrier(0);
                bf[(s + PF) % RB][0] = wsrc[0];
                __builtin_amdgcn_sched_barrier(0);
                acc[0][1] = mma(af[s % RA][0], bf[s % RB][1], (FIRST && s == 0) ? zero16 : acc[0][1]);
                __builtin_amdgcn_sched_barrier(0);
                bf[(s + PF) % RB][1] = wsrc[64];
                __builtin_amdgcn_sched_barrier(0);
                acc[1][0] = mma(af[s % RA][1], bf[s % RB][0], (FIRST && s == 0) ? zero16 : acc[1][0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PER_STEP; ++u) {
                    const int j = (s - S0) * PER_STEP + u;
                    if (s >= S0 && j < G::NITER)      // unconditional load: keeps the compiler's vmcnt counting exact
                        stg[j] = *reinterpret_cast<const h8*>(in_next + (goff[j] >= 0 ? goff[j] : 0));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[1][1] = mma(af[s % RA][1], bf[s % RB][1], (FIRST && s == 0) ? zero16 : acc[1][1]);
                __builtin_amdgcn_sched_barrier(0);
                if (s + RA - 1 < G::STEPS) {
                    const int sn = s + RA - 1;
                    af[sn % RA][0] = *reinterpret_cast<const h8*>(&lds[a_base + a_off(sn)]);
                    af[sn % RA][1] = *reinterpret_cast<const h8*>(&lds[a_base + A_MB + a_off(sn)]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            MPH_T(t_steps1);
            MPH_ADD(1, t_steps0, t_steps1);                    // the MFMA steps of a chunk
            __syncthreads();                                   // this chunk's LDS image fully consumed
            MPH_T(t_bar);
            MPH_ADD(2, t_steps1, t_bar);                       // barrier skew
            if (last) cur_pad = nxt_pad;
            if (!last || has_next) lds_write();                // staged registers are free again before the epilogue
            MPH_T(t_ldsw);
            MPH_ADD(3, t_bar, t_ldsw);                         // staging data landed + LDS write
            if (!last) __syncthreads();
        };
        chunk_body(0, std::true_type{});
        for (int c = 1; c < nchunks; ++c) chunk_body(c, std::false_type{});
        MPH_T(t_epi0);

        // ---------------- epilogue of item `cur` ----------------
        const int slice = cur.slice, img = cur.img, y0 = cur.y0, x0 = cur.x0;
        const long long px0 = cur.px0;
        // Addressing: wave-uniform 64-bit base (scalar unit) + one per-lane 32-bit offset computed once per item + uniform
        // per-store increments.  `full` items (tile entirely inside the output) store unconditionally; partial tiles
        // send masked lanes to a dummy line so that BOTH paths issue the same number of stores and hipcc's vmcnt
        // counting stays exact (a guarded store would make the next item's first operand wait cover every store).
        if constexpr (POOL) {
            // lane = channel (li), register r = pixel (r&3) + 8*(r>>2) + 4*half of the M-block; registers r, r+1
            // are horizontally adjacent pixels -> one packed pair
            // (the lane's item-invariant epilogue values -- parameter addresses, store offset -- are derived from an OPAQUE copy of the lane id,
            // so that hipcc recomputes them here instead of carrying them through the MFMA loop: it spilled 7 of them to scratch)
            int lq = lane;
            asm volatile("" : "+v"(lq));
            const int li = lq & 31, half = lq >> 5;
            f32x2 bia[2], scl[2], sft[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const float b = prm[nb * 32 + li], sc = prm[64 + nb * 32 + li], sh = prm[128 + nb * 32 + li];
                bia[nb] = f32x2{b, b}; scl[nb] = f32x2{sc, sc}; sft[nb] = f32x2{sh, sh};
            }
            const int Ho = p.H >> 1, Wo = p.W >> 1;
            const int cs = p.out_cstride;
            // Pool BEFORE the activation: bias add, fp16 rounding and ReLU are non-decreasing and the BatchNorm affine is monotonic in the direction of its
            // scale's sign, so the maximum of a window's four activations IS the activation of the maximum (scale < 0: the minimum) of its
            // four accumulators, bit for bit -- one activation per pooled value instead of four (channels li and 32 + li share a packed pair)
            const f32x2 biap = {bia[0][0], bia[1][0]}, sclp = {scl[0][0], scl[1][0]}, sftp = {sft[0][0], sft[1][0]};
            const bool neg0 = sclp[0] < 0.f, neg1 = sclp[1] < 0.f;
            auto pooled_first = [&](const float (&q)[2][4]) __attribute__((always_inline)) -> h2 {
                const float x0 = fmaxf(fmaxf(q[0][0], q[0][1]), fmaxf(q[0][2], q[0][3])), n0 = fminf(fminf(q[0][0], q[0][1]), fminf(q[0][2], q[0][3]));
                const float x1 = fmaxf(fmaxf(q[1][0], q[1][1]), fmaxf(q[1][2], q[1][3])), n1 = fminf(fminf(q[1][0], q[1][1]), fminf(q[1][2], q[1][3]));
                return act_h2<RELU, BNF>(neg0 ? n0 : x0, neg1 ? n1 : x1, biap, sclp, sftp);
            };
            const bool full = (y0 + G::TH <= p.H) && (x0 + G::TW <= p.W) && (slice * 64 + 64 <= p.cout);
            const int lane_off = 2 * half * cs + li;
            _Float16* const obase = p.out + ((long long)img * Ho * Wo) * cs + p.out_coff + slice * 64;
            // MBW == 32: rows 2*wave (mb 0) and 2*wave+1 (mb 1) pool together; otherwise both rows of a window are
            // registers r and r+RDOWN of one M-block
            constexpr int RDOWN = (MBW == 32) ? 0 : (MBW == 16) ? 8 : 4;
            constexpr int NMB = (MBW == 32) ? 1 : 2;
            auto store_all = [&](auto full_tag) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        if (RDOWN != 0 && (r & RDOWN) != 0) continue;
                        const int iu = (r & 3) + 8 * (r >> 2);                       // lane-independent part of the pixel index
                        const int oy = (MBW == 32) ? (y0 + 2 * wave) >> 1 : (y0 + (2 * wave + mb) * G::MBH + iu / MBW) >> 1;
                        const int oxu = (x0 + iu % MBW) >> 1;                        // + 2*half per lane
                        _Float16* const rowp = obase + ((long long)oy * Wo + oxu) * cs;
                        h2 vp;
                        {
                            float q[2][4];
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb) {
                                if constexpr (MBW == 32) {
                                    q[nb][0] = acc[0][nb][r]; q[nb][1] = acc[0][nb][r + 1]; q[nb][2] = acc[1][nb][r]; q[nb][3] = acc[1][nb][r + 1];
                                } else {
                                    q[nb][0] = acc[mb][nb][r]; q[nb][1] = acc[mb][nb][r + 1];
                                    q[nb][2] = acc[mb][nb][r + RDOWN]; q[nb][3] = acc[mb][nb][r + RDOWN + 1];
                                }
                            }
                            vp = pooled_first(q);
                        }
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const _Float16 v = vp[nb];
                            if constexpr (FULL) {
                                rowp[nb * 32 + lane_off] = v;
                            } else {
                                const bool ok = (oy < Ho) & (oxu + 2 * half < Wo) & (slice * 64 + nb * 32 + li < p.cout);
                                _Float16* dst = ok ? rowp + nb * 32 + lane_off : p.dummy + lq;
                                *dst = v;
                            }
                        }
                    }
            };
            if (full) store_all(std::true_type{}); else store_all(std::false_type{});
        } else {
            // non-pooled: lane = pixel, register r = channel (r&3) + 8*(r>>2) + 4*half of the N-block -> 8-byte stores
            const int cs = p.out_cstride;
            int lane_off;
            _Float16* obase;
            bool full;
            if constexpr (TAPS == 1) {
                lane_off = li * cs + half * 4;
                obase = p.out + px0 * cs + p.out_coff + slice * 64;
                full = (px0 + 256 <= p.total_px) && (slice * 64 + 64 <= p.cout);
            } else {
                lane_off = ((li / MBW) * p.W + li % MBW) * cs + half * 4;
                obase = p.out + (((long long)img * p.H + y0) * p.W + x0) * cs + p.out_coff + slice * 64;
                full = (y0 + G::TH <= p.H) && (x0 + G::TW <= p.W) && (slice * 64 + 64 <= p.cout);
            }
            auto store_all = [&](auto full_tag) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int cl = nb * 32 + rg * 8 + half * 4;
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&prm[cl]);
                        const f32x4 s4 = *reinterpret_cast<const f32x4*>(&prm[64 + cl]);
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&prm[128 + cl]);
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            const h2 lo = act_h2<RELU, BNF>(acc[mb][nb][rg * 4], acc[mb][nb][rg * 4 + 1], f32x2{b4[0], b4[1]},
                                                            f32x2{s4[0], s4[1]}, f32x2{t4[0], t4[1]});
                            const h2 hi = act_h2<RELU, BNF>(acc[mb][nb][rg * 4 + 2], acc[mb][nb][rg * 4 + 3], f32x2{b4[2], b4[3]},
                                                            f32x2{s4[2], s4[3]}, f32x2{t4[2], t4[3]});
                            const h4 v = h4{lo[0], lo[1], hi[0], hi[1]};
                            // M-block (2*wave+mb): rows (2*wave+mb)*MBH.. of the tile, or 32 consecutive pixels in flat mode
                            _Float16* const mp = (TAPS == 1) ? obase + (long long)((2 * wave + mb) * 32) * cs
                                                             : obase + (long long)((2 * wave + mb) * G::MBH) * p.W * cs;
                            _Float16* const dst = mp + nb * 32 + rg * 8 + lane_off;
                            if constexpr (FULL) {
                                *reinterpret_cast<h4*>(dst) = v;
                            } else {
                                bool okp;
                                if constexpr (TAPS == 1) okp = px0 + (2 * wave + mb) * 32 + li < p.total_px;
                                else okp = (y0 + (2 * wave + mb) * G::MBH + li / MBW < p.H) & (x0 + li % MBW < p.W);
                                const int ch0 = slice * 64 + cl;
                                if (ch0 + 3 < p.cout || !okp) {
                                    *reinterpret_cast<h4*>(okp ? dst : p.dummy + lane * 4) = v;
                                } else {                     // partial channel quad (cout = 65: the 1x1 detector head only)
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (ch0 + e < p.cout) dst[e] = v[e];
                                }
                            }
                        }
                    }
            };
            if (TAPS == 9 || slice * 64 + 64 <= p.cout) {      // (3x3 layers: cout is a multiple of 64, launch_conv_f16 refuses anything else)
                // Per M-block (32 pixels): every lane writes its eight 8-byte channel quads into the wave's staging block
                // [pixel][granule ^ (pixel & 7)][8 halfs], then lane l reads granule l & 7 of pixels l >> 3, + 8, + 16, + 24 and stores 16 bytes:
                // eight lanes = one pixel's 64 channels = one 128-byte line (8-byte pieces 128 bytes apart cost 32 partial lines per store
                // instruction: measured 20 % of an un-pooled launch)
                _Float16* const stg = ostage + wave * 2048;
                int lq = lane;
                asm volatile("" : "+v"(lq));              // the addresses below are item-invariant: keep hipcc from holding them in registers through the MFMA loop
                const int wrow = (lq & 31) * 64 + (lq >> 5) * 4, wsw = lq & 7;
                const int rg_l = lq & 7, rp0 = lq >> 3;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int cl = nb * 32 + rg * 8 + half * 4;
                            const f32x4 b4 = *reinterpret_cast<const f32x4*>(&prm[cl]);
                            const f32x4 s4 = *reinterpret_cast<const f32x4*>(&prm[64 + cl]);
                            const f32x4 t4 = *reinterpret_cast<const f32x4*>(&prm[128 + cl]);
                            const h2 lo = act_h2<RELU, BNF>(acc[mb][nb][rg * 4], acc[mb][nb][rg * 4 + 1], f32x2{b4[0], b4[1]},
                                                            f32x2{s4[0], s4[1]}, f32x2{t4[0], t4[1]});
                            const h2 hi = act_h2<RELU, BNF>(acc[mb][nb][rg * 4 + 2], acc[mb][nb][rg * 4 + 3], f32x2{b4[2], b4[3]},
                                                            f32x2{s4[2], s4[3]}, f32x2{t4[2], t4[3]});
                            *reinterpret_cast<h4*>(stg + wrow + (((nb * 4 + rg) ^ wsw) << 3)) = h4{lo[0], lo[1], hi[0], hi[1]};
                        }
                    asm volatile("" ::: "memory");                                // (same wave: the LDS executes its operations in order)
                    _Float16* const mp = (TAPS == 1) ? obase + (long long)((2 * wave + mb) * 32) * cs
                                                     : obase + (long long)((2 * wave + mb) * G::MBH) * p.W * cs;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int px = rp0 + 8 * k;                          // (rp0 < 8: px / MBW and the k-part of px % MBW are uniform)
                        const h8 v = *reinterpret_cast<const h8*>(stg + px * 64 + ((rg_l ^ rp0) << 3));
                        _Float16* dst = (TAPS == 1) ? mp + (8 * k) * cs + (rp0 * cs + rg_l * 8)
                                                    : mp + (((8 * k) / MBW) * p.W + (8 * k) % MBW) * cs + (rp0 * cs + rg_l * 8);
                        if (!full) {
                            bool okp;
                            if constexpr (TAPS == 1) okp = px0 + (2 * wave + mb) * 32 + px < p.total_px;
                            else okp = (y0 + (2 * wave + mb) * G::MBH + px / MBW < p.H) & (x0 + px % MBW < p.W);
                            dst = okp ? dst : p.dummy + lane * 8;
                        }
                        *reinterpret_cast<h8*>(dst) = v;
                    }
                    asm volatile("" ::: "memory");
                }
            } else if constexpr (TAPS == 1) {
                store_all(std::false_type{});          // a partial channel slice (cout = 65: the 1x1 detector head as its own launch)
            }
        }
        MPH_T(t_epi1);
        MPH_ADD(4, t_epi0, t_epi1);                            // epilogue
#ifdef MP_TIMING
        tsum[7] += 1;
        if (!has_next && tid == 0 && t_on)
            for (int i = 0; i < 8; ++i) g_timing_h[blockIdx.x * 8 + i] = tsum[i];
#endif
        if (!has_next) return;
        __syncthreads();                                       // next item's LDS image complete, every epilogue done
        if (nxt.slice != cur.slice) load_prm(nxt.slice);       // (rare) visible to the next epilogue via the post-steps barrier
        MPH_T(t_bar2);
        MPH_ADD(5, t_epi1, t_bar2);
        item = item_next;
        cur = nxt;
        wp = wnext;
    }
}

template <int TAPS, int MBW, bool POOL>
int launch_h(const ConvParamsH& p, hipStream_t s)
{
    long long ntiles;
    if (TAPS == 9) ntiles = (long long)p.B * p.tiles_x * p.tiles_y;
    else ntiles = (p.total_px + 255) / 256;
    const long long nitems = ntiles * p.nslices;
    if (nitems <= 0) return 0;
    ConvParamsH q = p;
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); };
    q.magic_slices = magic(p.nslices); q.magic_tx = magic(p.tiles_x); q.magic_ty = magic(p.tiles_y);
    const long long dmax = std::max(std::max(p.nslices, p.tiles_x), p.tiles_y);
    if (nitems * dmax >= 0x100000000ll) return 1;      // beyond the 32-bit tile decode: reported as MP_EINVAL
    q.nitems = (int)nitems;
    // persistent workgroups: two per CU, a multiple of the XCD count
    const long long nblk = persistent_grid(nitems, p.ncu, p.xcd_shift, 2);
    const ConvParamsH& pp = q;
    if (p.bn_first)
        hipLaunchKernelGGL((conv_f16_kernel<TAPS, MBW, POOL, true>), dim3((unsigned)nblk), dim3(256), 0, s, pp);
    else
        hipLaunchKernelGGL((conv_f16_kernel<TAPS, MBW, POOL, false>), dim3((unsigned)nblk), dim3(256), 0, s, pp);
    return 0;
}

// ---- first encoder block, fp16 flavour: image (fp32 in HBM, rounded to fp16 as autocast casts the conv input)
//      -> 64 fp16 channels.  HBM-write bound: thread = (pixel, 8 channels), one 16-byte store.
constexpr int FTH = 32, FTW = 64, FLW = FTW + 2, FLH = FTH + 2;

__global__ __launch_bounds__(256) void conv_first_f16_kernel(const Conv1ParamsH p)
{
    __shared__ float tile[FLH * FLW];
    const int tid = threadIdx.x;
    const int tiles_x = (p.W + FTW - 1) / FTW, tiles_y = (p.H + FTH - 1) / FTH;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int bi = t / tiles_y;
    const int img = p.img_list ? p.img_list[bi] : bi;
    const int y0 = ty * FTH, x0 = tx * FTW;
    const float* in = p.in + (long long)img * p.H * p.W;
    for (int f = tid; f < FLH * FLW; f += 256) {
        const int ly = f / FLW, lx = f - ly * FLW;
        int gy = y0 + ly - 1, gx = x0 + lx - 1;
        float v;
        if (p.pad_zero) {
            const bool zero = (gy < 0) | (gy >= p.H) | (gx < 0) | (gx >= p.W);
            gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
            v = zero ? 0.f : in[gy * p.W + gx];
        } else {
            v = in[reflect_clamp_h(gy, p.H) * p.W + reflect_clamp_h(gx, p.W)];
        }
        tile[f] = round_h(v);
    }
    // channel pairs in f32x2 registers: the 72 multiply-adds and the activation of a pixel's 8 channels are packed
    // instructions (v_pk_fma_f32 with the pixel value broadcast, act_h2) -- the kernel is otherwise VALU-bound
    // (~180 scalar VALU instructions per 16-byte store against ~116 clocks of HBM time per KiB and CU)
    const int c8 = (tid & 7) * 8;
    f32x2 w[9][4], bia[4], scl[4], sft[4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) w[k][e] = f32x2{p.w[k * 64 + c8 + 2 * e], p.w[k * 64 + c8 + 2 * e + 1]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        bia[e] = f32x2{p.bias[c8 + 2 * e], p.bias[c8 + 2 * e + 1]};
        scl[e] = f32x2{p.scale[c8 + 2 * e], p.scale[c8 + 2 * e + 1]};
        sft[e] = f32x2{p.shift[c8 + 2 * e], p.shift[c8 + 2 * e + 1]};
    }
    __syncthreads();
    const int psub = tid >> 3;                    // 32 pixels per pass
    auto pixel = [&](int py, int px) __attribute__((always_inline)) -> h8 {
        float x[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = tile[(py + kh) * FLW + px + kw];
        h8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x2 a = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) a = __builtin_elementwise_fma(f32x2{x[k], x[k]}, w[k][e], a);
            const h2 r = p.bn_first ? act_h2<true, true>(a[0], a[1], bia[e], scl[e], sft[e])
                                    : act_h2<true, false>(a[0], a[1], bia[e], scl[e], sft[e]);
            o[2 * e] = r[0]; o[2 * e + 1] = r[1];
        }
        return o;
    };
    if (p.pool) {
        // double_convolution: false -- Conv -> ReLU -> BN -> MaxPool2d(2,2) under autocast: the maximum of four fp16 values (exact)
        const int Ho = p.H >> 1, Wo = p.W >> 1;
        _Float16* out = p.out + (long long)img * Ho * Wo * 64;
        for (int it = 0; it < (FTH * FTW) / 4 / 32; ++it) {
            const int pix = it * 32 + psub;
            const int py = pix / (FTW / 2), px = pix % (FTW / 2);
            const h8 a = pixel(2 * py, 2 * px), b = pixel(2 * py, 2 * px + 1), c = pixel(2 * py + 1, 2 * px), d = pixel(2 * py + 1, 2 * px + 1);
            h8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const _Float16 m0 = a[i] > b[i] ? a[i] : b[i], m1 = c[i] > d[i] ? c[i] : d[i];
                o[i] = m0 > m1 ? m0 : m1;
            }
            const int oy = (y0 >> 1) + py, ox = (x0 >> 1) + px;
            if (oy < Ho && ox < Wo) *reinterpret_cast<h8*>(out + ((long long)oy * Wo + ox) * 64 + c8) = o;
        }
        return;
    }
    _Float16* out = p.out + (long long)img * p.H * p.W * 64;
#pragma unroll 2
    for (int it = 0; it < (FTH * FTW) / 32; ++it) {
        const int pix = it * 32 + psub;
        const int py = pix / FTW, px = pix % FTW;
        const h8 o = pixel(py, px);
        const int oy = y0 + py, ox = x0 + px;
        if (oy < p.H && ox < p.W) __builtin_nontemporal_store(o, reinterpret_cast<h8*>(out + ((long long)oy * p.W + ox) * 64 + c8));   // streaming, as conv_first.hip
    }
}

}  // namespace

int launch_conv_f16(const ConvParamsH& p, int taps, int mbw, bool pool, hipStream_t s)
{
    if (taps == 1) return launch_h<1, 32, false>(p, s);
    if (p.cout != 64 * p.nslices) return 2;            // 3x3 layers: whole 64-channel slices (the model loader pads them)
    if (mbw == 32) return pool ? launch_h<9, 32, true>(p, s) : launch_h<9, 32, false>(p, s);
    if (mbw == 16) return pool ? launch_h<9, 16, true>(p, s) : launch_h<9, 16, false>(p, s);
    return pool ? launch_h<9, 8, true>(p, s) : launch_h<9, 8, false>(p, s);
}

void launch_conv_first_f16(const Conv1ParamsH& p, hipStream_t s)
{
    const int tiles_x = (p.W + FTW - 1) / FTW, tiles_y = (p.H + FTH - 1) / FTH;
    const long long nblk = (long long)p.B * tiles_x * tiles_y;
    if (nblk <= 0) return;
    hipLaunchKernelGGL(conv_first_f16_kernel, dim3((unsigned)nblk), dim3(256), 0, s, p);
}

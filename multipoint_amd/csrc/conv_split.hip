// Second half of the split small launches of the F(4x4,3x3) kernels (single-pair latency, B <= 2: conv_wino43.hip / conv_wino43b.hip
// with p.ks_shift > 0).  Those launches run the input channels of a (tile block, slice) item as 2^ks_shift separate items -- enough
// workgroups to fill the GPU -- and leave each range's PRE-BIAS output tiles in p.split_scratch.  This kernel, the next launch on the
// stream, sums the ranges' shares in range order (deterministic), adds the bias, applies ReLU / BatchNorm [and the 2x2 max-pool] and
// stores the result: one thread per (item, thread of the convolution workgroup, channel pair, pair of output rows), so the 8 values
// x up to 8 ranges a thread needs are ONE round trip of independent loads.  (Round 3 did the same inside the convolution launch --
// the last range to arrive, found through a device-scope counter, gathered the shares: 64 values x 8 ranges per thread are 8-16
// dependent ~1.5 us trips behind a write-through / counter / barrier hand-off, which was most of what a small launch cost.)
#include "mp_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float relu_s(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

// GEN 1: conv_wino43.hip (512 threads per item, 32 values each: [h][a][b]); GEN 2: conv_wino43b.hip (256 threads, 64 values:
// [m][h][a][b]).  The index arithmetic below is the epilogues' own (lane = tile, register quad = 4 consecutive output channels).
template <int GEN, bool POOL, bool BNF, int TC4>
__global__ __launch_bounds__(256) void split_reduce_kernel(const ConvParams p)
{
    constexpr int T = GEN == 1 ? 512 : 256;            // threads of a convolution workgroup
    constexpr int NV = GEN == 1 ? 32 : 64;             // shares (f32x2) per thread
    constexpr int NQ = NV / 8;                         // (m,) h, row pair
    constexpr int OY = 4 * (32 / TC4), OX = 4 * TC4;   // output pixels of an item
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int tid = (int)(gid % T), q = (int)((gid / T) % NQ);
    const int g = (int)(gid / (T * NQ));               // (tile block, slice)
    if (g >= p.nitems) return;
    auto udiv = [](unsigned n, unsigned magic, unsigned d) -> unsigned { return d == 1 ? n : __umulhi(n, magic); };
    const int tile = (int)udiv((unsigned)g, p.magic_slices, (unsigned)p.nslices);
    const int slice = g - tile * p.nslices;
    const int trow = (int)udiv((unsigned)tile, p.magic_tx, (unsigned)p.tiles_x);
    const int tx0 = tile - trow * p.tiles_x;
    const int bi = (int)udiv((unsigned)trow, p.magic_ty, (unsigned)p.tiles_y);
    const int ty0 = trow - bi * p.tiles_y;
    const int img = p.img_list ? p.img_list[bi] : bi;
    const int lane = tid & 63, wave = tid >> 6, tb = wave & 1;
    const int tl = tb * 16 + (lane & 15);              // tile of the item
    int t_ty, t_tx, cl, h, idx0;
    const int ap = q & 1;                              // output rows 2 ap, 2 ap + 1 of the 4 x 4 tile
    if (GEN == 1) {
        t_ty = tl / TC4; t_tx = tl % TC4;
        cl = (wave >> 1) * 16 + 4 * (lane >> 4);
        h = q >> 1; idx0 = h * 16;
    } else {
        t_ty = TC4 == 8 ? (tl >> 2) & 3 : tl >> 2; t_tx = TC4 == 8 ? 4 * (tl >> 4) + (tl & 3) : tl & 3;
        const int m = q >> 2;
        cl = (wave >> 1) * 32 + m * 16 + 4 * (lane >> 4);
        h = (q >> 1) & 1; idx0 = (m * 2 + h) * 16;
    }
    const int KS = 1 << p.ks_shift;
    const unsigned long long* const part = reinterpret_cast<const unsigned long long*>(p.split_scratch) +
                                           ((long long)g << p.ks_shift) * (NV * T) + (long long)(idx0 + 8 * ap) * T + tid;
    unsigned long long rawv[8][8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < KS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rawv[k][i] = part[(long long)k * (NV * T) + i * T];
        }
    const int ch = slice * 64 + cl + 2 * h;            // this thread's two output channels
    const f32x2 bb = {p.bias[ch], p.bias[ch + 1]}, ss = {p.scale[ch], p.scale[ch + 1]}, tt = {p.shift[ch], p.shift[ch + 1]};
    f32x2 yv[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x2 v = __builtin_bit_cast(f32x2, rawv[0][i]);
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < KS) v += __builtin_bit_cast(f32x2, rawv[k][i]);         // range order: deterministic
        v = w43_add_bias(v, 2 * ap + (i >> 2), i & 3, bb);          // (the shares are the SCALED transform's values: at6s, mp_common.h)
        if (BNF) { v = v * ss + tt; v = f32x2{relu_s(v[0]), relu_s(v[1])}; }
        else { v = f32x2{relu_s(v[0]), relu_s(v[1])}; v = v * ss + tt; }
        yv[i >> 2][i & 3] = v;
    }
    if (ch >= p.cout) return;
    const int oy = ty0 * OY + 4 * t_ty, ox = tx0 * OX + 4 * t_tx;
    const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
    const int cs = p.out_cstride;
    // NHWC: pixel stride cs floats; planar [B][cout/4][Ho][Wo][4]: the channel pair is floats 2h, 2h+1 of plane ch / 4
    float* const img_base = p.out_planar ? p.out + (long long)img * (p.cout / 4) * Ho * Wo * 4 + ((long long)(ch >> 2) * Ho * Wo) * 4 + (ch & 3)
                                         : p.out + (long long)img * Ho * Wo * cs + p.out_coff + ch;
    const int ps = p.out_planar ? 4 : cs;
    if (POOL) {
        const int py = (oy >> 1) + ap;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int px = (ox >> 1) + b;
            f32x2 r;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                r[c] = fmaxf(fmaxf(yv[0][2 * b][c], yv[0][2 * b + 1][c]), fmaxf(yv[1][2 * b][c], yv[1][2 * b + 1][c]));
            if (py < Ho && px < Wo) *reinterpret_cast<f32x2*>(img_base + ((long long)py * Wo + px) * ps) = r;
        }
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int py = oy + 2 * ap + a, px = ox + b;
                if (py < Ho && px < Wo) *reinterpret_cast<f32x2*>(img_base + ((long long)py * Wo + px) * ps) = yv[a][b];
            }
    }
}

template <int GEN, bool POOL, int TC4>
void launch_r(const ConvParams& q, hipStream_t s)
{
    const long long threads = (long long)q.nitems * (GEN == 1 ? 512 * 4 : 256 * 8);
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (q.bn_first) hipLaunchKernelGGL((split_reduce_kernel<GEN, POOL, true, TC4>), dim3(grid), dim3(256), 0, s, q);
    else hipLaunchKernelGGL((split_reduce_kernel<GEN, POOL, false, TC4>), dim3(grid), dim3(256), 0, s, q);
}

}  // namespace

// p: the layer's parameters as api.hip built them (REAL slices, ks_shift > 0); gen: 1 conv_wino43.hip, 2 conv_wino43b.hip.  The item
// shape is the one the convolution launchers pick (fewer items; 16 x 32 pixels on a tie)
int launch_split_reduce(const ConvParams& p, int gen, bool pool, hipStream_t s)
{
    const long long wide = (long long)((p.W + 31) / 32) * ((p.H + 15) / 16), tall = (long long)((p.W + 15) / 16) * ((p.H + 31) / 32);
    const int tc4 = tall < wide ? 4 : 8;
    const int OY = 4 * (32 / tc4), OX = 4 * tc4;
    ConvParams q = p;
    q.tiles_x = (p.W + OX - 1) / OX; q.tiles_y = (p.H + OY - 1) / OY;
    const long long nitems = (long long)p.B * q.tiles_x * q.tiles_y * p.nslices;
    if (nitems <= 0) return 0;
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); };
    q.magic_slices = magic(q.nslices); q.magic_tx = magic(q.tiles_x); q.magic_ty = magic(q.tiles_y);
    q.nitems = (int)nitems;
    if (gen == 1) {
        if (tc4 == 4) { if (pool) launch_r<1, true, 4>(q, s); else launch_r<1, false, 4>(q, s); }
        else { if (pool) launch_r<1, true, 8>(q, s); else launch_r<1, false, 8>(q, s); }
    } else {
        if (tc4 == 4) { if (pool) launch_r<2, true, 4>(q, s); else launch_r<2, false, 4>(q, s); }
        else { if (pool) launch_r<2, true, 8>(q, s); else launch_r<2, false, 8>(q, s); }
    }
    return 0;
}

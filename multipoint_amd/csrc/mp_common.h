// Shared declarations for the gfx950 (MI355X / CDNA4) kernels of libmultipoint_hip.so.
// Internal header: the public C ABI is include/multipoint_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MP_WAVE 64

// ---------------------------------------------------------------------------------------------
// conv (implicit GEMM on v_mfma_f32_32x32x2_f32)
// ---------------------------------------------------------------------------------------------
// Activations are NHWC fp32.  One workgroup (256 threads = 4 waves) produces a tile of 256 output
// pixels x 64 output channels; the 256 pixels are 8 "M-blocks" of 32 pixels, an M-block being
// (32/MBW) rows x MBW columns, stacked vertically:  tile = (256/MBW) rows x MBW cols.
// Wave w owns M-blocks 2w, 2w+1 and both 32-wide N-blocks of the 64-channel slice.
struct ConvParams {
    const float* in;      // [B][H][W][in_cstride], channels [in_coff, in_coff+cin) are read
    float* out;           // [B][Ho][Wo][out_cstride], channels [out_coff, out_coff+cout) written
    const float* wpack;   // packed weights, see pack_conv_weights() in api.hip
    const float* bias;    // [nslices*64] conv bias (zero padded)
    const float* scale;   // [nslices*64] BN scale  s = gamma / sqrt(var + eps)
    const float* shift;   // [nslices*64] BN shift  t = beta - mean * s
    const int* img_list;  // optional: image ids this launch processes (nullptr = 0..B-1)
    // fused first encoder block (FUSE1 variants): the image and the Cin=1 layer's parameters
    const float* img;     // [B][H][W]
    const float *w1, *b1, *s1, *t1;   // [9][64] tap-major weights, bias, BN scale, BN shift
    int B, H, W;          // conv input == conv output spatial size (stride 1, 'same' padding)
    int in_cstride, in_coff, cin;
    int out_cstride, out_coff, cout;
    int nslices;          // ceil(cout / 64)
    int tiles_x, tiles_y; // tiles per image
    int pad_zero;         // 0: reflection pad (ReflectionPad2d(1)), 1: zero pad (ZeroPad2d(1))
    int bn_first;         // 0: conv -> ReLU -> BN (reference default), 1: conv -> BN -> ReLU
    int relu;             // 0: no ReLU (final 1x1 convs)
    long long total_px;   // TAPS==1 (flat) mode: number of pixels
    unsigned magic_slices, magic_tx, magic_ty;   // multiply-high division constants (filled in by the launcher)
    int nitems;           // work items (tile, slice) of the launch (filled in by the launcher)
    int persist;          // 0: per-tile kernel only; n > 0: persistent workgroups for launches with >= n items per CU
    // conv_wino43.hip only: channel-quad-planar tensors [B][C/4][H][W][4] instead of NHWC (a unit of 4 input channels is then
    // contiguous row by row: its patch DMA touches ~10 cache lines per instruction instead of 64)
    int in_planar, out_planar;
    // machine shape (api.hip: mp_create derives it from the device, run_conv copies it into every launch): compute units of
    // the device = workgroups of a one-per-CU persistent grid, and log2 of its XCD count (workgroup b runs on XCD b mod nxcd;
    // each XCD has its own L2, so a persistent workgroup walks a contiguous share of ITS XCD's items)
    int ncu, xcd_shift;
    // conv_wino43.hip / conv_wino43b.hip, small launches (single-pair latency): the input channels of an item are cut into 2^ks_shift ranges that
    // run as separate items; their pre-bias output tiles meet in split_scratch, and split_reduce_kernel (conv_split.hip, the next
    // launch on the stream) sums them in range order and runs the rest of the epilogue.  ks_shift = 0: off.
    int ks_shift;
    float* split_scratch;     // [virtual item][32 or 64 values][512 or 256 threads] f32x2 (128 KiB per virtual item)
    // conv_wino43.hip, layers with many output slices (the 3x3 head convolutions): the input transform runs as a pass of its own
    // into vglobal [tile block][cin / 4][4][32][36] (conv_wino43_vglobal_floats()) and every slice's launch items DMA V from there
    // instead of transforming the same windows again.  nullptr: the transform runs inside the kernel, per slice.
    float* vglobal;
};

// persistent schedule shared by the persistent kernels: the items are cut into one contiguous range per XCD, and the
// workgroups of an XCD (blockIdx mod nxcd, gridDim a multiple of nxcd) walk that range with the stride of their number
struct XcdRange { int item, item_end, stride; };
__device__ __forceinline__ XcdRange xcd_range(int nitems, int xcd_shift)
{
    const int nxcd = 1 << xcd_shift;
    const int per_xcd = (nitems + nxcd - 1) >> xcd_shift;
    const int xcd = (int)blockIdx.x & (nxcd - 1);
    XcdRange r;
    r.stride = (int)gridDim.x >> xcd_shift;
    r.item_end = min((xcd + 1) * per_xcd, nitems);
    r.item = xcd * per_xcd + ((int)blockIdx.x >> xcd_shift);
    return r;
}
// host side: workgroups of a persistent launch with `per_cu` workgroups per CU -- never more than the items rounded up to a
// multiple of the XCD count (every XCD gets the same number of workgroups)
inline unsigned persistent_grid(long long nitems, int ncu, int xcd_shift, int per_cu = 1)
{
    const long long nxcd = 1ll << xcd_shift;
    const long long full = ((long long)ncu * per_cu) / nxcd * nxcd;
    const long long need = (nitems + nxcd - 1) / nxcd * nxcd;
    return (unsigned)(need < full ? need : full);
}

// first layer (Cin = 1, direct VALU conv, HBM-write bound)
struct Conv1Params {
    const float* in;      // [B][H][W]
    float* out;           // [B][H][W][channels]
    const float* w;       // [9][channels]  (tap-major, cout contiguous)
    const float* bias; const float* scale; const float* shift;   // [channels]
    const int* img_list;
    int B, H, W;
    int pad_zero, bn_first;
    int channels;         // output channels incl. zero padding: 64 (channel_version 0) or 32
    int out_planar;       // write [B][channels/4][H][W][4] (the consumer is conv_wino43.hip) instead of NHWC
    int pool;             // double_convolution: false (MultiPoint.py:147-148): MaxPool2d(2,2) follows the block -> out [B][H/2][W/2][channels]
};

// the conv launchers return 0, or 1 when the launch has more work items than the kernels' 32-bit magic-number tile decode
// can address (items * max divisor >= 2^32): nothing is launched and the caller reports MP_EINVAL
int launch_conv_mfma(const ConvParams& p, int taps, int mbw, bool pool, bool fuse1, hipStream_t s);
void launch_conv_first(const Conv1Params& p, hipStream_t s);
// Winograd F(4x4,3x3) (conv_wino43.hip): p.wpack = pack_wino43_weights() output; supports() says whether the shape is covered
// Interpolation points of the F(4x4,3x3) transforms: {0, +-a, +-b, inf}.  The textbook choice a = 1, b = 2 (Lavin & Gray) has
// integer transform matrices but the worst conditioning of the family: its fp32 error is ~20x that of a direct fp32
// convolution.  The same set scaled by 3/4 -- a = 3/4, b = 3/2 -- keeps every transform coefficient an exact binary
// fraction (a^2 = 9/16, b^2 = 9/4, a^2 b^2 = 81/64, a^2 + b^2 = 45/16), keeps the even/odd structure (12 multiply-adds per
// 1-D input transform, as before) and cuts the maximum error 3.4x and the rms error 2x (CPU emulation over a grid of dyadic
// (a, b): the minimum is broad around a b ~ 1, b / a ~ 2; docs/HISTORY.md section 4).  Shared by the kernel and the host-side
// weight transform U = G g G^T (api.hip).
#ifndef MP_W43_A
#define MP_W43_A 0.75
#endif
#ifndef MP_W43_B
#define MP_W43_B 1.5
#endif
// 1-D output transform A^T m (6 -> 4) of the F(4x4,3x3) kernels, packed over two output channels, with the powers of `a` factored
// out of rows 1 and 2: A^T[i][p] = p^i over the points (0, a, -a, b, -b, inf) and b = 2a give
//   y0 = m0 + s1 + s2,  y1 = a (d1 + 2 d2),  y2 = a^2 (s1 + 4 s2),  y3 = a^3 (d1 + 8 d2) + m5     (s = sums, d = differences)
// so z = (y0, y1 / a, y2 / a^2, y3) takes 10 packed instructions instead of 13 (every bracket is ONE multiply-add with an exact
// coefficient), and the factor sigma_r sigma_c, sigma = (1, a, a^2, 1), of output (r, c) goes into the multiply-add that adds the
// bias anyway (w43_out_scale; 1, a .. a^4 are exact binary fractions).  Two roundings fewer per y1 / y2 than the plain form.
typedef float mp_f32x2 __attribute__((ext_vector_type(2)));
static_assert(MP_W43_B == 2 * MP_W43_A, "at6s() needs b = 2a");
__device__ __forceinline__ void at6s(const mp_f32x2 m[6], mp_f32x2 z[4])
{
    constexpr float a3 = (float)(MP_W43_A * MP_W43_A * MP_W43_A);
    const mp_f32x2 s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
    z[0] = (m[0] + s1) + s2;
    z[1] = __builtin_elementwise_fma(d2, mp_f32x2{2.f, 2.f}, d1);
    z[2] = __builtin_elementwise_fma(s2, mp_f32x2{4.f, 4.f}, s1);
    z[3] = __builtin_elementwise_fma(__builtin_elementwise_fma(d2, mp_f32x2{8.f, 8.f}, d1), mp_f32x2{a3, a3}, m[5]);
}
constexpr float w43_sigma(int r) { return r == 1 ? (float)MP_W43_A : r == 2 ? (float)(MP_W43_A * MP_W43_A) : 1.f; }
constexpr float w43_out_scale(int r, int c) { return w43_sigma(r) * w43_sigma(c); }
// pre-bias output (r, c) of a tile from the scaled transform's value + the bias: ONE multiply-add (an add where the factor is 1)
__device__ __forceinline__ mp_f32x2 w43_add_bias(mp_f32x2 z, int r, int c, mp_f32x2 bias)
{
    const float k = w43_out_scale(r, c);
    return k == 1.f ? z + bias : __builtin_elementwise_fma(z, mp_f32x2{k, k}, bias);
}
bool conv_wino43_supports(const ConvParams& p);
long long conv_wino43_items(const ConvParams& p);      // work items the launch would have (B x tile blocks x slices)
long long conv_wino43_vglobal_floats(const ConvParams& p);      // size of ConvParams::vglobal for this layer
int launch_conv_wino43(const ConvParams& p, bool pool, hipStream_t s, bool fuse_first = false);
// second generation (conv_wino43b.hip): one wave per SIMD, whole-window input transform; any frame size
bool conv_wino43b_supports(const ConvParams& p);
int launch_conv_wino43b(const ConvParams& p, bool pool, hipStream_t s);
// the split small launches' second half (conv_split.hip): sums the ranges' shares, activation, [pool], store; gen 1 / 2 = which kernel wrote them
int launch_split_reduce(const ConvParams& p, int gen, bool pool, hipStream_t s);

// fp16 path (mixed_precision: activations and packed weights fp16, fp32 accumulate; conv_f16.hip).
// Same tiling as ConvParams; a chunk is 64 input channels, so cin must be a multiple of 64.
struct ConvParamsH {
    const _Float16* in;   // [B][H][W][in_cstride]
    _Float16* out;        // [B][Ho][Wo][out_cstride]
    const _Float16* wpack;
    const float* bias;    // fp16-representable values
    const float* scale;   // fp32 BN terms (BatchNorm runs in fp32 under autocast)
    const float* shift;
    const int* img_list;
    int B, H, W;
    int in_cstride, in_coff, cin;
    int out_cstride, out_coff, cout;
    int nslices, tiles_x, tiles_y;
    int pad_zero, bn_first;
    long long total_px;
    unsigned magic_slices, magic_tx, magic_ty;
    int nitems;           // work items (tile, slice) of the launch (filled in by the launcher)
    _Float16* dummy;      // >= 1 KiB scratch line that masked-off store lanes write to
    int ncu, xcd_shift;   // machine shape, as in ConvParams
    int res_groups;       // conv_f16_res.hip: independent four-wave groups per CU (3; 2 = MP_DEBUG=f16_res_groups=2; the fused-first-block launch always runs 2)
    // conv_f16_res.hip with the first encoder block fused in: the fp32 image [B][H][W] and the Cin = 1 layer's parameters
    // ([9][64] tap-major fp16-representable weights, bias, BN scale / shift); img == nullptr: p.in is read
    const float* img;
    const float *w1, *b1, *s1, *t1;
};
struct Conv1ParamsH {
    const float* in;      // [B][H][W] fp32 image (rounded to fp16 on load)
    _Float16* out;        // [B][H][W][64]
    const float* w;       // [9][64] fp16-representable values
    const float* bias; const float* scale; const float* shift;
    const int* img_list;
    int B, H, W;
    int pad_zero, bn_first;
    int pool;             // double_convolution: false (MultiPoint.py:147-148): MaxPool2d(2,2) follows the block -> out [B][H/2][W/2][64]
};
int launch_conv_f16(const ConvParamsH& p, int taps, int mbw, bool pool, hipStream_t s);
// 64 -> 64 3x3 layers with the packed weights resident in LDS (conv_f16_res.hip)
bool conv_f16_res_supports(const ConvParamsH& p, int taps);
int launch_conv_f16_res(const ConvParamsH& p, int mbw, bool pool, hipStream_t s);
void launch_conv_first_f16(const Conv1ParamsH& p, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// heads post-processing
// ---------------------------------------------------------------------------------------------
// logits [B*Hc*Wc][lstride] (65 valid) -> prob [B][Hc*8][Wc*8]  (softmax over 65, drop dustbin,
// depth-to-space 8) and/or logits_nchw [B][65][Hc][Wc]
// fused tail of both heads (head_tail.hip): 1x1 convolutions + BN + softmax/shuffle + L2 normalisation in one launch
struct HeadTailParams {
    const float* x;             // [npx][xstride] output of the 3x3 head convolution: detector channels [0,K), descriptor [K,2K)
    int xstride, K;             // K = head channels (multiple of 32)
    const float *wdet, *bdet, *sdet, *tdet;       // detector 1x1: pack_conv_weights(taps = 1) fragments, bias / BN scale / shift.
                                                  // The kernel reads entries [0, 96) of bdet / sdet / tdet (three 32-channel blocks for the 65
                                                  // detector channels) and [0, D) of the descriptor arrays: api.hip's build_conv pads every
                                                  // per-channel array to nslices * 64 >= 128 entries (zeros / identity BatchNorm)
    const float *wdesc, *bdesc, *sdesc, *tdesc;   // descriptor 1x1 (unused when desc == nullptr)
    int D;                      // descriptor size (64, 128 or 256)
    long long npx;              // B * Hc * Wc
    int B, Hc, Wc;
    float* prob;                // [B][1][8Hc][8Wc] or nullptr
    float* logits_nchw;         // [B][65][Hc][Wc] or nullptr
    float* desc;                // [npx][D] or nullptr
    int softmax_mode;           // 0: Softmax2d, 1: SuperPointMagicLeap heat map
    int normalize;              // F.normalize the descriptors
    int ncu;                    // compute units of the device: one persistent workgroup each
};
int launch_head_tail(const HeadTailParams& p, hipStream_t s);      // 0: launched, 1: shape not covered (use the separate kernels)
// the same for the fp16 path (head_tail_f16.hip): x and the weight fragments are fp16 (conv_f16.hip's packing, taps = 1), the
// biases are the fp16-rounded ones, outputs stay fp32
struct HeadTailParamsH {
    const _Float16* x;          // [npx][xstride] fp16 output of the 3x3 head convolution: detector channels [0,K), descriptor [K,2K)
    int xstride, K;             // K = head channels (multiple of 128: two 64-channel chunks per pass of the double buffer)
    const _Float16 *wdet, *wdesc;
    const float *bdet, *sdet, *tdet, *bdesc, *sdesc, *tdesc;      // padded like HeadTailParams' (>= 96 / D entries)
    int D;
    long long npx;
    int B, Hc, Wc;
    float* prob; float* logits_nchw; float* desc;
    int softmax_mode, normalize, ncu;
};
int launch_head_tail_f16(const HeadTailParamsH& p, hipStream_t s); // 0: launched, 1: shape not covered
void launch_det_post(const float* logits, int lstride, int B, int Hc, int Wc, float* prob,
                     float* logits_nchw, int mode, hipStream_t s);
void launch_det_post_f16(const _Float16* logits, int lstride, int B, int Hc, int Wc, float* prob,
                         float* logits_nchw, int mode, hipStream_t s);
// raw [npx][D] -> out [npx][D] rows divided by max(||row||, 1e-12)   (D multiple of 4, <= 1024)
void launch_desc_l2norm(const float* raw, float* out, long long npx, int D, int normalize,
                        hipStream_t s);
void launch_desc_l2norm_f16(const _Float16* raw, float* out, long long npx, int D, int normalize,
                            hipStream_t s);

// ---------------------------------------------------------------------------------------------
// keypoint extraction
// ---------------------------------------------------------------------------------------------
#define MP_NMS_MAX_R 15  // footprint radius supported by the NMS kernels (box size <= 16: a row of the footprint is one 32-bit mask)
struct NmsFootprint {
    int R;                                   // radius
    unsigned rowmask[2 * MP_NMS_MAX_R + 1];  // bit (dx+R) of rowmask[dy+R] set <=> IoU > thr
};

// work map encoding: > 0 undecided candidate (its score), 0 dead / not a candidate, < 0 kept (-score)
void launch_nms_init(const float* prob, const uint8_t* mask, float min_prob, float* work,
                     long long n, hipStream_t s);
// tie_pairs (optional, int [B]): footprint tie guard -- += candidates that died to kept neighbours within tie_eps of their score only
void launch_nms_round(float* work, int B, int H, int W, const NmsFootprint& fp, int* remaining,
                      int round, hipStream_t s, float tie_eps = 0.f, int* tie_pairs = nullptr);
// W: width of the work map (a multiple of 4), Ws <= W: row stride (true width) of prob / mask
void launch_nms_round0(const float* prob, const uint8_t* mask, float min_prob, float* work, int B, int H, int W,
                       const NmsFootprint& fp, int* remaining, hipStream_t s, int Ws, float tie_eps = 0.f, int* tie_pairs = nullptr);
void launch_nms_accumulate(int* remaining, int B, int H, int W, int round, int* total, hipStream_t s);
// per image: ordered (row-major) list of kept pixels, top-k selection by (score desc, index asc),
// outputs kp_yx [B][K][2] int32, kp_score [B][K], kp_count [B]; optional dense map prob_nms
// Wout <= W: row stride of prob_nms (the caller's true width when the work map's rows are padded to a multiple of 4); kp_yx are
// true coordinates either way.  tie_pairs / pairs_min: the footprint tie guard's per-image counts (nms.hip; read and reset here)
void launch_select_keypoints(const float* work, int B, int H, int W, int topk, int K,
                             int* list_idx, float* list_score, int list_cap, int* kp_yx,
                             float* kp_score, int* kp_count, float* prob_nms, int* seg_scratch, hipStream_t s,
                             float tie_eps = 0.f, int tie_min = 0, int* tie_state = nullptr, int Wout = 0,
                             int* tie_pairs = nullptr, int pairs_min = 0);
// top-k tie guard state (device ints): [0] flagged images since the last read, [1 + b] flag of image b of the latest call
#define MP_TIE_MAX_IMAGES 1000
// ints of device scratch (segment counts + list totals) launch_select_keypoints / launch_extract_threshold need
size_t keypoint_scratch_ints(int B, int H, int W);
// plain threshold extraction: (map > thr) -> row-major list (torch.nonzero semantics)
void launch_extract_threshold(const float* map, const unsigned char* mask, int B, int H, int W, float thr, int K, int* kp_yx,
                              float* kp_score, int* kp_count, int* seg_scratch, hipStream_t s);

// bilinear sampling (grid_sample align_corners=True, zeros) + L2 normalise.
// desc [B][Hc][Wc][D] channels-last; kp_yx [B][K][2]; out [B][K][D]
void launch_sample_desc(const float* desc, int B, int Hc, int Wc, int D, int H, int W,
                        const int* kp_yx, const int* kp_count, int K, float* out, hipStream_t s);

// mutual nearest neighbour matching of P pairs; rowbest/colbest: [P][K] packed scratch
constexpr int MATCH_SHARES = 2;     // column shares of the matcher's arg-min passes (sample_match.hip); rowbest / colbest hold one array per share
void launch_match_impl(const float* dA, const int* nA, const float* dB, const int* nB,
                       long long pair_stride, int count_stride, int P, int K, int D, float thr,
                       unsigned long long* rowbest, unsigned long long* colbest, int* match_idx,
                       float* match_dist, int* match_count, hipStream_t s);

// remaining get_matches modes (match_extra.hip): two nearest train rows per query (BFMatcher knnMatch / match without
// crossCheck), idx/dist [P][K][2]; all pairs closer than thr (ThresholdMatcher), list_count [P] pre-set to 0
void launch_match_knn2(const float* dA, const int* nA, const float* dB, const int* nB, long long pair_stride,
                       int count_stride, int P, int K, int D, int* idx, float* dist, hipStream_t s);
void launch_match_threshold(const float* dA, const int* nA, const float* dB, const int* nB, long long pair_stride,
                            int count_stride, int P, int K, int D, float thr, int capacity, int* list_ij,
                            float* list_dist, int* list_count, hipStream_t s);

// GPU-resident pair metrics (evaluation.py:287-328): hom [2P][9] double (slot 2p: optical->thermal ground-truth
// homography, 2p+1: its inverse), warped [2P][K][2] double scratch, inv_idx [P][K] scratch (pre-set to -1),
// tp [2P][K] (pre-set to 0), metrics [P][8] (pre-set to 0): n_gt_o, n_gt_t, matched_o, matched_t, N_o, N_t, n_matches
void launch_pair_metrics(const int* kp_yx, const int* kp_count, const int* match_idx, const double* hom, int P, int K,
                         int H, int W, float thr, double* warped, int* inv_idx, unsigned char* tp, int* metrics,
                         hipStream_t s);
// keypoint repeatability (evaluation.py:156-199): hom [2P][2][9] double (slot b: inverse of its own homography, then the
// other image's homography), warped [2P][K][2] int64 scratch, out [P][4] (pre-set to 0): count1, count2, N_thermal, N_optical
void launch_repeatability(const int* kp_yx, const int* kp_count, const double* hom, int P, int K, int H, int W, double thr,
                          long long* warped, int* out, hipStream_t s);
// batched RANSAC homography (predict_align_image_pair.py:205-216): best [P] scratch (pre-set to 0), H_out [P][9] double
// (x,y) optical -> (x,y) thermal, mask [P][K] uint8 per optical keypoint (pre-set to 0), n_inliers [P]
void launch_ransac_homography(const int* kp_yx, const int* kp_count, const int* match_idx, int P, int K, int T, double thr,
                              unsigned long long seed, unsigned long long* best, double* H_out, unsigned char* mask,
                              int* n_inliers, hipStream_t s);

// single-image detector metrics (detector_metrics.hip; evaluation.py:56-97): best [B][H][W] scratch, rec_count / n_gt [B]
// (all three pre-set to 0), records [B][H*W]
void launch_detector_metrics(const float* prob, const unsigned char* gt, int B, int H, int W, float zero_thr,
                             float distance_thr, unsigned long long* best, int* rec_index, float* rec_prob,
                             unsigned* rec_bits, int* rec_count, int* n_gt, hipStream_t s);

// homographic adaptation (homog_adapt.hip; reference multipoint/utils/homographies.py:38-189, 361-433).
// hom arrays are device double [n][9], row-major 3x3 acting on pixel (x, y, 1).
// warp: dst[n](x, y) = src[n % n_src] sampled at hom[n] * (x, y, 1); mode 0 bilinear / 1 nearest, padding 0 zeros / 1 reflection
void launch_warp_perspective(const float* src, int n_src, int H, int W, const double* hom, int n_out, int Ho, int Wo,
                             int mode, int padding, float* dst, hipStream_t s);
void launch_cv_warp_linear(const float* src, int n, int H, int W, const double* hom_inv, int border, float* dst,
                           hipStream_t s);
void launch_ha_valid_mask(const double* hom_inv, int G, int H, int W, int r, int mask_border, unsigned char* mask,
                          hipStream_t s);
void launch_ha_begin(const float* pa, const float* pb, long long n, int aggregation, float* prob, float* count,
                     hipStream_t s);
void launch_ha_accumulate(const float* pa, const float* pb, const unsigned char* mask, const double* hom, int G, int B,
                          int H, int W, int aggregation, float* prob, float* count, hipStream_t s);
void launch_ha_finalize(const float* prob, const float* count, long long n, int aggregation, float min_count, float* out,
                        hipStream_t s);
void launch_gaussian_filter(const float* in, int B, int H, int W, int k, const float* wgt, float* out, hipStream_t s);

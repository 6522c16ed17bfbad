/*
 * libmultipoint_hip.so -- C ABI of the MI355X (gfx950) implementation of the MultiPoint
 * inference hot path (ethz-asl/multipoint).
 *
 * The reference has no FFI: its boundary is Python (class MultiPoint + three free functions).
 * Each entry point below names the reference interface it replaces (paths relative to the
 * reference repository root); multipoint_amd/ binds them with ctypes (see INTEGRATION.md for the
 * stub a reference maintainer would add).
 *
 * Conventions
 *   - every function returns MP_OK (0) or a negative MP_E* code; mp_last_error() gives the text.
 *   - all tensor arguments are DEVICE pointers owned by the caller unless marked "host".
 *     The library never frees or reallocates caller memory; outputs are fully overwritten.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Kernels are enqueued on
 *     it; functions do not synchronise unless stated.
 *   - one handle per (process, device); a handle is not thread-safe.  The handle owns the packed
 *     device weights and one grow-only workspace.
 *   - images / probability maps are fp32 [B][H][W] (== NCHW with C = 1), H and W multiples of 8.
 *   - coarse descriptor maps are channels-last fp32 [B][H/8][W/8][D].
 *   - keypoint lists are int32 [B][K][2] as (y, x) in row-major order (torch.nonzero order),
 *     counts int32 [B].
 */
#ifndef MULTIPOINT_HIP_H
#define MULTIPOINT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MP_OK 0
#define MP_EINVAL (-1)      /* bad argument / unsupported configuration */
#define MP_EHIP (-2)        /* HIP runtime error */
#define MP_ESTATE (-3)      /* call order (e.g. forward before load_weights) */
#define MP_ENOMEM (-4)

typedef struct mp_handle mp_handle;

/* model: keys of MultiPoint.default_config (multipoint/models/MultiPoint.py:9-23) that change
 * the computation */
typedef struct mp_model_config {
    int multispectral;          /* two encoders routed by is_optical (MultiPoint.py:55-59,107-122) */
    int descriptor_head;
    int descriptor_size;        /* 64 (shipped params.yaml) | 128 | 256 */
    int normalize_descriptors;
    int final_batchnorm;
    int reflection_pad;         /* 1: ReflectionPad2d(1), 0: ZeroPad2d(1)  (MultiPoint.py:33-36) */
    int bn_first;               /* MultiPoint.py:137-141 */
    int double_convolution;     /* 1: two 3x3 convolutions per stage; 0: one (MultiPoint.py:144-148) */
    int channel_version;        /* 0: [1,64,64,128,128], heads 256; 1: [1,32,64,96,128]; 2: [1,8,16,32,64] (heads = descriptor_size) */
    /* model.type 'SuperPointMagicLeap' (multipoint/models/SuperPointMagicLeap.py): same layer shapes, no
     * BatchNorm, zero padding, state_dict keys conv1a..conv4b / convPa,convPb / convDa,convDb, heat map =
     * exp(x) / (sum exp(x) + 1e-5) without max subtraction (generate_heatmap, :68-85). */
    int batchnorm;              /* 1: BatchNorm2d after every 3x3 conv (MultiPoint), 0: none (MagicLeap) */
    int key_layout;             /* 0: MultiPoint nn.Sequential keys, 1: SuperPointMagicLeap keys */
    int softmax_mode;           /* 0: nn.Softmax2d, 1: MagicLeap generate_heatmap arithmetic */
    /* MultiPoint.py:21,99-103: forward under torch.cuda.amp.autocast.  1: fp16 activations and weights on the
     * fp16 MFMA (fp32 accumulate), BatchNorm / softmax / descriptor normalisation in fp32; inputs and outputs
     * of mp_forward stay fp32. */
    int mixed_precision;
    /* Algorithm of the 3x3 convolutions (no reference counterpart: ATen picks its own).  0 auto (= 1), 1 winograd43: Winograd
     * F(4x4,3x3) on the fp32 MFMA -- the fastest; equal to the fp32 reference within 2e-5 (prob) / 2e-6 (desc), which can reorder
     * EXACT ties of the heat map (flat or saturated image regions; a top-k cut inside a plateau of tied scores picks other members of
     * the plateau: parity.structured of bench.py); 2
     * winograd43_general: the same arithmetic on the any-frame-size kernel only; 3 direct: implicit-GEMM convolution, a k-ordered
     * fp32 multiply-add chain per output like the reference's -- keeps exact ties, 2.4x slower.  INTEGRATION.md has the numbers. */
    int conv_algorithm;
    /* 1: a forward's output bits do not depend on how many images it holds.  By default (0) forwards of ONE or TWO images (counted over
     * the whole forward, not per encoder of a multispectral model) --
     * the reference's shipped batchsize 1 -- run launches that are too small to fill the GPU with their input channels cut into
     * ranges (single-pair latency: 0.51 instead of 0.56 ms at 480x640, 0.33 instead of 0.44 ms at 240x320), which sums the same products in another order than the
     * batched launch does: equal within 3e-5 (prob) / 3e-6 (desc), deterministic from run to run, but not bit-identical to the
     * same images inside a larger batch.  Set it when shards of <= 2 images must reproduce a batched run bit for bit. */
    int batch_invariant;
} mp_model_config;

/* one entry of the reference state_dict (torch.save(net.state_dict()), train.py:161-173), host fp32 */
typedef struct mp_tensor {
    const char* name;           /* e.g. "encoder.5.weight", "detector_head_convolutions.5.running_var" */
    const float* data;          /* host pointer, contiguous, reference layout (conv: OIHW) */
    long long numel;
} mp_tensor;

/* lifetime ----------------------------------------------------------------------------------- */
int mp_create(mp_handle** out, int device);
void mp_destroy(mp_handle* h);
const char* mp_last_error(const mp_handle* h);      /* h may be NULL: error of the last mp_create */
const char* mp_version(void);
/* The machine shape the handle's persistent kernels are sized for, derived from the device in mp_create (compute units;
 * XCDs = L2 domains the work items are cut into; workgroups of a one-per-CU persistent launch).  No reference counterpart: the
 * reference leaves scheduling to ATen.  mp_create fails with MP_EINVAL on a shape the kernels cannot be scheduled on. */
int mp_device_shape(const mp_handle* h, int* compute_units, int* xcds, int* persistent_workgroups);

/* replaces MultiPoint.__init__ + load_state_dict (MultiPoint.py:25-91,
 * predict_align_image_pair.py:57-62): validates the key set strictly, repacks conv weights into
 * MFMA fragment order, precomputes eval-mode BatchNorm scale/shift, uploads. */
int mp_load_weights(mp_handle* h, const mp_model_config* cfg, const mp_tensor* tensors, int n_tensors);

/* replaces MultiPoint.forward / forward_impl in eval mode (MultiPoint.py:99-135).
 *   images      [B][H][W] fp32 in [0,1]
 *   is_optical  host uint8[B] or NULL (only read when cfg.multispectral)
 *   prob        [B][H][W] or NULL          (softmax -> drop dustbin -> PixelShuffle(8))
 *   logits      [B][65][H/8][W/8] or NULL  (force_return_logits path, MultiPoint.py:153-154)
 *   desc        [B][H/8][W/8][D] or NULL   (channels-last; L2-normalised if cfg says so) */
int mp_forward(mp_handle* h, const float* images, const unsigned char* is_optical, int B, int H,
               int W, float* prob, float* logits, float* desc, void* stream);

/* replaces utils.box_nms (multipoint/utils/utils.py:78-122) incl. the `prob * valid_mask`
 * multiply of its callers (predict_align_image_pair.py:128,133).
 *   valid_mask  uint8 [B][H][W] or NULL
 *   prob_nms    [B][H][W] dense output (zeros except kept pixels, which keep their score)
 *   max_rounds  0: run until converged (groups of 8 rounds, one 4-byte host read per group: synchronises
 *               `stream`; at most 4096 rounds); >0: enqueue exactly that many (<= 64) fixed-point rounds
 *               asynchronously, check with mp_nms_unresolved() after a sync.
 *   iou         DOUBLE: torchvision's CPU kernel (nms_kernel_impl(dets, scores, double iou_threshold)) compares the fp32 overlap
 *               ratio with the caller's Python float in double, and 0.1 is not 0.1f (they differ where ratio == (float)iou,
 *               e.g. size 11, iou 0.1, offset 9: 22 / 220).
 * Any H x W (the reference takes any 2-D / 4-D map, utils.py:90-91; W need not be a multiple of 4 here either).  Deviation:
 * box sizes above 16 are refused with MP_EINVAL (a footprint row is one 32-bit mask; the reference has no limit, its shipped
 * configs use 4).
 * Tie-break (stated rule): priority = (score descending, row-major index ascending). */
int mp_box_nms(mp_handle* h, const float* prob, const unsigned char* valid_mask, int B, int H, int W,
               float size, float min_prob, double iou, int keep_top_k, float* prob_nms,
               int max_rounds, void* stream);

/* fused box_nms + torch.nonzero(prob_nms > min_prob) (predict_align_image_pair.py:170-171,
 * evaluation.py:262-263): same NMS, but the survivors are returned as lists.
 *   kp_yx [B][K][2], kp_score [B][K] (may be NULL), kp_count [B]; kp_count may exceed K when
 *   keep_top_k == 0 and more than K pixels survive (only the first K are stored). */
int mp_detect_keypoints(mp_handle* h, const float* prob, const unsigned char* valid_mask, int B, int H,
                        int W, float size, float min_prob, double iou, int keep_top_k, int K,
                        int* kp_yx, float* kp_score, int* kp_count, int max_rounds, void* stream);

/* number of still-undecided NMS candidates summed over all mp_box_nms / mp_detect_keypoints calls since
 * the previous mp_nms_unresolved (0 = every result exact); resets the counter.  Synchronises `stream`. */
int mp_nms_unresolved(mp_handle* h, int* unresolved, void* stream);

/* Top-k tie guard (no reference counterpart; it exists because the reference's list is a function of EXACT score order,
 * multipoint/utils/utils.py:97-116: score-ordered indices, per-image [:keep_top_k]).  The default convolution algorithm
 * (Winograd F(4x4,3x3)) equals the fp32 reference within ~1e-5 in prob, which reorders exact ties of the heat map; when the
 * top-k cut of an image falls inside a plateau of (near-)tied scores, WHICH members of the plateau are kept is then decided by
 * that noise.  mp_box_nms / mp_detect_keypoints (keep_top_k > 0) therefore count, per image, the NMS survivors whose score lies
 * within `eps` (default 6e-5: the measured prob noise) of the k-th score -- admitted ones and cut-off ones separately -- and flag
 * the image when BOTH counts reach `min_each_side` (default 4; 0 switches the guard off).  A flagged image is one whose list the
 * caller should recompute from a forward with conv_algorithm 3 (direct), which keeps exact ties; the Python mirror does that in
 * PairPipeline.run_converged / utils.box_nms_tie_robust, the throughput entry only reports the count.
 * Two more conditions raise the same per-image flag:
 *   - the cut splits a run of EXACTLY equal scores (some admitted, some not), whatever their number;
 *   - footprint tie guard (also for keep_top_k == 0, the shipped configs' `topk: 0`): at least `min_pairs` (default 16; 0: off)
 *     of the image's NMS decisions were taken between scores within `eps` -- a candidate suppressed by kept neighbours that are
 *     ALL within eps of its own score, i.e. a suppression the noise could have turned around -- and they are at least 1 % of the
 *     image's survivors.  Heat maps of independent scores hold about one such pair per 1000 survivors (0-5 per 480x640 image at
 *     eps 6e-5, measured on the oracle's maps); a plateau of tied scores inside one footprint holds several per survivor.
 * What the guard does NOT cover (stated residual): fewer than min_each_side / min_pairs near-ties -- a single near-tied pair
 * straddling the cut or inside a footprint flips with the noise; those are the explained fp32 flips the parity tests bound
 * (<= 0.2 % of the keypoints, tests/test_gpu_e2e_parity.py).
 *   mp_topk_ambiguous: flags [B] (host ints, 0/1) of the LATEST mp_box_nms / mp_detect_keypoints call; *total = flagged
 *   images summed over all calls since the previous read (resets).  Synchronises `stream`. */
int mp_topk_tie_guard(mp_handle* h, float eps, int min_each_side);
int mp_nms_tie_guard(mp_handle* h, int min_pairs);
int mp_topk_ambiguous(mp_handle* h, int* flags, int B, int* total, void* stream);

/* replaces torch.nonzero(map > thr) on an arbitrary dense map; with valid_mask (uint8 [B][H][W] or NULL) it is
 * torch.nonzero((map > thr) * valid_mask) (multipoint/utils/evaluation.py:156-157, predict_keypoints.py:176-178).  Any H x W. */
int mp_extract_keypoints(mp_handle* h, const float* map, const unsigned char* valid_mask, int B, int H, int W, float thr,
                         int K, int* kp_yx, float* kp_score, int* kp_count, void* stream);

/* replaces utils.interpolate_descriptors (multipoint/utils/utils.py:159-167).
 *   desc [B][Hc][Wc][D] channels-last, out [B][K][D]; rows k >= kp_count[b] are written as zeros. */
int mp_sample_descriptors(mp_handle* h, const float* desc, int B, int Hc, int Wc, int D, int H, int W,
                          const int* kp_yx, const int* kp_count, int K, float* out, void* stream);

/* replaces utils.get_matches(..., 'nnmatcher' | 'bfmatcher' crossCheck=True)
 * (multipoint/utils/matching.py:4-33, NNMatcher.match :41-72) for P independent pairs.
 *   descA/descB: first pair's descriptors [K][D]; pair p at + p * pair_stride floats
 *   countA/countB: int32, pair p at [p * count_stride]
 *   threshold < 0 disables the distance test (plain mutual NN == crossCheck)
 *   match_idx [P][K]: train index of query i or -1; match_dist [P][K]; match_count [P] */
int mp_match_mutual_nn(mp_handle* h, const float* descA, const int* countA, const float* descB,
                       const int* countB, long long pair_stride, int count_stride, int P, int K,
                       int D, float threshold, int* match_idx, float* match_dist, int* match_count,
                       void* stream);

/* replaces the per-sample arithmetic of utils.compute_descriptor_metrics (multipoint/utils/evaluation.py:287-328) on
 * the device-resident lists the calls above produced, for P pairs stored interleaved (image slot 2p = optical,
 * 2p+1 = thermal):
 *   kp_yx [2P][K][2], kp_count [2P]; match_idx [P][K]: thermal index matched to optical keypoint i, or -1
 *   homography  device double [2P][9], row-major 3x3 acting on (x, y, 1): slot 2p = ground-truth optical->thermal
 *               homography h_t * inv(h_o) (evaluation.py:259), slot 2p+1 = its inverse (:288)
 *   metrics [P][8] int32: n_gt_optical, n_gt_thermal (:297-298), num_matched_optical, num_matched_thermal (:301-311),
 *               N_optical, N_thermal (warped keypoints inside the H x W image, :315-316), number of matches, 0
 *   tp [2P][K] uint8: tp[2p][i] = match of optical keypoint i is correct; tp[2p+1][j] = match of thermal keypoint j
 *               (the same mutual pair seen from the other side) is correct; 0 for unmatched keypoints */
int mp_pair_metrics(mp_handle* h, const int* kp_yx, const int* kp_count, const int* match_idx, const double* homography,
                    int P, int K, int H, int W, float threshold_keypoints, int* metrics, unsigned char* tp,
                    void* stream);

/* replaces the per-sample arithmetic of utils.compute_repeatability_multispectral (multipoint/utils/evaluation.py:156-199)
 * for P pairs stored interleaved (slot 2p = optical, 2p+1 = thermal):
 *   homography  device double [2P][2][9]: for slot b first the INVERSE of its own homography, then the other image's
 *               homography (evaluation.py:168-169,173-174); both warps truncate to integers like warp_keypoints' default
 *   counts [P][4] int32: count1 (warped thermal points with an optical keypoint within distance_thresh), count2 (warped
 *               optical points near a thermal keypoint), N_thermal, N_optical (warped points inside the H x W frame);
 *               repeatability = (count1 + count2) / (N_thermal + N_optical)  (:198-199) */
int mp_repeatability(mp_handle* h, const int* kp_yx, const int* kp_count, const double* homography, int P, int K, int H,
                     int W, double distance_thresh, int* counts, void* stream);

/* replaces cv2.findHomography(optical_pts, thermal_pts, cv2.RANSAC, ransacReprojThreshold) on the matched keypoints
 * (predict_align_image_pair.py:205-216, multipoint/utils/evaluation.py:330-349) for P pairs (slot 2p optical, 2p+1
 * thermal).  Not a bit-level restatement of OpenCV (absent third-party code with its own RNG): max_iters hypotheses
 * from 4 random matches each (counter-based RNG on `seed`: reproducible), forward reprojection error test, most
 * inliers wins, normalised-DLT refit over the winner's inliers.
 *   homography   device double [P][9], row-major, maps optical (x, y, 1) to thermal; all zeros when < 4 matches or
 *                no hypothesis found 4 inliers (the reference's `H_est is None`)
 *   inlier_mask  uint8 [P][K] per OPTICAL keypoint (1 = its match is an inlier); n_inliers int32 [P] */
int mp_find_homography(mp_handle* h, const int* kp_yx, const int* kp_count, const int* match_idx, int P, int K,
                       double reproj_threshold, int max_iters, unsigned long long seed, double* homography,
                       unsigned char* inlier_mask, int* n_inliers, void* stream);

/* the other modes of utils.get_matches (multipoint/utils/matching.py:4-33); same descriptor / count addressing as
 * mp_match_mutual_nn, 1 <= D <= 256.
 * mp_match_knn2 replaces cv2.BFMatcher(cv2.NORM_L2).knnMatch(d1, d2, 2) (:21, followed by Lowe's ratio test :23-27)
 * and .match() without crossCheck (:7,31): nn_idx / nn_dist [P][K][2] = the two nearest train rows of every query
 * row under ||a - b||_2 (ties: lower train index first), idx -1 where the pair has fewer than 1 / 2 train rows.
 * mp_match_threshold replaces ThresholdMatcher.match (:81-99): every (i, j) with sqrt(2 - 2 clip(a.b, -1, 1)) <
 * threshold is appended (arbitrary order) to list_ij [P][capacity][2] / list_dist [P][capacity]; list_count [P] is
 * the number FOUND (may exceed capacity: the caller retries with a larger list). */
int mp_match_knn2(mp_handle* h, const float* descA, const int* countA, const float* descB, const int* countB,
                  long long pair_stride, int count_stride, int P, int K, int D, int* nn_idx, float* nn_dist,
                  void* stream);
int mp_match_threshold(mp_handle* h, const float* descA, const int* countA, const float* descB, const int* countB,
                       long long pair_stride, int count_stride, int P, int K, int D, float threshold, int capacity,
                       int* list_ij, float* list_dist, int* list_count, void* stream);

/* ---- single-image detector metrics: multipoint/utils/evaluation.py:10-97 (predict_keypoints.py:88-104) ----
 * mp_detector_metrics replaces compute_tp_fp_dist (evaluation.py:56-97) for B heat maps at once:
 *   prob          fp32 [B][H][W]  detector map after valid mask / NMS (evaluation.py:19-25)
 *   keypoint_map  uint8 [B][H][W] ground-truth label map (nonzero = keypoint; ImagePairDataset 'keypoints')
 * predictions = pixels with prob > zero_threshold (reference default 1e-4), ranked by (prob desc, flat index asc);
 * each prediction names the first ground-truth point in row-major order within distance_thresh (reference default
 * 2.0; must be < 3) and is a true positive iff it is the best-ranked prediction naming that point (the closed form
 * of the reference's greedy loop, :84-93).  Outputs, one record per prediction in arbitrary order:
 *   rec_index int32 [B][H*W] flat pixel index y*W+x;  rec_prob fp32 [B][H*W];
 *   rec_bits uint32 [B][H*W]: bit (dy+2)*5+(dx+2) set for every ground-truth point at offset (dy, dx) within
 *             distance_thresh (the entries of `dist[matches]`, :97), bit 31 = true positive;
 *   rec_count int32 [B] predictions per image;  n_gt int32 [B] ground-truth points per image (`len(kp)`);
 *   work uint64 [B][H][W] scratch. */
int mp_detector_metrics(mp_handle* h, const float* prob, const unsigned char* keypoint_map, int B, int H, int W,
                        float zero_threshold, float distance_thresh, unsigned long long* work, int* rec_index,
                        float* rec_prob, unsigned int* rec_bits, int* rec_count, int* n_gt, void* stream);

/* ---- homographic adaptation: multipoint/utils/homographies.py:38-189, export_keypoints.py:64-103 ----
 * Homographies are device double [n][9], row-major 3x3 acting on pixel coordinates (x, y, 1).
 *
 * mp_warp_perspective replaces WarpingModule / warp_perspective_tensor (homographies.py:404-433; kornia's
 * homography_warp = F.grid_sample(align_corners=True) on the homography in normalised coordinates): single-channel
 * maps src [n_src][H][W] -> dst [n_out][Ho][Wo],  dst[n](x, y) = src[n % n_src] sampled at dst_to_src[n] * (x, y, 1)
 * (dst_to_src = inverse of the M the reference passes); mode 0 bilinear / 1 nearest (half to even);
 * padding 0 zeros / 1 reflection (about pixel centres 0 and size-1). */
int mp_warp_perspective(mp_handle* h, const float* src, int n_src, int H, int W, const double* dst_to_src, int n_out,
                        int Ho, int Wo, int mode, int padding, float* dst, void* stream);

/* replaces cv2.warpPerspective(image, M, (W, H), borderMode=...) with the default INTER_LINEAR, as the dataset's
 * homographic augmentation calls it (multipoint/datasets/augmentation/augmentation.py:33-36): source coordinates in
 * 1/32-pixel fixed point (OpenCV's INTER_BITS = 5), float32 bilinear weight table, block-wise float64 coordinate
 * arithmetic.  src/dst fp32 [n][H][W] (dst != src); hom_inv device double [n][9] = inverse of the M the reference
 * passes (cv2 inverts it first); border 0 = BORDER_CONSTANT (0), 1 = BORDER_REFLECT_101. */
int mp_warp_perspective_cv(mp_handle* h, const float* src, int n, int H, int W, const double* hom_inv, int border,
                           float* dst, void* stream);

/* replaces compute_valid_mask (homographies.py:361-389) for G homographies: cv2.warpPerspective(ones, M, INTER_NEAREST)
 * (1 where the rounded source pixel hom_inv * (x, y, 1) lies in the frame) followed by cv2.erode with a
 * (2*erosion_radius+1)^2 box (erosion_radius <= 16); mask_border != 0 also erodes from the image border.
 * mask: uint8 [G][H][W]. */
int mp_ha_valid_mask(mp_handle* h, const double* hom_inv, int G, int H, int W, int erosion_radius, int mask_border,
                     unsigned char* mask, void* stream);

/* aggregation state of homographic_adaptation(_multispectral): prob, count fp32 [B][H][W].
 * aggregation 0: one map (prob_b NULL); 1 'prod' / 2 'sum' of the optical and thermal maps (homographies.py:63-68).
 *   mp_ha_begin       prob = map(s) of the un-warped images, count = 1                            (:60-68, :149-150)
 *   mp_ha_accumulate  for g < G in order: cs = nearest(mask[g], hom[g]); count += cs;
 *                     prob += bilinear(map[g][b], hom[g]) * cs   (zeros padding)                   (:111-113, :178-180)
 *                     prob_a / prob_b: [G][B][H][W] heat maps of the images warped by hom[g]
 *   mp_ha_finalize    out = prob / count; sqrt (prod) or * 0.5 (sum); 0 where count < min_count    (:115-127, :182-187) */
int mp_ha_begin(mp_handle* h, const float* prob_a, const float* prob_b, int B, int H, int W, int aggregation,
                float* prob, float* count, void* stream);
int mp_ha_accumulate(mp_handle* h, const float* prob_a, const float* prob_b, const unsigned char* mask,
                     const double* hom, int G, int B, int H, int W, int aggregation, float* prob, float* count,
                     void* stream);
int mp_ha_finalize(mp_handle* h, const float* prob, const float* count, int B, int H, int W, int aggregation,
                   float min_count, float* out, void* stream);

/* replaces filter(pad(prob)) (homographies.py:55-58: ReflectionPad2d((k-1)/2) + the k x k depthwise filter of
 * utils.get_gaussian_filter, utils.py:124-160): in/out fp32 [B][H][W], weights device fp32 [k][k], k odd, k <= 31. */
int mp_gaussian_filter(mp_handle* h, const float* in, int B, int H, int W, int ksize, const float* weights, float* out,
                       void* stream);

/* per-launch timing of mp_forward with hipEvents on the caller's stream (bench.py roofline leg).
 * mp_profile_read synchronises; names[i] points to static strings. */
int mp_profile_enable(mp_handle* h, int enable);
int mp_profile_read(mp_handle* h, const char** names, float* ms, double* flop, int capacity, int* n);

#ifdef __cplusplus
}
#endif
#endif /* MULTIPOINT_HIP_H */

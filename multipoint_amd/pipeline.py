"""Batched detect + describe + match driver: the per-batch loop head of
utils.compute_descriptor_metrics (reference multipoint/utils/evaluation.py:224-285) re-shaped for the
GPU.  The reference runs two forwards (optical, thermal), two box_nms calls and a Python loop over
samples with three cv2 matcher calls each; here both images of all pairs go through ONE forward as an
interleaved batch (image 2p = optical, 2p+1 = thermal), keypoints stay on the device as fixed-capacity
lists, and all pairs are matched by one launch.  Nothing synchronises until results are read."""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .utils import utils as U


class PairResults:
    def __init__(self, kp_yx, kp_score, kp_count, desc, match_idx, match_dist, match_count, H, W):
        self.kp_yx, self.kp_score, self.kp_count = kp_yx, kp_score, kp_count
        self.desc = desc
        self.match_idx, self.match_dist, self.match_count = match_idx, match_dist, match_count
        self.H, self.W = H, W
        self.done = None            # event recorded on the stream that produced these tensors
        self.inputs_consumed = None # event: the forward has read the input images (they may be overwritten after it)
        self.stage_events = None    # run_interleaved(timings=True): (forward begin, forward end, post begin, keypoints done, matches done)

    def stage_ms(self):
        """Milliseconds this batch spent per stage (synchronises): {'forward', 'nms' (box-NMS + top-k + keypoint lists),
        'match' (descriptor sampling + mutual-NN matching)} -- the t_forward / t_nms / t_match of the gathered per-pair record
        (multipoint_amd.dist.pair_metric_records).  None unless the batch ran with `timings=True`."""
        if self.stage_events is None:
            return None
        f0, f1, p0, p1, p2 = self.stage_events
        p2.synchronize()
        return {'forward': f0.elapsed_time(f1), 'nms': p0.elapsed_time(p1), 'match': p1.elapsed_time(p2)}

    def wait(self):
        """Make the current stream wait for the post-processing stream that produced the results."""
        if self.done is not None:
            torch.cuda.current_stream(self.kp_yx.device).wait_event(self.done)
        return self

    @property
    def num_pairs(self):
        return self.match_idx.shape[0]

    def to_host(self):
        """Per-pair python lists (synchronises)."""
        self.wait()
        kp = self.kp_yx.cpu().numpy(); cnt = self.kp_count.cpu().numpy()
        K = kp.shape[1]
        mi = self.match_idx.cpu().numpy(); md = self.match_dist.cpu().numpy()
        desc = self.desc.cpu().numpy()
        out = []
        for p in range(self.num_pairs):
            no, nt = min(int(cnt[2 * p]), K), min(int(cnt[2 * p + 1]), K)
            q = np.nonzero(mi[p, :no] >= 0)[0]
            out.append(dict(kp_optical=kp[2 * p, :no].astype(np.int64), kp_thermal=kp[2 * p + 1, :nt].astype(np.int64),
                            desc_optical=desc[2 * p, :no], desc_thermal=desc[2 * p + 1, :nt],
                            match_query=q.astype(np.int64), match_train=mi[p, q].astype(np.int64),
                            match_dist=md[p, q]))
        return out


class PairPipeline:
    """config: the `prediction:` block of the reference yaml
    (configs/config_image_pair_dataset_prediction.yaml:40-53)."""

    def __init__(self, net, config, capacity=None, nms_rounds=8, overlap_post=True, tie_robust=True, keep_maps=False):
        self.net = net
        self.tie_robust = tie_robust    # run_converged re-evaluates images flagged by the top-k tie guard with the tie-exact algorithm
        self.keep_maps = keep_maps      # run_converged keeps the (possibly redone) heat map / coarse descriptors on its results
        self.tie_redone = 0             # images the latest run_converged() redid; tie_redone_total: since construction
        self.tie_redone_total = 0
        self.tie_flagged = 0            # images flagged since the previous check_converged() (throughput entry: reported only)
        self.overlap_post = overlap_post and _lib.debug_switch('post_overlap', '1') != '0'     # (developer A/B switch: MP_DEBUG=post_overlap=0)
        self._post_stream = None
        self._fwd_stream = None
        self._last = None           # results of the latest run_interleaved() (check_converged inspects their counts)
        self.nms = config.get('nms', 4)
        self.thr = config.get('detection_threshold', 0.015)
        self.topk = config.get('topk', 0)
        self.capacity = capacity
        self.nms_rounds = nms_rounds
        m = config.get('matching', {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True},
                                    'knn_matches': False})
        # the batched pipeline keeps ONE match per optical keypoint (match_idx [P][K]): the mutual-NN matchers.  The other
        # get_matches modes (ratio test, one-directional nearest, thresholdmatcher) run per pair through
        # utils.get_matches (mp_match_knn2 / mp_match_threshold), as utils.compute_descriptor_metrics does.
        if m.get('knn_matches', False):
            raise NotImplementedError('PairPipeline batches the mutual-NN matchers only; use utils.get_matches for knn_matches')
        if m['method'] == 'bfmatcher':
            if not m.get('method_kwargs', {}).get('crossCheck', False):
                raise NotImplementedError('PairPipeline batches the mutual-NN matchers only; use utils.get_matches for '
                                          'bfmatcher without crossCheck')
            self.match_threshold = -1.0
        elif m['method'] == 'nnmatcher':
            self.match_threshold = float(m.get('method_kwargs', {}).get('threshold', 0.7))
            if self.match_threshold < 0:
                raise ValueError('\'threshold\' should be non-negative')
        elif m['method'] in ('thresholdmatcher', 'flann'):
            raise NotImplementedError("PairPipeline batches the mutual-NN matchers only; use utils.get_matches for '%s'"
                                      % m['method'])
        else:
            raise ValueError('unknown matching method')

    def set_tie_guard(self, device=None, eps=6e-5, min_each_side=4, min_pairs=16):
        """Sensitivity of the tie guards run_converged / check_converged read (include/multipoint_hip.h): `eps` = the score window
        (default: the default convolution algorithm's measured prob noise), `min_each_side` = survivors within eps of the k-th score
        needed on EACH side of the top-k cut (0: that guard off; the split of an exact tie is flagged whatever the number),
        `min_pairs` = NMS decisions between near-tied scores needed (and >= 1 % of the image's survivors; 0: the footprint guard off).
        Lower thresholds redo more images with the tie-exact algorithm; what stays below them is decided by fp32 rounding.  The
        setting belongs to the device's shared post-processing handle, i.e. to every pipeline on that device."""
        U.topk_tie_guard(device, eps, min_each_side)
        U.nms_tie_guard(device, min_pairs)

    @staticmethod
    def interleave(optical, thermal):
        """(P,1,H,W) x 2 -> (2P,1,H,W) with image 2p = optical[p], 2p+1 = thermal[p]."""
        P = optical.shape[0]
        return torch.stack((optical, thermal), dim=1).reshape(2 * P, *optical.shape[1:])

    def run_interleaved(self, images, valid_mask=None, is_optical=None, order_caller=True, timings=False):
        """One batch: the forward on a high-priority stream of the pipeline's own, then NMS / top-k / sampling / matching
        on a second side stream (`overlap_post=True`): those kernels are small and latency-bound, so they run in the
        shadow of the NEXT batch's convolutions instead of serialising behind this one.  The returned tensors belong to
        the side stream: `PairResults.wait()` (or a device synchronise) orders them.  The CALLER's stream is ordered
        behind the forward (not behind the post-processing): work it enqueues after this call -- overwriting `images` in
        place, for one -- cannot overtake the forward's reads; `PairResults.inputs_consumed` is the same event for
        callers that write the inputs from another stream (bench.py --host-input).  `order_caller=False` leaves that wait out
        for callers that never touch `images` again (or wait on `inputs_consumed` themselves): the caller's stream then stays
        idle, and the NEXT call need not order the forward stream behind it -- a cross-queue dependency that costs ~40 us of
        idle GPU between two forwards (`rocprofv3 --kernel-trace`: 55 instead of 13 us between the head tail of batch n and the
        first convolution of batch n+1).  `timings=True` brackets the stages with timing events (`PairResults.stage_ms()`); off by
        default -- a timing event is a packet of its own in the queue."""
        dev = images.device
        B, _, H, W = images.shape
        if B % 2:
            raise ValueError('interleaved batch must hold an even number of images')
        if is_optical is None:
            is_optical = (torch.arange(B) % 2 == 0).reshape(B, 1)
        main = torch.cuda.current_stream(dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if timings else None
        if self.overlap_post:
            # Two streams of the pipeline's own: the forward on a HIGH-priority one, the post-processing on a normal one.  A
            # convolution workgroup needs a whole CU; when batch n's post-processing and batch n+1's first convolution become
            # ready together, the dispatcher must hand the CUs to the convolution first -- otherwise the chain of ~20 small
            # dependent post-processing kernels keeps landing on freed CUs and the convolution's last workgroups start up to
            # 0.4 ms late (measured: enc.conv1+2 4.35 instead of 4.05 ms; docs/HISTORY.md section 7).
            if self._post_stream is None or self._post_stream.device != dev:
                self._post_stream = torch.cuda.Stream(device=dev, priority=int(_lib.debug_switch("post_priority", "0")))
                self._fwd_stream = torch.cuda.Stream(device=dev, priority=int(_lib.debug_switch("fwd_priority", "-1")))
            post, fwd = self._post_stream, self._fwd_stream
            if not main.query():                                    # the inputs were produced on the caller's stream: wait for whatever
                fwd.wait_stream(main)                               # is still pending there (nothing, in a steady pipeline: no packet)
            with torch.cuda.stream(fwd):
                if timings:
                    ev[0].record(fwd)
                out = self.net({'image': images, 'is_optical': is_optical})
                consumed = ev[1] if timings else torch.cuda.Event()
                consumed.record(fwd)
            if order_caller:
                main.wait_event(consumed)                           # the caller's stream stays ordered behind the forward
            for t in (images, valid_mask, is_optical if is_optical.is_cuda else None):
                if t is not None:
                    t.record_stream(fwd)
            post.wait_stream(fwd)      # (a forward the caller enqueues next on ITS stream is ordered behind this one by the model itself)
            for t in (out['prob'], out['desc'], valid_mask):
                if t is not None:
                    t.record_stream(post)
        else:
            if timings:
                ev[0].record(main)
            out = self.net({'image': images, 'is_optical': is_optical})
            post = main
            consumed = ev[1] if timings else torch.cuda.Event()
            consumed.record(main)
        with torch.cuda.stream(post):
            if timings:
                ev[2].record(post)
            res = self._post(out, valid_mask, dev, B, H, W, mark=(lambda: ev[3].record(post)) if timings else None)
            res.done = ev[4] if timings else torch.cuda.Event()
            res.done.record(post)
        if timings:
            res.stage_events = tuple(ev)
        res.inputs_consumed = consumed
        self._last = res
        return res

    def _capacity_is_exact(self, K):
        """A list of K slots holds every keypoint the reference would return iff top-k limits them to <= K."""
        return self.nms > 0 and 0 < self.topk <= K

    def _post(self, out, valid_mask, dev, B, H, W, nms_rounds=None, capacity=None, mark=None):
        prob = out['prob']
        if self.nms > 0:
            # `topk: 0` (the shipped configs, like the reference's) = unlimited: 4096 slots is a first guess that
            # run_converged() grows on overflow and check_converged() reports (the reference keeps every keypoint)
            K = capacity or self.capacity or (self.topk if self.topk > 0 else 4096)
            kp, sc, cnt = U.detect_keypoints(prob, self.nms, self.thr, keep_top_k=self.topk, capacity=K,
                                             valid_mask=valid_mask,
                                             max_rounds=self.nms_rounds if nms_rounds is None else nms_rounds)
        else:
            kp, sc, cnt = U.extract_keypoints(prob, self.thr, capacity=capacity or self.capacity or 4096, valid_mask=valid_mask)
        if mark is not None:
            mark()                  # keypoint lists done (stage timing)
        K = kp.shape[1]
        desc = U.interpolate_descriptors_batched(kp, cnt, out['desc'], H, W)        # [B,K,D]
        D = desc.shape[2]
        P = B // 2
        midx = torch.empty((P, K), dtype=torch.int32, device=dev)
        mdist = torch.empty((P, K), dtype=torch.float32, device=dev)
        mcnt = torch.empty((P,), dtype=torch.int32, device=dev)
        h = _lib.get_handle(dev)
        with torch.cuda.device(dev):
            h.check(h.lib.mp_match_mutual_nn(
                h.ptr, _lib.ptr(desc), _lib.ptr(cnt), ctypes.c_void_p(desc.data_ptr() + K * D * 4),
                ctypes.c_void_p(cnt.data_ptr() + 4), 2 * K * D, 2, P, K, D, float(self.match_threshold),
                _lib.ptr(midx), _lib.ptr(mdist), _lib.ptr(mcnt), _lib.stream_ptr(dev)))
        return PairResults(kp, sc, cnt, desc, midx, mdist, mcnt, H, W)

    def __call__(self, optical, thermal, mask_optical=None, mask_thermal=None):
        images = self.interleave(optical, thermal)
        mask = None
        if mask_optical is not None or mask_thermal is not None:
            ones = torch.ones_like(optical, dtype=torch.bool)
            mask = self.interleave(mask_optical if mask_optical is not None else ones,
                                   mask_thermal if mask_thermal is not None else ones)
        # convenience entry of the dataset drivers: results are ordered behind the caller's stream and the NMS is
        # exact whatever the heat map looks like (run_interleaved() is the high-throughput entry that leaves the
        # results on the side stream and reports non-convergence through check_converged())
        return self.run_converged(images, mask)

    def run_converged(self, images, valid_mask=None, is_optical=None):
        """One batch with a guaranteed-exact NMS, for the dataset drivers (evaluation loops): the fixed number of
        asynchronous rounds first; if that left candidates undecided (chains of dependent decisions longer than
        `nms_rounds` tiles -- smooth ramps across the frame), the post-processing of THIS batch is redone with the
        synchronising NMS that iterates until nothing is undecided.  Results are ordered behind the caller's stream."""
        dev = images.device
        B, _, H, W = images.shape
        if B % 2:
            raise ValueError('interleaved batch must hold an even number of images')
        if is_optical is None:
            is_optical = (torch.arange(B) % 2 == 0).reshape(B, 1)
        data = {'image': images, 'is_optical': is_optical}
        out = self.net(data)
        res = self._settle(self._post(out, valid_mask, dev, B, H, W), out, valid_mask, dev, B, H, W)
        self.tie_redone = 0
        if self.tie_robust and self.nms > 0:
            # tie guards (include/multipoint_hip.h), evaluated on the CONVERGED pass -- the flags of the latest detect call, i.e. of
            # the lists `res` holds: images whose top-k cut fell inside a plateau of scores tied within the default convolution
            # algorithm's rounding noise (or whose NMS decided many such near-ties, whatever `topk`) are re-evaluated ONCE with the
            # tie-exact algorithm, so that their lists follow the reference's exact score order (utils.py:97-116); one small host
            # read per batch.  The read also drains the running total the extra passes of _settle() added to, so that a later
            # check_converged().tie_flagged only counts lists that were actually returned.
            flags, _ = U.topk_ambiguous(dev, B)
            self.tie_redone = U.tie_robust_redo(self.net, data, out, flags) if any(flags) else 0
            self.tie_redone_total += self.tie_redone
            if self.tie_redone:
                res = self._settle(self._post(out, valid_mask, dev, B, H, W), out, valid_mask, dev, B, H, W)
                U.topk_ambiguous(dev, B)                         # (the redone lists flag the same plateaus again: read and drop)
        self._last = res
        if self.keep_maps:
            res.prob, res.desc_map = out['prob'], out['desc']
        return res

    def _settle(self, res, out, valid_mask, dev, B, H, W):
        """Redo the post-processing of a batch until its NMS is exact and its lists hold every keypoint (run_converged)."""
        for attempt in range(6):
            # both conditions are evaluated after EVERY pass, the last one included (check first, at most five redone passes): the
            # exact NMS of a redone pass can keep more keypoints than the asynchronous rounds left, i.e. overflow lists that fitted
            redo_nms = self.nms > 0 and U.nms_unresolved(dev)
            K = res.kp_yx.shape[1]
            # (no device-to-host read of the counts when top-k bounds them by the capacity anyway)
            need = 0 if (self._capacity_is_exact(K) or not B) else int(res.kp_count.max())
            overflow = need > K
            if not (redo_nms or overflow):
                return res
            if attempt == 5:
                break
            # lists that overflowed their capacity (topk == 0: the reference keeps EVERY keypoint, utils.py:109-116) are
            # rebuilt with the exact size -- dropping the row-major tail would silently change nn_map / m_score
            res = self._post(out, valid_mask, dev, B, H, W, nms_rounds=0 if redo_nms else None,
                             capacity=((need + 255) // 256) * 256 if overflow else K)
        raise RuntimeError('run_converged: keypoint lists / NMS did not settle after 5 redone passes (capacity %d)' % res.kp_yx.shape[1])

    def check_converged(self, device=None):
        """Synchronises; raises if the fixed number of asynchronous NMS rounds was not enough.  Also reads the top-k tie guard:
        `self.tie_flagged` = images, over all batches since the previous check, whose top-k cut fell inside a plateau of
        (near-)tied scores -- reported, not raised: the throughput entry does not re-evaluate them (run_converged does)."""
        if self._post_stream is not None:
            self._post_stream.synchronize()
            with torch.cuda.stream(self._post_stream):
                n = U.nms_unresolved(device)
                self.tie_flagged = U.topk_ambiguous(device, 0)[1] if self.nms > 0 else 0
        else:
            n = U.nms_unresolved(device)
            self.tie_flagged = U.topk_ambiguous(device, 0)[1] if self.nms > 0 else 0
        if n:
            raise RuntimeError('box_nms: %d candidates undecided after %d rounds; raise nms_rounds'
                               % (n, self.nms_rounds))
        last = self._last
        if last is not None:
            K = last.kp_yx.shape[1]
            if not self._capacity_is_exact(K):
                need = int(last.kp_count.max()) if last.kp_count.numel() else 0
                if need > K:
                    raise RuntimeError('keypoint lists overflowed: an image has %d keypoints but the lists hold %d; '
                                       'pass capacity >= %d (or use run_converged(), which regrows them)' % (need, K, need))

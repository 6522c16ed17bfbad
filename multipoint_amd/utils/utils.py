"""Host-side mirror of the hot-path functions of multipoint/utils/utils.py.  Signatures, argument
meaning and error behaviour follow the reference; the arithmetic runs in libmultipoint_hip.so."""
import collections
import collections.abc
import ctypes

import torch

from .. import _lib

__all__ = ['dict_update', 'data_to_device', 'data_unsqueeze', 'fix_model_weigth_keys', 'depth_to_space',
           'space_to_depth', 'box_nms', 'detect_keypoints', 'extract_keypoints', 'nms_unresolved',
           'interpolate_descriptors', 'interpolate_descriptors_batched', 'topk_ambiguous', 'topk_tie_guard', 'nms_tie_guard',
           'tie_robust_redo', 'box_nms_tie_robust']


def dict_update(d, u):
    """Update for nested dictionaries (multipoint/utils/utils.py:10-26)."""
    for k, v in u.items():
        if isinstance(v, collections.abc.Mapping):
            d[k] = dict_update(d.get(k, {}), v)
        else:
            d[k] = v
    return d


def data_to_device(data, device):
    """multipoint/utils/utils.py:28-34."""
    for key in data.keys():
        if type(data[key]) is torch.Tensor:
            data[key] = data[key].to(device)
        elif type(data[key]) is dict:
            data[key] = data_to_device(data[key], device)
    return data


def data_unsqueeze(data, dim):
    """multipoint/utils/utils.py:44-50."""
    for key in data.keys():
        if type(data[key]) is torch.Tensor:
            data[key] = data[key].unsqueeze(dim)
        elif type(data[key]) is dict:
            data[key] = data_unsqueeze(data[key], dim)
    return data


def fix_model_weigth_keys(weights):
    """multipoint/utils/utils.py:169-175: keep the text after the last '__' of every key."""
    new_weights = collections.OrderedDict()
    for key, value in weights.items():
        new_weights[key.split('__')[-1]] = value
    return new_weights


def depth_to_space(x, block_size):
    """multipoint/utils/utils.py:64-69 (pure tensor re-indexing, no arithmetic)."""
    N, C, H, W = x.size()
    x = x.view(N, block_size, block_size, C // (block_size ** 2), H, W)
    x = x.permute(0, 3, 4, 1, 5, 2).contiguous()
    return x.view(N, C // (block_size ** 2), H * block_size, W * block_size)


def space_to_depth(x, block_size):
    """multipoint/utils/utils.py:71-76."""
    N, C, H, W = x.size()
    x = x.view(N, C, H // block_size, block_size, W // block_size, block_size)
    x = x.permute(0, 3, 5, 1, 2, 4).contiguous()
    return x.view(N, C * (block_size ** 2), H // block_size, W // block_size)


def _as_cuda_f32(t):
    dev = t.device if t.device.type == 'cuda' else _lib.require_cuda(None)
    return t.to(dev, torch.float32).contiguous(), dev


def _prep_prob(prob, valid_mask):
    if not (len(prob.shape) == 2 or len(prob.shape) == 4):
        raise ValueError('The probability must be either 2D (H,W), or 4D (B, 1, H, W)')
    if len(prob.shape) == 4 and prob.shape[1] != 1:
        raise ValueError('The probability must be either 2D (H,W), or 4D (B, 1, H, W)')
    p, dev = _as_cuda_f32(prob)
    B = 1 if p.dim() == 2 else p.shape[0]
    H, W = p.shape[-2:]
    m = None
    if valid_mask is not None:
        m = valid_mask.to(dev).reshape(B, H, W).ne(0).to(torch.uint8).contiguous()
    return p, m, dev, B, H, W


def box_nms(prob, size, min_prob, iou=0.1, keep_top_k=0, on_cpu=False, valid_mask=None):
    """Non maximum suppression on the heatmap with hypothetical boxes of side `size` centred on
    each pixel; optionally only the top k detections are kept (multipoint/utils/utils.py:78-122).

    Arguments as in the reference.  `on_cpu` is accepted and ignored: the suppression always runs on
    the GPU (the reference default cpu_nms=true only works around a slow GPU path).  The result is
    returned on the device of `prob`.  `valid_mask` (extension) fuses the callers'
    `prob * valid_mask` (predict_align_image_pair.py:128).
    Tie-break: (score descending, row-major pixel index ascending)."""
    p, m, dev, B, H, W = _prep_prob(prob, valid_mask)
    out = torch.empty_like(p)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_box_nms(h.ptr, _lib.ptr(p), _lib.ptr(m), B, H, W, float(size), float(min_prob),
                                 float(iou), int(keep_top_k), _lib.ptr(out), 0, _lib.stream_ptr(dev)))
    return out.to(prob.device)


def detect_keypoints(prob, size, min_prob, iou=0.1, keep_top_k=0, capacity=None, valid_mask=None,
                     max_rounds=0):
    """Fused box_nms + torch.nonzero(prob_nms > min_prob) (predict_align_image_pair.py:127-137,170-171):
    returns (kp_yx int32 [B,K,2], kp_score f32 [B,K], kp_count int32 [B]) on the GPU, rows in
    row-major order.  K = keep_top_k, or `capacity` when keep_top_k == 0."""
    p, m, dev, B, H, W = _prep_prob(prob, valid_mask)
    K = int(capacity) if capacity else (int(keep_top_k) if keep_top_k > 0 else H * W // 4)
    kp = torch.empty((B, K, 2), dtype=torch.int32, device=dev)
    sc = torch.empty((B, K), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_detect_keypoints(h.ptr, _lib.ptr(p), _lib.ptr(m), B, H, W, float(size),
                                          float(min_prob), float(iou), int(keep_top_k), K, _lib.ptr(kp),
                                          _lib.ptr(sc), _lib.ptr(cnt), int(max_rounds), _lib.stream_ptr(dev)))
    return kp, sc, cnt


def nms_unresolved(device=None):
    """Undecided NMS candidates left by the last asynchronous detect_keypoints(max_rounds>0) call."""
    dev = _lib.require_cuda(device)
    h = _lib.get_handle(dev)
    n = ctypes.c_int(0)
    h.check(h.lib.mp_nms_unresolved(h.ptr, ctypes.byref(n), _lib.stream_ptr(dev)))
    return n.value


def topk_ambiguous(device=None, B=0):
    """Tie guards (include/multipoint_hip.h: mp_topk_ambiguous): (flags, total) -- flags[b] is True when, in the LATEST box_nms /
    detect_keypoints call, the top-k cut of image b fell inside a plateau of scores tied within the convolution's rounding noise
    (keep_top_k > 0), split a run of exactly equal scores, or -- footprint guard, any keep_top_k -- many of the image's NMS
    decisions were taken between such scores; total counts the flagged images of all calls since the previous read.  Synchronises."""
    dev = _lib.require_cuda(device)
    h = _lib.get_handle(dev)
    flags = (ctypes.c_int * max(B, 1))()
    n = ctypes.c_int(0)
    h.check(h.lib.mp_topk_ambiguous(h.ptr, flags, int(B), ctypes.byref(n), _lib.stream_ptr(dev)))
    return [bool(flags[b]) for b in range(B)], n.value


def topk_tie_guard(device=None, eps=6e-5, min_each_side=4):
    """Parameters of the top-k tie guard of this device's handle (min_each_side=0 switches it off)."""
    dev = _lib.require_cuda(device)
    h = _lib.get_handle(dev)
    h.check(h.lib.mp_topk_tie_guard(h.ptr, float(eps), int(min_each_side)))


def nms_tie_guard(device=None, min_pairs=16):
    """Footprint tie guard of this device's handle: an image is flagged when at least `min_pairs` candidates were suppressed by
    kept neighbours that are all within the guard's eps of their own score (0 switches it off)."""
    dev = _lib.require_cuda(device)
    h = _lib.get_handle(dev)
    h.check(h.lib.mp_nms_tie_guard(h.ptr, int(min_pairs)))


def tie_robust_redo(net, data, out, flags):
    """Re-evaluate the flagged images of a forward with the tie-exact algorithm: `out['prob']` / `out['desc']` rows of the
    images with flags[b] set are replaced IN PLACE by the forward of `net.direct_twin()` (model.conv_algorithm: direct -- a
    k-ordered multiply-add chain per output like the reference's convolution, so exact ties of the heat map stay ties).
    Returns the number of images redone (0 when the model has no other algorithm: already direct, or mixed_precision)."""
    idx = [b for b, f in enumerate(flags) if f]
    twin = net.direct_twin() if idx else None
    if twin is None:
        return 0
    sel = torch.as_tensor(idx, dtype=torch.long, device=out['prob'].device)
    sub = {'image': data['image'].index_select(0, sel)}
    if data.get('is_optical') is not None:
        sub['is_optical'] = data['is_optical'].to(sel.device).index_select(0, sel) if data['is_optical'].is_cuda \
            else data['is_optical'][torch.as_tensor(idx)]
    redo = twin(sub)
    for k in ('prob', 'logits', 'desc'):
        if out.get(k) is not None and redo.get(k) is not None:
            out[k].index_copy_(0, sel, redo[k])
    return len(idx)


def box_nms_tie_robust(net, data, out, size, min_prob, iou=0.1, keep_top_k=0, on_cpu=False, valid_mask=None):
    """box_nms(out['prob'], ...) for the output `out = net(data)` with the top-k tie guard applied: images whose top-k cut fell
    inside a plateau of (near-)tied scores are re-evaluated with the tie-exact convolution algorithm (`out` is updated in place
    for them) and suppressed again, so the kept indices follow the reference's exact score order (utils.py:97-116) where the
    default algorithm's rounding noise would have picked other members of the plateau.  What the predict_* CLIs call."""
    res = box_nms(out['prob'], size, min_prob, iou, keep_top_k, on_cpu, valid_mask)
    B = out['prob'].shape[0] if out['prob'].dim() == 4 else 1
    flags, _ = topk_ambiguous(out['prob'].device, B)         # (keep_top_k == 0: the footprint guard's flags)
    if any(flags) and tie_robust_redo(net, data, out, flags):
        res = box_nms(out['prob'], size, min_prob, iou, keep_top_k, on_cpu, valid_mask)
        topk_ambiguous(out['prob'].device, B)                # the redone call flags the same plateaus again: read and drop
    return res


def extract_keypoints(prob, thr, capacity=None, valid_mask=None):
    """torch.nonzero((prob > thr)) -- with `valid_mask`, torch.nonzero((prob > thr) * valid_mask) (evaluation.py:156-157)
    -- on the GPU with deterministic row-major order.
    prob (H,W) or (B,1,H,W); returns (kp_yx [B,K,2] int32, score [B,K], count [B])."""
    p, m, dev, B, H, W = _prep_prob(prob, valid_mask)
    K = int(capacity) if capacity else H * W
    kp = torch.empty((B, K, 2), dtype=torch.int32, device=dev)
    sc = torch.empty((B, K), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_extract_keypoints(h.ptr, _lib.ptr(p), _lib.ptr(m), B, H, W, float(thr), K, _lib.ptr(kp),
                                           _lib.ptr(sc), _lib.ptr(cnt), _lib.stream_ptr(dev)))
    return kp, sc, cnt


def _channels_last_desc(desc):
    """(B,D,Hc,Wc) logical -> contiguous (B,Hc,Wc,D) storage without a copy when the tensor already
    is the channels-last view produced by MultiPoint.forward."""
    cl = desc.permute(0, 2, 3, 1)
    return cl if cl.is_contiguous() else cl.contiguous()


def interpolate_descriptors_batched(kp_yx, kp_count, desc, H, W):
    """kp_yx [B,K,2] int32, kp_count [B] int32, desc (B,D,Hc,Wc) -> [B,K,D] unit rows (rows beyond
    kp_count[b] are zero)."""
    dev = kp_yx.device
    d = _channels_last_desc(desc.to(dev, torch.float32))
    B, Hc, Wc, D = d.shape
    K = kp_yx.shape[1]
    out = torch.empty((B, K, D), dtype=torch.float32, device=dev)        # (the kernel writes the zero rows too)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_sample_descriptors(h.ptr, _lib.ptr(d), B, Hc, Wc, D, int(H), int(W),
                                            _lib.ptr(kp_yx.contiguous()), _lib.ptr(kp_count.contiguous()),
                                            K, _lib.ptr(out), _lib.stream_ptr(dev)))
    return out


def interpolate_descriptors(keypoints, descriptors_lowres, H, W):
    """multipoint/utils/utils.py:159-167: bilinear sampling (grid_sample, align_corners=True) of the
    coarse descriptor map at the keypoints followed by L2 normalisation.
    keypoints (N,2) (y,x) integer tensor; descriptors_lowres (D,Hc,Wc); returns (N,D).
    The input keypoints are not modified."""
    dev = descriptors_lowres.device if descriptors_lowres.device.type == 'cuda' else _lib.require_cuda(None)
    N = keypoints.shape[0]
    D = descriptors_lowres.shape[0]
    if N == 0:
        return torch.zeros((0, D), dtype=torch.float32, device=descriptors_lowres.device)
    kp = keypoints.to(dev, torch.int32).reshape(1, N, 2).contiguous()
    cnt = torch.full((1,), N, dtype=torch.int32, device=dev)
    out = interpolate_descriptors_batched(kp, cnt, descriptors_lowres.unsqueeze(0), H, W)
    return out[0].to(descriptors_lowres.device)

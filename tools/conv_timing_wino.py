"""Developer tool (GPU box; library built with -DMP_TIMING, MP_LIB pointing at it): cycles per phase of the persistent
Winograd conv workgroups per work item (one wave per SIMD, so s_memtime is uncontended).  MP_TIMING_H selects the launch
by input height: +H a pooled layer (480: conv2, 240: conv4, 120: conv6), -H an un-pooled one (-240: conv3, -120: conv5,
-60: conv7, conv8, heads -- the last writer)."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O
import multipoint_amd.models as models
from multipoint_amd import _lib
cfg = O.SHIPPED_MODEL_CONFIG
net = models.MultiPoint(cfg); net.load_state_dict(O.make_weights(0, cfg)); net.to('cuda'); net.eval()
lib = ctypes.CDLL(_lib.LIB_PATH)
img = torch.rand(64, 1, 480, 640, device='cuda')
for _ in range(2): net({'image': img})
torch.cuda.synchronize()
sel = int(os.environ.get('MP_TIMING_H', '240'))
K43 = os.environ.get('MP_TIMING_KERNEL') == '43'          # conv_wino43.hip instead of conv_wino.hip
assert (lib.mp_debug_select_height_wino43 if K43 else lib.mp_debug_select_height_wino)(sel) == 0
net.profile(True)
net({'image': img}); torch.cuda.synchronize()
prof = {n: ms for n, ms, fl in net.profile_read()}
buf = (ctypes.c_ulonglong * (256 * 8))()
assert (lib.mp_debug_read_timing_wino43 if K43 else lib.mp_debug_read_timing_wino)(buf, 256 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.float64)
t = t[(t[:, 7] > 0) & (t[:, 7] < 1e6)]
n = t[:, 7]
for i, nm in enumerate(['MFMA steps (all units)', 'unit barriers', 'epilogue', 'unit loop incl. barriers'] + (['s_barrier behind the DMA wait', 'fused first block (next item)', '... its staging part'] if K43 else [])):
    v = t[:, i] / n
    print('%-28s mean %8.0f  p10 %8.0f  p90 %8.0f cycles/item' % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
layer = {480: 'enc.conv1+2' if ('enc.conv1+2' in prof) else 'enc.conv2', 240: 'enc.conv4', 120: 'enc.conv6', -240: 'enc.conv3', -120: 'enc.conv5', -60: 'heads.conv3x3'}.get(sel)
nunits = {480: 8, 240: 8, 120: 16, -240: 8, -120: 8, -60: 16}.get(sel, 0)
print('workgroups %d, items per workgroup %.1f; MFMA per item: %d units x 64 MFMA x 64 = %d cycles' % (len(t), n.mean(), nunits, nunits * 4096))
if layer in prof:
    tot = (t[:, 2] + t[:, 3]).mean()
    print('%s %.3f ms, %.0f ticks per workgroup -> %.3f GHz' % (layer, prof[layer], tot, tot / prof[layer] * 1e-6))

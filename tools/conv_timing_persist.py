"""Developer tool (GPU box, library built with -DMP_TIMING, MP_LIB pointing at it): cycles per phase of the persistent
fp32 conv workgroups, per work item.  MP_TIMING_H selects the launch by input height (240: conv3 then conv4 (last
writer), 120: conv5/conv6, 60: conv7/8/heads...)."""
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
assert lib.mp_debug_select_height(sel) == 0
net.profile(True)
net({'image': img}); torch.cuda.synchronize()
prof = {n: ms for n, ms, fl in net.profile_read()}
net.profile(False)
buf = (ctypes.c_ulonglong * (512 * 8))()
assert lib.mp_debug_read_timing(buf, 512 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8).astype(np.float64)
t = t[(t[:, 7] > 0) & (t[:, 7] < 1e6)]
n = t[:, 7]
names = ['item start -> first step', 'MFMA steps (all chunks)', 'chunk-end barrier', '(unused)', 'epilogue', '(unused)', '(unused)']
tot = 0
for i, nm in enumerate(names):
    v = t[:, i] / n
    tot += v.mean()
    print('%-28s mean %8.0f  p10 %8.0f  p90 %8.0f cycles/item' % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
print('workgroups %d, items per workgroup %.1f, sum %.0f cycles/item (MFMA: 36 steps x 1024 = 36864 per 32-channel chunk)' % (len(t), n.mean(), tot))
layer = {240: 'enc.conv4', 120: 'enc.conv6', 60: 'desc.conv1x1'}.get(sel)
if layer in prof:
    ticks = (t[:, :7].sum(axis=1)).mean()
    print('%s: %.3f ms in this (instrumented) run, %.0f ticks per workgroup -> %.3f GHz shader clock during the kernel' % (layer, prof[layer], ticks, ticks / prof[layer] * 1e-6))
    nch = {240: 2, 120: 4}.get(sel, 0)
    if nch:
        need = n.mean() * nch * 576 * 64
        print('MFMA cycles needed per SIMD (1 wave) %.0f = %.1f %% of the ticks' % (need, 100 * need / ticks))

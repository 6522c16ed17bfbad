"""Developer tool (GPU box, library built with MP_HIPCC_FLAGS=-DMP_TIMING): cycles per phase of the persistent fp16
conv workgroups, averaged per work item.  MP_TIMING_H selects the launch by input height (1024: enc.conv2,
512: conv3 then conv4 (last writer), ...)."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O
import multipoint_amd.models as models
from multipoint_amd import _lib
cfg = dict(O.SHIPPED_MODEL_CONFIG); cfg['mixed_precision'] = True
net = models.MultiPoint(cfg); net.load_state_dict(O.make_weights(0, cfg)); net.to('cuda'); net.eval()
lib = ctypes.CDLL(_lib.LIB_PATH)
img = torch.rand(16, 1, 1024, 1280, device='cuda')
for _ in range(2): net({'image': img})
torch.cuda.synchronize()
sel = int(os.environ.get('MP_TIMING_H', '1024'))
assert lib.mp_debug_select_height_f16(sel) == 0
net({'image': img}); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (512 * 8))()
assert lib.mp_debug_read_timing_f16(buf, 512 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8).astype(np.float64)
t = t[t[:, 7] > 0]
n = t[:, 7]
names = ['item start -> first step', 'MFMA steps', 'barrier after steps', 'data landed + LDS write', 'epilogue', 'barrier before next item']
tot = 0
for i, nm in enumerate(names):
    v = t[:, i] / n
    tot += v.mean()
    print('%-28s mean %8.0f  p10 %8.0f  p90 %8.0f cycles/item' % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
print('items per workgroup %.1f, sum %.0f cycles/item (ideal MFMA: 36 steps x 128 = 4608 per 64-channel chunk)' % (n.mean(), tot))

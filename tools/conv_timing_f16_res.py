"""Developer tool (GPU box; library variant built with -DMP_TIMING for conv_f16_res.hip: tools/build_variant.sh rt "-DMP_TIMING"
conv_f16_res.hip, run with MP_LIB=...): cycles per phase and item of the fused fp16 conv1+conv2 launch (wave 0 of groups 0 / 1)."""
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
h = net._handle
# only the first conv launch: time it alone through the profile of a forward (the last writer of the table is conv4/conv5's
# launches of the same kernel family, so read after a forward restricted to... simplest: every res launch overwrites; the fused
# launch is identified by its item count)
for _ in range(2): net({'image': img})
torch.cuda.synchronize()
net({'image': img}); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (512 * 8))()
assert lib.mp_debug_read_timing_f16_res(buf, 512 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8).astype(np.float64)
t = t[t[:, 7] > 0]
n = t[:, 7]
names = ['item start -> first step', 'MFMA steps (2 chunks)', 'barrier behind steps (x2)', 'tile production / LDS write (x2)',
         'barrier before chunk 1', 'epilogue', 'barrier before next item']
tot = 0
for i, nm in enumerate(names):
    v = t[:, i] / n
    tot += v.mean()
    print('%-34s mean %8.0f  p10 %8.0f  p90 %8.0f cycles/item' % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
print('items per group %.1f, sum %.0f cycles/item (MFMA time: 36 steps x 128 = 4608)' % (n.mean(), tot))

"""Developer tool (GPU box; library variant built with -DMP_TIMING for conv_f16_one.hip, run with MP_LIB=...): ticks per phase and
item of the one-wave-per-SIMD fused fp16 conv1+conv2 launch (wave 0 of every workgroup)."""
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
for _ in range(3): net({'image': img})
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8))()
assert lib.mp_debug_read_timing_f16_one(buf, 256 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.float64)
t = t[t[:, 7] > 0]
n = t[:, 7]
names = ['chunk 0 steps', 'mid barrier', 'chunk 1 steps', 'accumulator copy', 'decode + next offsets', 'end barrier']
tot = 0
for i, nm in enumerate(names):
    v = t[:, i] / n
    tot += v.mean()
    print('%-24s mean %8.0f  p10 %8.0f  p90 %8.0f ticks/item' % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
print('items per workgroup %.1f, sum %.0f ticks/item (MFMA time: 36 steps x 128 = 4608)' % (n.mean(), tot))

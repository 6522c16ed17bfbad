"""Developer tool (GPU box, library built with MP_HIPCC_FLAGS=-DMP_TIMING): where does a conv workgroup's
time go?  Prints mean cycles of prologue / chunk 0 / chunk boundary / chunk 1 / epilogue for the fused
conv1+conv2 launch and for conv4 (both 2 chunks per tile)."""
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
# the LAST conv launch that wrote the stamps is heads/1x1; so run layers selectively: use small hack --
# run a forward on an input whose later layers are tiny?  Instead read after a forward restricted by env.
sel_h = int(os.environ.get('MP_TIMING_H', '480'))     # 480: fused conv1+2, 240: conv3 (no pool) then conv4 (pool; last writer)
assert lib.mp_debug_select_height(sel_h) == 0
buf = (ctypes.c_ulonglong * (8192 * 16))()
net({'image': img}); torch.cuda.synchronize()
assert lib.mp_debug_read_timing(buf, 8192 * 16) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 16).astype(np.float64)
ok = (t[:, 7] > t[:, 0]) & (t[:, 0] > 0)
t = t[ok]
print('blocks', len(t))
d = {'  index+issue (t1-t0)': t[:, 1] - t[:, 0], '  land+ldswrite (t3-t1)': t[:, 3] - t[:, 1], '  barrier (t2-t3)': t[:, 2] - t[:, 3],
     'prologue (t2-t0)': t[:, 2] - t[:, 0], 'chunk0 (t5-t2)': t[:, 5] - t[:, 2], 'boundary (t4-t5)': t[:, 4] - t[:, 5],
     'chunk1 (t6-t4)': t[:, 6] - t[:, 4], 'epilogue (t7-t6)': t[:, 7] - t[:, 6], 'total (t7-t0)': t[:, 7] - t[:, 0]}
if os.environ.get('MP_TIMING_PROLOGUE') == '1':
    d = {'decode (t8-t0)': t[:, 8] - t[:, 0], 'goff (t9-t8)': t[:, 9] - t[:, 8], 'issue loads (t1-t9)': t[:, 1] - t[:, 9],
         'land+ldswrite (t3-t1)': t[:, 3] - t[:, 1], 'barrier (t2-t3)': t[:, 2] - t[:, 3]}
if os.environ.get('MP_TIMING_CAL') == '1':
    d = {'256 dependent v_add (t9-t8)': t[:, 9] - t[:, 8]}
for k, v in d.items():
    print('%-24s mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f' % (k, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))

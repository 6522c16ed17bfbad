"""Developer tool (GPU box; library built with -DMP_TIMING, MP_LIB pointing at it): how evenly conv_wino43.hip's persistent
workgroups finish -- per launch (MP_TIMING_H as in conv_timing_wino43.py) the per-workgroup time (unit loops + epilogues of all its
items, s_memtime ticks of 10 ns) as mean / max over the grid and per XCD (workgroup id & 7).  A static item partition ends with its
slowest workgroup: max / mean - 1 is what a dynamic item queue could recover at most."""
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
for _ in range(3): net({'image': img})
torch.cuda.synchronize()
for sel in [int(x) for x in os.environ.get('MP_TIMING_H', '480,-240,240,-120,120,-60').split(',')]:
    assert lib.mp_debug_select_height_wino43(sel) == 0
    rows = []
    for rep in range(3):
        net({'image': img}); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (256 * 8))()
        assert lib.mp_debug_read_timing_wino43(buf, 256 * 8) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.float64)
        tot = t[:, 3] + t[:, 2]
        n = t[:, 7]
        ok = (n > 0) & (n < 1e6)
        x = np.arange(256) & 7
        rows.append('rep %d: workgroups %d, items/wg %.2f (min %d max %d); wg time mean %.0f max %.0f (+%.2f %%) min %.0f (%.2f %%); per XCD mean: %s'
                    % (rep, ok.sum(), n[ok].mean(), n[ok].min(), n[ok].max(), tot[ok].mean(), tot[ok].max(), 100 * (tot[ok].max() / tot[ok].mean() - 1),
                       tot[ok].min(), 100 * (tot[ok].min() / tot[ok].mean() - 1),
                       ' '.join('%.0f' % tot[ok & (x == k)].mean() for k in range(8))))
    print('H %d' % sel); print('\n'.join('   ' + r for r in rows), flush=True)

"""Developer tool (GPU box; library built with -DMP_TIMING, MP_LIB pointing at it): cycles per phase of conv_wino43.hip's
persistent workgroups per work item, as seen by ONE wave (MP_TIMING_WAVE, default 0; waves 0-3 carry the input transform).
MP_TIMING_H selects the launch by input height: +H a pooled layer (480: conv1+2, 240: conv4, 120: conv6), -H an un-pooled one
(-240: conv3, -120: conv5, -60: conv7, conv8, heads -- the last writer)."""
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
for sel in [int(x) for x in os.environ.get('MP_TIMING_H', '240').split(',')]:
    for wv in [int(x) for x in os.environ.get('MP_TIMING_WAVE', '0,2,4').split(',')]:
        assert lib.mp_debug_select_height_wino43(sel) == 0
        if hasattr(lib, 'mp_debug_select_wave_wino43'): assert lib.mp_debug_select_wave_wino43(wv) == 0
        net({'image': img}); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (256 * 8))()
        assert lib.mp_debug_read_timing_wino43(buf, 256 * 8) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.float64)
        t = t[(t[:, 7] > 0) & (t[:, 7] < 1e6)]
        n = t[:, 7]
        print('H %d wave %d: workgroups %d, items per workgroup %.1f' % (sel, wv, len(t), n.mean()))
        for i, nm in enumerate(['units incl. barriers', 'DMA / LDS wait in front of the barrier', 'epilogue', 'unit loop of an item', 's_barrier behind the wait', '... in even units', '... in odd units']):
            v = t[:, i] / n
            print('   %-40s mean %8.0f  p10 %8.0f  p90 %8.0f ticks/item' % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))

"""Developer tool (GPU box): per-layer hipEvent timings of mp_forward at the bench shape."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O
import multipoint_amd.models as models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
cfg = dict(O.SHIPPED_MODEL_CONFIG)
if len(sys.argv) > 4 and sys.argv[4] == 'f16':
    cfg['mixed_precision'] = True             # BASELINE configs[4]: fp16 MFMA path
sd = O.make_weights(0, cfg)
ZERO = os.environ.get('ZERO') == '1'
if ZERO:
    sd = {k: (torch.zeros_like(v) if (v.dim() == 4) else v) for k, v in sd.items()}
net = models.MultiPoint(cfg); net.load_state_dict(sd); net.to('cuda'); net.eval()
img = torch.rand(B, 1, H, W, device='cuda') * (0.0 if ZERO else 1.0)
for _ in range(3): net({'image': img})
torch.cuda.synchronize()
net.profile(True)
for _ in range(5): out = net({'image': img})
torch.cuda.synchronize()
prof = net.profile_read()
by = {}
for n, ms, fl in prof: by.setdefault(n, []).append((ms, fl))
tot = 0
for n, v in by.items():
    ms = np.median([m for m, _ in v]); fl = v[0][1]; tot += ms
    print('%-22s %8.3f ms  %7.2f TF/s  (%4.1f%% of 157.3)' % (n, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100))
flt = sum(v[0][1] for v in by.values())
print('total %.3f ms  %.2f TF/s  -> %.1f pairs/s forward-only' % (tot, flt / tot / 1e9, B / 2 / tot * 1e3))

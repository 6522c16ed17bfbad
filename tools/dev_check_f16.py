"""Developer smoke script (GPU box): fp16 (mixed_precision) forward vs the oracle's autocast restatement, layer
timings, and distance to the fp32 result."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O
import multipoint_amd.models as models

cfg = dict(O.SHIPPED_MODEL_CONFIG); cfg['mixed_precision'] = True
cfg32 = dict(O.SHIPPED_MODEL_CONFIG)
sd = O.make_weights(0, cfg)
net = models.MultiPoint(cfg); net.load_state_dict(sd); net.to('cuda'); net.eval()
for (B, H, W) in [(2, 64, 64), (1, 240, 320), (3, 72, 104), (1, 480, 640)]:
    img = O.make_images(3, B, H, W)
    ref = O.forward(sd, img, cfg)
    ref32 = O.forward(sd, img, cfg32)
    out = net({'image': img.cuda()})
    torch.cuda.synchronize()
    p = out['prob'].cpu(); d = out['desc'].cpu()
    print('f16 fwd', (B, H, W), 'vs oracle-f16: prob', (p - ref['prob']).abs().max().item(), 'desc', (d - ref['desc']).abs().max().item(),
          '| oracle-f16 vs oracle-f32: prob', (ref['prob'] - ref32['prob']).abs().max().item(), 'desc', (ref['desc'] - ref32['desc']).abs().max().item(),
          'nan', torch.isnan(p).any().item(), torch.isnan(d).any().item())
    net.set_force_return_logits(True)
    lg = net({'image': img.cuda()})['logits'].cpu()
    net.set_force_return_logits(False)
    rl = O.forward(sd, img, cfg, return_logits=True)['logits']
    print('   logits maxabs', (lg - rl).abs().max().item(), 'mismatching', (lg != rl).float().mean().item())
if len(sys.argv) > 1:
    B = int(sys.argv[1]); H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 640)
    img = torch.rand(B, 1, H, W, device='cuda')
    for _ in range(3): net({'image': img})
    net.profile(True)
    for _ in range(5): net({'image': img})
    torch.cuda.synchronize()
    prof = net.profile_read()
    by = {}
    for n, ms, fl in prof: by.setdefault(n, []).append((ms, fl))
    tot = 0; totf = 0
    for n, v in by.items():
        ms = np.mean([m for m, _ in v]); fl = v[0][1]; tot += ms; totf += fl
        print('  %-20s %8.3f ms  %8.1f TFLOP/s' % (n, ms, fl / ms / 1e9))
    print('  total %.3f ms  %.1f TFLOP/s  -> %.1f pairs/s forward-only' % (tot, totf / tot / 1e9, B / 2 / tot * 1e3))

"""Developer tool (GPU box): same-box A/B of library builds, per layer.  Runs tools/bench_layers.py-style timings of each variant in
ALTERNATING child processes (the part's clock drifts by a few percent over tens of seconds: single runs are not comparable) and
prints, per layer, the median and the minimum over the rounds.
    python tools/ab_layers.py ROUNDS variantA variantB ... [-- bench_layers args]     (variant `new` = the regular library,
    anything else = multipoint_amd/libmultipoint_hip_exp_<name>.so; `name:KEY=VAL,..` adds MP_DEBUG switches)"""
import json, os, subprocess, sys
import numpy as np
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(here)
args = sys.argv[1:]
extra = []
if '--' in args:
    i = args.index('--'); extra = args[i + 1:]; args = args[:i]
rounds = int(args[0]); variants = args[1:]
CHILD = r'''
import sys, os, json
import numpy as np, torch
sys.path.insert(0, %r)
from oracle import mp_oracle as O
import multipoint_amd.models as models
a = sys.argv[1:]
B = int(a[0]) if len(a) > 0 else 64; H = int(a[1]) if len(a) > 1 else 480; W = int(a[2]) if len(a) > 2 else 640
cfg = dict(O.SHIPPED_MODEL_CONFIG)
if len(a) > 3 and a[3] == 'f16': cfg['mixed_precision'] = True
sd = O.make_weights(0, cfg)
net = models.MultiPoint(cfg); net.load_state_dict(sd); net.to('cuda'); net.eval()
img = torch.rand(B, 1, H, W, device='cuda')
for _ in range(8): net({'image': img})
torch.cuda.synchronize()
net.profile(True)
for _ in range(24): net({'image': img})
torch.cuda.synchronize()
by = {}
for n, ms, fl in net.profile_read(): by.setdefault(n, []).append(ms)
print('AB_RESULT ' + json.dumps({k: float(np.median(v)) for k, v in by.items()}))
''' % root
res = {v: [] for v in variants}
for r in range(rounds):
    for v in (variants if r % 2 == 0 else variants[::-1]):
        name, _, dbg = v.partition(':')
        env = dict(os.environ)
        env.pop('MP_LIB', None); env.pop('MP_DEBUG', None)
        if name != 'new':
            env['MP_LIB'] = os.path.join(root, 'multipoint_amd', 'libmultipoint_hip_exp_%s.so' % name)
        if dbg:
            env['MP_DEBUG'] = dbg
        out = subprocess.run([sys.executable, '-c', CHILD] + extra, env=env, capture_output=True, text=True).stdout
        line = [l for l in out.split('\n') if l.startswith('AB_RESULT ')]
        if line:
            res[v].append(json.loads(line[0][10:]))
layers = list(res[variants[0]][0].keys()) if res[variants[0]] else []
print('%-16s' % 'layer' + ''.join('%24s' % v for v in variants))
for k in layers + ['total']:
    row = '%-16s' % k
    for v in variants:
        xs = [sum(d.values()) if k == 'total' else d[k] for d in res[v]]
        row += '   med %7.4f min %7.4f' % (float(np.median(xs)), min(xs))
    print(row)
print('rounds per variant:', {v: len(res[v]) for v in variants})

"""Developer tool (GPU box): random frame sizes / batch sizes / model variants -- the default routing (conv_wino43.hip with the
fused first block where it applies, conv_wino43b.hip elsewhere; split launches for B <= 2) against the any-frame kernel alone
(MP_DEBUG=wino43_gen=2) and the direct kernels (MP_DEBUG=no_winograd) on the same inputs; `f16` as the first
argument: the mixed_precision path (resident weights + fused first block) against the streaming kernels (MP_DEBUG=f16_no_res) and
the fp16 oracle.     python tools/fuzz_shapes.py [f16] [trials] [seed]"""
import os, sys, random, torch
sys.path.insert(0, os.getcwd())
from oracle import mp_oracle as O
import multipoint_amd.models as M
F16 = len(sys.argv) > 1 and sys.argv[1] == 'f16'
args = [a for a in sys.argv[1:] if a != 'f16']
NTR = int(args[0]) if args else 24
random.seed(int(args[1]) if len(args) > 1 else 7)
def net(cfg, seed):
    n = M.MultiPoint(dict(cfg)); n.load_state_dict(O.make_weights(seed, cfg)); n.to('cuda'); n.eval(); return n
worst = 0.0
for trial in range(NTR):
    cfg = dict(O.SHIPPED_MODEL_CONFIG)
    if F16: cfg['mixed_precision'] = True
    if trial % 4 == 1: cfg['multispectral'] = True
    if trial % 4 == 2: cfg['bn_first'] = True
    if trial % 6 == 5 and not F16: cfg['descriptor_size'] = 128
    if trial % 5 == 3 and not F16: cfg['reflection_pad'] = False
    H = 8 * random.randint(2, 40); W = 8 * random.randint(2, 60); B = random.choice([1, 1, 2, 2, 3, 4, 5])
    img = O.make_images(trial, B, H, W).cuda()
    flags = torch.tensor([[i % 2 == 0] for i in range(B)])
    os.environ.pop('MP_DEBUG', None)
    a = net(cfg, trial)({'image': img, 'is_optical': flags})
    if F16:
        os.environ['MP_DEBUG'] = 'f16_no_res'
        b = net(cfg, trial)({'image': img, 'is_optical': flags})
        os.environ.pop('MP_DEBUG')
        ref = O.forward(O.make_weights(trial, cfg), img.cpu(), cfg, is_optical=flags)
        c = {'prob': ref['prob'].cuda(), 'desc': ref['desc'].cuda()}
    else:
        os.environ['MP_DEBUG'] = 'wino43_gen=2'
        b = net(cfg, trial)({'image': img, 'is_optical': flags})
        os.environ.pop('MP_DEBUG')
        os.environ['MP_DEBUG'] = 'no_winograd'
        c = net(cfg, trial)({'image': img, 'is_optical': flags})
        os.environ.pop('MP_DEBUG')
    dp = max((a['prob'] - b['prob']).abs().max().item(), (a['prob'] - c['prob']).abs().max().item())
    dd = max((a['desc'] - b['desc']).abs().max().item(), (a['desc'] - c['desc']).abs().max().item())
    worst = max(worst, dp, dd)
    print(trial, B, H, W, {k: cfg[k] for k in ('multispectral', 'bn_first', 'descriptor_size', 'reflection_pad')}, '%.2e %.2e' % (dp, dd), flush=True)
    assert (dp < 2e-2 and dd < 4e-3) if F16 else (dp < 1e-4 and dd < 1e-4), 'MISMATCH'
print('worst', worst)

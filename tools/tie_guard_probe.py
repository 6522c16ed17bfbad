"""Developer probe (GPU): what the top-k tie guard sees on the structured / trained-like parity inputs.
For each case: keypoints differing from the CPU oracle with the raw default forward, and through PairPipeline.run_converged
(tie-robust: flagged images redone with the direct algorithm); the guard's flags; the kind of every flip."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mp_oracle as O          # noqa: E402
from oracle import trained_like as T       # noqa: E402
from oracle import flip_accounting as FA   # noqa: E402
import multipoint_amd.models as M          # noqa: E402
import multipoint_amd.utils as U           # noqa: E402
from multipoint_amd.pipeline import PairPipeline  # noqa: E402

PRED = {'nms': 4, 'detection_threshold': 0.015, 'topk': 1000, 'cpu_nms': False,
        'matching': {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True}, 'knn_matches': False}}
nms = lambda m: O.box_nms(m, 4, 0.015, keep_top_k=0)


def account(tag, prob_cpu, prob_gpu):
    s, per = FA.account_batch(prob_cpu, prob_gpu, nms, 4, 0.015, 0.1, 1000)
    print(tag, json.dumps({k: s[k] for k in ('keypoints_total', 'keypoints_differing', 'unexplained', 'root_flips',
                                             'topk_boundary_flips', 'max_prob_err')}),
          'per image:', [p['keypoints_differing'] for p in per], flush=True)
    return s


def run(name, cfg, sd, img):
    prob_cpu = O.forward(sd, img, cfg)['prob'].numpy()
    net = M.MultiPoint(dict(cfg)); net.load_state_dict(sd); net.to('cuda'); net.eval()
    raw = net({'image': img.cuda()})['prob'].cpu().numpy()
    account('[%s raw auto]' % name, prob_cpu, raw)
    for eps, mn in ((6e-5, 4), (6e-5, 2), (2e-4, 4)):
        U.topk_tie_guard('cuda', eps, mn)
        pipe = PairPipeline(net, PRED, capacity=1000, keep_maps=True)
        res = pipe.run_converged(img.cuda())
        print('[%s eps %g min %d] redone %d' % (name, eps, mn, pipe.tie_redone))
        account('[%s tie-robust eps %g min %d]' % (name, eps, mn), prob_cpu, res.prob.cpu().numpy())
    U.topk_tie_guard('cuda', 6e-5, 4)


cfg = dict(O.SHIPPED_MODEL_CONFIG)
sd = T.trained_like_weights(1, cfg, **T.SEVERITIES['wide'])
run('structured8', cfg, sd, T.structured_images(4, 8, 480, 640))
for sev in ('mild', 'wide', 'wide+hot'):
    c, sd, img, r32, r64 = T.case(sev, 11, 2, 480, 640)
    run('trained-like ' + sev, c, sd, img)
sd = O.make_weights(0, cfg)
run('noise', cfg, sd, O.make_images(3, 8, 480, 640))

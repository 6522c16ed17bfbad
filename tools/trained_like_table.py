"""GPU box: error of every convolution family on trained-like statistics (oracle/trained_like.py), for docs/HISTORY.md section 4.
    python tools/trained_like_table.py [H W B]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import trained_like as T          # noqa: E402  (test infrastructure; this is a measurement tool, not the product)
from oracle import mp_oracle as O             # noqa: E402
from oracle import flip_accounting as FA      # noqa: E402

H, W, B = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (480, 640, 4)))
rows = []
for sev in T.SEVERITIES:
    cfg, sd, img, r32, r64 = T.case(sev, 11, B, H, W)
    cpu = {k + '_vs_f64': float((r32[k].double() - r64[k]).abs().max()) for k in ('prob', 'desc', 'logits')}
    rows.append(dict(severity=sev, path='ATen CPU fp32 (oracle)', **cpu))
    print(json.dumps(rows[-1]), flush=True)
    nms = lambda m: O.box_nms(m, 4, 0.015, keep_top_k=0)
    for name, env in T.VARIANT_ENV.items():
        got = T.gpu_outputs(cfg, sd, img, env)
        e = T.errors(got, r32, r64)
        s, _ = FA.account_batch(r32['prob'].numpy(), got['prob'].numpy(), nms, 4, 0.015, 0.1, 1000)
        e.update(severity=sev, path=name, keypoints_total=s['keypoints_total'], keypoints_differing=s['keypoints_differing'],
                 unexplained=s['unexplained'], max_root_margin=s['max_root_margin'])
        rows.append(e)
        print(json.dumps(e), flush=True)

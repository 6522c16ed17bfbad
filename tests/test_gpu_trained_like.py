"""GPU (MI355X): every convolution family on TRAINED-LIKE statistics (round-2 verdict, "What's missing" 1).

All other parity inputs are white noise through benign synthetic weights.  oracle/trained_like.py builds the hard case:
structured images (polygons, ramps, saturated regions, DC offsets, periodic texture) through weights whose BatchNorm
statistics are calibrated on such images, with running_var log-uniform over [1e-3, 1e2], |gamma| log-uniform over [0.1, 10]
(15 % negative) and -- 'wide+hot' -- three filters per layer x10 outside their statistics.  The reference's pretrained blob
(model_weights/multipoint/latest.model, MultiPoint.py:143-148 is the block it parameterises) is not shipped, so this is
the closest available stand-in for it.

What is asserted, per severity and for F(4x4,3x3) (default), the any-frame-size F(4x4,3x3) kernel and the direct kernel on the SAME inputs:
  * descriptors within 1e-4 of the fp32 CPU oracle (north_star's bar) -- observed <= 5e-5 everywhere;
  * prob: within max(1e-4, 4 x the error the fp32 CPU oracle ITSELF has against an fp64 evaluation; 8 x on 'wide+hot'): on
    'wide+hot' ATen's own fp32 result is 5.5e-4 from the truth, so no fp32 implementation can be held to 1e-4 there; observed
    F(4x4,3x3) 2.8x on mild / wide and 2.9x - 5.8x on wide+hot (depending on the summation order of the first block), direct
    <= 2.2x ATen's error;
  * every keypoint that differs from the oracle's list is an explained fp32-noise flip (oracle/flip_accounting.py).  On
    exactly piecewise-constant images the CPU map holds EXACT ties (ATen evaluates equal patches equally); F(4x4,3x3)
    evaluates the 16 outputs of a tile by 16 different formulas, so it breaks those ties by rounding noise where the direct
    kernel keeps them -- 110 of 2 055 keypoints on the 'wide' case, margin 0, all explained.
The interpolation points of the F(4x4,3x3) transforms were changed from the textbook {0, +-1, +-2} to {0, +-3/4, +-3/2} on
this evidence: 3.3x smaller error on every severity (csrc/mp_common.h; the table is in docs/HISTORY.md section 4)."""
import json

import pytest

pytestmark = pytest.mark.gpu

H, W, B = 480, 640, 2
_cases = {}


def _case(sev):
    from oracle import trained_like as T
    if sev not in _cases:
        _cases[sev] = T.case(sev, 11, B, H, W)
    return _cases[sev]


@pytest.mark.parametrize('variant', ['F(4x4,3x3)', 'F(4x4,3x3) general', 'direct'])
@pytest.mark.parametrize('sev', ['mild', 'wide', 'wide+hot'])
def test_conv_families_on_trained_like_statistics(oracle, sev, variant):
    from oracle import trained_like as T
    from oracle import flip_accounting as FA
    cfg, sd, img, r32, r64 = _case(sev)
    aten = {k: float((r32[k].double() - r64[k]).abs().max()) for k in ('prob', 'desc', 'logits')}
    got = T.gpu_outputs(cfg, sd, img, T.VARIANT_ENV[variant])
    e = T.errors(got, r32, r64)
    nms = lambda m: oracle.box_nms(m, 4, 0.015, keep_top_k=0)
    # keypoints: from the heat map the product's drivers use (PairPipeline.run_converged: images flagged by the top-k tie guard are
    # re-evaluated with the tie-exact algorithm); the raw forward's own accounting is printed beside it
    s_raw, _ = FA.account_batch(r32['prob'].numpy(), got['prob'].numpy(), nms, 4, 0.015, 0.1, 1000)
    s, _ = FA.account_batch(r32['prob'].numpy(), got['prob_tie_robust'].numpy(), nms, 4, 0.015, 0.1, 1000)
    print('\n[trained-like %s %s] %s | ATen fp32 vs fp64: %s | keypoints %d differing %d (raw forward: %d; %d images redone) unexplained %d'
          % (sev, variant, json.dumps(e), json.dumps(aten), s['keypoints_total'], s['keypoints_differing'],
             s_raw['keypoints_differing'], got['tie_redone'], s['unexplained']))
    assert s_raw['unexplained'] == 0 and s_raw['max_unexplained_margin'] == 0.0, s_raw
    assert e['desc_vs_cpu32'] <= 1e-4, e
    # 'wide+hot' is ill-conditioned on purpose (filters 10x outside their BatchNorm statistics): the max-norm there moves by 2x with
    # the summation order of ONE layer (F(4x4,3x3): 1.6e-3 with the first block's bias added last, 3.2e-3 with the bias as the
    # accumulator's initial value -- on the benign inputs the second form is the closer one), so the bound for it is a looser multiple
    k = 8.0 if sev == 'wide+hot' else 4.0
    assert e['prob_vs_f64'] <= max(1e-4, k * aten['prob']), (e, aten)
    assert e['logits_vs_f64'] <= max(1e-3, (k + 1.0) * aten['logits']), (e, aten)
    assert s['unexplained'] == 0 and s['max_unexplained_margin'] == 0.0, s
    assert s['roots_within_measured_noise']
    # the direct kernel evaluates equal patches equally: the exact ties of piecewise-constant images survive; the F(4x4,3x3) kernels
    # reorder them by rounding noise (round 4: 110-120 of 2 060 keypoints on 'wide', all at a top-k cut inside a plateau of tied
    # scores) -- the top-k tie guard flags exactly those images and the pipeline redoes them with `direct`: 0 of 2 000 on 'mild' /
    # 'wide', 4 of 2 002 on the ill-conditioned 'wide+hot' (where `direct` itself is 5.8e-4 from the oracle's map)
    assert s['keypoints_differing'] <= 0.01 * s['keypoints_total'], s
    # (the 'direct' variant here is selected by the developer switch MP_DEBUG=no_winograd, which the model config does not see: its flagged
    # images are redone too -- by the same kernels, to the same bits)


def test_f16_path_on_trained_like_statistics(oracle):
    """The fp16 MFMA path (mixed_precision) on the 'mild' trained-like weights and structured images: nothing overflows fp16,
    and the HIP path stays within the fp16 noise floor of the oracle's autocast restatement -- measured, as everywhere for this
    mode, in fp16 steps of the quantity the network rounds (tests/test_gpu_f16.py).  The wider activation range of calibrated
    BatchNorm statistics (|gamma| up to 3, running_var over three decades) is what distinguishes this from the benign case."""
    import numpy as np
    import torch
    import multipoint_amd.models as M
    from oracle import trained_like as T
    from oracle import f16_stats as S
    cfg32 = dict(oracle.SHIPPED_MODEL_CONFIG)
    sd = T.trained_like_weights(11, cfg32, **T.SEVERITIES['mild'])
    img = T.structured_images(14, 2, 240, 320)
    cfg = dict(cfg32); cfg['mixed_precision'] = True
    ref = oracle.forward(sd, img, cfg)
    ref_l = oracle.forward(sd, img, cfg, return_logits=True)['logits']
    assert torch.isfinite(ref['desc']).all() and torch.isfinite(ref_l).all()
    net = M.MultiPoint(dict(cfg)); net.load_state_dict(sd); net.to('cuda'); net.eval()
    out = net({'image': img.cuda()})
    net.set_force_return_logits(True)
    lg = net({'image': img.cuda()})['logits'].cpu().numpy()
    assert np.isfinite(lg).all() and torch.isfinite(out['desc']).all() and torch.isfinite(out['prob']).all()
    ls = S.logits_stats(lg, ref_l.numpy())
    ds = S.desc_stats(out['desc'].cpu().numpy(), ref['desc'].numpy(), channel_axis=1)
    print('\n[f16 trained-like mild] logits %s\n desc %s' % (ls, ds))
    assert ls['median'] <= 1.0 and ls['p999'] <= 8.0, ls
    assert ds['median'] <= 1.0 and ds['p999'] <= 8.0, ds
    assert abs(ls['mean_signed']) <= 0.05 * ls['mean_abs'] and abs(ds['mean_signed']) <= 0.05 * ds['mean_abs'], (ls, ds)

"""GPU (MI355X): homographic adaptation (mp_warp_perspective, mp_ha_valid_mask, mp_ha_begin / accumulate / finalize,
mp_gaussian_filter and the driver multipoint_amd.utils.homographic_adaptation*) against the oracle's restatement of
multipoint/utils/homographies.py and against the golden outputs of the reference driver.

Tolerances.  The reference evaluates the warp in float32 NORMALISED coordinates (kornia), the HIP kernels in float64
pixel coordinates: sample positions differ by ~1e-5 px at 64x64 (~1e-4 px at 480x640), which moves a bilinear sample
by that fraction of the local gradient and can flip a nearest-neighbour sample that falls within that distance of a
pixel boundary.  Hence: exact equality where the coordinates are exact (integer shifts, flips), <= 2e-3 on bilinear
samples of white noise, and a bounded fraction of differing pixels for nearest / mask-edge effects."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ha_oracle as HA

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda:0'


def _homographies(seed, n, H, W, **kw):
    from multipoint_amd.utils.homographies import sample_homography
    np.random.seed(seed)
    return np.stack([sample_homography(np.array([H, W]), **kw) for _ in range(n)])


def test_warp_exact_cases():
    import multipoint_amd.utils as U
    torch.manual_seed(1)
    src = torch.rand(3, 1, 24, 40)
    cases = [torch.eye(3), torch.tensor([[1., 0, 5], [0, 1, -3], [0, 0, 1]]),
             torch.tensor([[-1., 0, 39], [0, 1, 0], [0, 0, 1]]), torch.tensor([[1., 0, 0], [0, -1, 23], [0, 0, 1]])]
    for M in cases:
        Mb = M[None].repeat(3, 1, 1)
        for mode in ('bilinear', 'nearest'):
            for pad in ('zeros', 'reflection'):
                got = U.warp_perspective_tensor(src.to(DEV), Mb.to(DEV), (24, 40), mode, pad).cpu()
                want = HA.warp_perspective(src, Mb, (24, 40), mode, pad)
                assert (got - want).abs().max().item() <= 1e-5, (M, mode, pad)
    got = U.WarpingModule()(src.to(DEV), torch.eye(3)[None].repeat(3, 1, 1).to(DEV), (24, 40))
    assert got.device.type == 'cuda' and torch.equal(got.cpu(), src)
    with pytest.raises(ValueError):
        U.warp_perspective_tensor(src.to(DEV), torch.eye(3)[None].to(DEV), (24, 40))
    with pytest.raises(ValueError):
        U.warp_perspective_tensor(src.to(DEV), torch.eye(3)[None].repeat(3, 1, 1), (24, 40), 'bicubic')


@pytest.mark.parametrize('H,W', [(64, 64), (120, 160)])
def test_warp_random_homographies(H, W):
    import multipoint_amd.utils as U
    torch.manual_seed(2)
    src = torch.rand(4, 1, H, W)
    homs = _homographies(3, 4, H, W, perspective_amplitude_x=0.2, perspective_amplitude_y=0.2, max_angle=1.57)
    M = torch.from_numpy(homs.astype(np.float32))
    for Mx in (M, torch.inverse(M)):
        for pad in ('zeros', 'reflection'):
            got = U.warp_perspective_tensor(src.to(DEV), Mx.to(DEV), (H, W), 'bilinear', pad).cpu()
            want = HA.warp_perspective(src, Mx, (H, W), 'bilinear', pad)
            assert (got - want).abs().max().item() <= 2e-3, pad
            assert (got - want).abs().mean().item() <= 2e-5
            got = U.warp_perspective_tensor(src.to(DEV), Mx.to(DEV), (H, W), 'nearest', pad).cpu()
            want = HA.warp_perspective(src, Mx, (H, W), 'nearest', pad)
            assert (got != want).float().mean().item() <= 2e-3, pad


@pytest.mark.parametrize('H,W', [(64, 64), (120, 160), (37, 53)])
def test_valid_mask(H, W):
    import multipoint_amd.utils as U
    homs = _homographies(5, 6, H, W, perspective_amplitude_x=0.2, perspective_amplitude_y=0.2, patch_ratio=0.85)
    for h in homs:
        for r, border in ((0, True), (3, True), (5, False), (16, True)):
            got = U.compute_valid_mask((H, W), h, r, border)
            want = HA.compute_valid_mask((H, W), h, r, border)
            assert got.dtype == np.float64 and got.shape == (H, W)
            assert np.array_equal(got, want), (r, border, int((got != want).sum()))
    from multipoint_amd import _lib
    with pytest.raises(ValueError):
        U.compute_valid_mask((H, W), homs[0], 17, True)
    assert _lib is not None


def test_gaussian_filter():
    import multipoint_amd.utils as U
    torch.manual_seed(3)
    p = torch.rand(3, 1, 40, 56)
    for k in (1, 3, 5, 9):
        got = U.gaussian_filter(p.to(DEV), k).cpu()
        assert (got - HA.smooth(p, k)).abs().max().item() <= 1e-6
    with pytest.raises(ValueError):
        U.gaussian_filter(p.to(DEV), 4)


def _run_case(oracle, golden, name, **kw):
    import multipoint_amd.models as M
    import multipoint_amd.utils as U
    pair, img_seed, _, hc = json.loads(str(golden['ha_cases']))[name]
    cfg = json.loads(str(golden['model_cfg_pair'])) if pair else dict(oracle.SHIPPED_MODEL_CONFIG)
    sd = oracle.make_weights(int(golden['weight_seed']), cfg)
    net = M.MultiPoint(dict(cfg)).eval()
    net.load_state_dict(sd)
    net.to(DEV)
    img = oracle.make_images(img_seed, 4 if pair else 2, 64, 64).to(DEV)
    hc = dict(hc, homographies=json.loads(str(golden['ha_homographies_cfg'])))
    homs = golden['ha_%s_homographies' % name]
    if pair:
        data = {'optical': {'image': img[:2], 'is_optical': torch.ones(2, 1, dtype=torch.bool, device=DEV)},
                'thermal': {'image': img[2:], 'is_optical': torch.zeros(2, 1, dtype=torch.bool, device=DEV)}}
        return U.homographic_adaptation_multispectral(data, net, hc, homographies=homs, **kw)
    return U.homographic_adaptation({'image': img}, net, hc, homographies=homs, **kw)


@pytest.mark.parametrize('name', ['single', 'single_filter', 'pair_prod', 'pair_sum'])
def test_driver_against_reference_golden(oracle, golden_dir, name):
    golden = np.load(os.path.join(golden_dir, 'homographic_adaptation.npz'))
    got = _run_case(oracle, golden, name).cpu()
    want = torch.from_numpy(golden['ha_%s_out' % name])
    assert got.shape == want.shape and got.device.type == 'cpu'
    diff = (got - want).abs()
    # pixels whose valid-count differs by one view (nearest sample on a mask edge) change by O(prob / count)
    assert (diff > 1e-4).float().mean().item() <= 5e-3, (name, (diff > 1e-4).float().mean().item())
    assert diff.median().item() <= 1e-6 and diff.mean().item() <= 2e-5, (name, diff.mean().item())
    assert ((got == 0) != (want == 0)).float().mean().item() <= 5e-3        # min_count zeroing
    # grouping of the views into (batched) forwards does not change a bit
    assert torch.equal(_run_case(oracle, golden, name, max_images=8).cpu(), got)
    # ... forwards of one or two images take the single-pair latency path (input channels of the small launches summed in
    # ranges, conv_wino43.hip SPLIT): another summation order, the same tolerance class
    two = _run_case(oracle, golden, name, max_images=2).cpu()
    assert (two - got).abs().max().item() <= 2e-5 and ((two == 0) != (got == 0)).float().mean().item() <= 1e-3


def test_driver_draws_the_reference_sequence(oracle, golden_dir):
    """without explicit matrices the driver consumes np.random exactly like the reference's loop"""
    import multipoint_amd.models as M
    import multipoint_amd.utils as U
    golden = np.load(os.path.join(golden_dir, 'homographic_adaptation.npz'))
    _, img_seed, rng_seed, hc = json.loads(str(golden['ha_cases']))['single']
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    net = M.MultiPoint(dict(cfg)).eval()
    net.load_state_dict(oracle.make_weights(int(golden['weight_seed']), cfg))
    net.to(DEV)
    img = oracle.make_images(img_seed, 2, 64, 64).to(DEV)
    hc = dict(hc, homographies=json.loads(str(golden['ha_homographies_cfg'])))
    np.random.seed(rng_seed)
    a = U.homographic_adaptation({'image': img}, net, hc)
    b = U.homographic_adaptation({'image': img}, net, hc, homographies=golden['ha_single_homographies'])
    assert torch.equal(a, b)


def test_identity_views_full_size():
    """480x640 (BASELINE size), size-independent property: with identity homographies every view reproduces the
    heat map, so out == prob wherever min_count is reached and 0 on the eroded border."""
    import multipoint_amd.models as M
    import multipoint_amd.utils as U
    net = M.MultiPoint({'multispectral': False, 'descriptor_size': 64}).eval()
    net.init_random_weights(1)
    net.to(DEV)
    torch.manual_seed(0)
    img = torch.rand(2, 1, 480, 640, device=DEV)
    p0 = net({'image': img})['prob']
    eye = np.stack([np.eye(3)] * 4)
    out = U.homographic_adaptation({'image': img}, net, {'num': 5, 'erosion_radius': 3, 'min_count': 2}, homographies=eye)
    inner = out[:, :, 3:-3, 3:-3]
    assert (inner - p0[:, :, 3:-3, 3:-3]).abs().max().item() <= 1e-6
    border = torch.ones_like(out, dtype=torch.bool)
    border[:, :, 3:-3, 3:-3] = False
    assert out[border].abs().max().item() == 0.0


def test_export_keypoints_cli(tmp_path, oracle):
    import yaml
    d = tmp_path / 'multipoint'
    d.mkdir()
    torch.save(oracle.make_weights(0, oracle.SHIPPED_MODEL_CONFIG), d / 'latest.model')
    with open(os.path.join(ROOT, 'model_weights', 'multipoint', 'params.yaml')) as f:
        (d / 'params.yaml').write_text(f.read())
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'configs', 'config_export_keypoints.yaml')))
    cfg['dataset'].update({'num_samples': 3, 'height': 64, 'width': 64})
    cfg['prediction'].update({'batchsize': 2})
    cfg['prediction']['homographic_adaptation'].update({'num': 4, 'min_count': 2})
    (tmp_path / 'cfg.yaml').write_text(yaml.safe_dump(cfg))
    out_file = tmp_path / 'labels.npz'
    cmd = [sys.executable, os.path.join(ROOT, 'export_keypoints.py'), '-y', str(tmp_path / 'cfg.yaml'), '-o', str(out_file),
           '-m', str(d), '-v', 'latest', '-s', '11']
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'of 3 samples' in out.stdout
    got = np.load(out_file)
    assert sorted(got.files) == ['synthetic_%06d/keypoints' % i for i in range(3)]
    # sample 0+1 (first batch) through the oracle with the homographies the seeded sampler draws
    from multipoint_amd.datasets import SyntheticPairs
    hc = cfg['prediction']['homographic_adaptation']
    homs = _homographies(11, 3, 64, 64, **hc['homographies'])
    sd = oracle.make_weights(0, oracle.SHIPPED_MODEL_CONFIG)
    pairs = [SyntheticPairs.make_pair(0, i, 64, 64) for i in range(2)]
    opt = torch.from_numpy(np.stack([p[0] for p in pairs])); thr = torch.from_numpy(np.stack([p[1] for p in pairs]))
    fwd = lambda i, x: oracle.forward(sd, x, oracle.SHIPPED_MODEL_CONFIG)['prob']
    ref, _ = HA.homographic_adaptation([opt, thr], fwd, hc, homs, aggregation='prod')
    ref_nms = oracle.box_nms(ref.numpy(), cfg['prediction']['nms'], cfg['prediction']['detection_threshold'])
    for i in range(2):
        k = got['synthetic_%06d/keypoints' % i]
        assert k.dtype == np.int64 and k.ndim == 2 and k.shape[1] == 2
        want = set(map(tuple, oracle.keypoints_from_map(ref_nms[i, 0], cfg['prediction']['detection_threshold']).tolist()))
        have = set(map(tuple, k.tolist()))
        assert len(want ^ have) <= max(2, len(want) // 50), (len(want), len(have), len(want ^ have))
    # -skip leaves processed samples alone
    out = subprocess.run(cmd + ['-skip'], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and 'of 0 samples' in out.stdout

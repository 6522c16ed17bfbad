"""Build-container only (needs /root/reference): the oracle against the IMPORTED reference on fresh
seeds.  Skipped on the GPU box; the committed golden vectors (test_oracle_golden.py) travel instead."""
import collections
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.reference


@pytest.fixture(scope='module')
def ref():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import ref_shim
    return ref_shim.install()


def test_state_dict_layout_and_forward(oracle, ref):
    models, utils = ref
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    net = models.MultiPoint(dict(cfg)).eval()
    spec = oracle.state_dict_spec(cfg)
    rsd = net.state_dict()
    assert [k for k, _, _ in spec] == list(rsd.keys())
    for k, s, d in spec:
        assert tuple(rsd[k].shape) == tuple(s) and rsd[k].dtype == d
    sd = oracle.make_weights(21, cfg)
    net.load_state_dict(sd)
    img = oracle.make_images(22, 2, 120, 160)
    with torch.no_grad():
        r = net({'image': img})
    o = oracle.forward(sd, img, cfg)
    assert (r['prob'] - o['prob']).abs().max().item() <= 1e-7
    assert (r['desc'] - o['desc']).abs().max().item() <= 1e-7


def test_multispectral_default_config(oracle, ref):
    models, utils = ref
    cfg = {'multispectral': True, 'descriptor_size': 128}
    net = models.MultiPoint(dict(cfg)).eval()
    spec = oracle.state_dict_spec(cfg)
    assert [k for k, _, _ in spec] == list(net.state_dict().keys())
    sd = oracle.make_weights(5, cfg)
    net.load_state_dict(sd)
    img = oracle.make_images(6, 4, 32, 40)
    flags = torch.tensor([[True], [False], [False], [True]])
    with torch.no_grad():
        r = net({'image': img, 'is_optical': flags})
    o = oracle.forward(sd, img, cfg, is_optical=flags)
    assert (r['prob'] - o['prob']).abs().max().item() <= 1e-7
    assert (r['desc'] - o['desc']).abs().max().item() <= 1e-7


def test_interpolate_and_matcher(oracle, ref):
    models, utils = ref
    rng = np.random.default_rng(1)
    desc = rng.standard_normal((64, 60, 80)).astype(np.float32)
    kp = np.stack([rng.integers(0, 480, 500), rng.integers(0, 640, 500)], 1).astype(np.int64)
    kp_t = torch.from_numpy(kp.copy())
    r = utils.interpolate_descriptors(kp_t, torch.from_numpy(desc), 480, 640).numpy()
    assert np.array_equal(kp_t.numpy(), kp)                      # reference does not mutate its input
    assert np.abs(oracle.interpolate_descriptors(kp, desc, 480, 640) - r).max() <= 1e-6
    d1 = r[:250]; d2 = r[200:]
    m = utils.NNMatcher(0.7).match(d1, d2)
    q, t, d = oracle.nn_match(d1, d2, 0.7)
    assert [x.queryIdx for x in m] == list(q) and [x.trainIdx for x in m] == list(t)
    assert np.allclose([x.distance for x in m], d, atol=1e-7)
    with pytest.raises(ValueError):
        utils.NNMatcher(-1.0)


def test_product_host_helpers_match_reference(ref):
    """dict_update / fix_model_weigth_keys / depth_to_space of the product package vs the reference."""
    models, utils = ref
    import multipoint_amd.utils as U
    a = {'a': 1, 'b': {'c': 2, 'd': 3}}; b = {'b': {'c': 5}, 'e': 6}
    import copy
    assert U.dict_update(copy.deepcopy(a), b) == utils.dict_update(copy.deepcopy(a), b)
    w = collections.OrderedDict([('module__encoder.1.weight', 1), ('x__y__detector.4.bias', 2), ('plain', 3)])
    assert list(U.fix_model_weigth_keys(w).items()) == list(utils.fix_model_weigth_keys(w).items())
    x = torch.randn(2, 64, 3, 4)
    assert torch.equal(U.depth_to_space(x, 8), utils.depth_to_space(x, 8))
    assert torch.equal(U.space_to_depth(U.depth_to_space(x, 8), 8), x)
    import multipoint_amd.models as M
    assert M.MultiPoint.default_config == models.MultiPoint.default_config
    for cfg in ({'multispectral': False, 'descriptor_size': 64}, {'bn_first': True}, {'final_batchnorm': False}):
        ours = M.MultiPoint(dict(cfg)).state_dict_spec()
        theirs = models.MultiPoint(dict(cfg)).state_dict()
        assert [k for k, _, _ in ours] == list(theirs.keys())
        assert all(tuple(theirs[k].shape) == tuple(s) for k, s, _ in ours)


def test_magicleap_vs_reference(oracle, ref):
    models, utils = ref
    net = models.SuperPointMagicLeap().eval()
    sd = oracle.make_weights_magicleap(8)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    img = oracle.make_images(9, 2, 64, 96)
    with torch.no_grad():
        r = net({'image': img})
    o = oracle.forward_magicleap(sd, img)
    for k in ('logits', 'desc', 'prob'):
        assert (r[k] - o[k]).abs().max().item() <= 1e-6
    import multipoint_amd.models as M
    assert [k for k, _, _ in M.SuperPointMagicLeap().state_dict_spec()] == list(sd.keys())

"""CPU: host-side logic of the product package and the C-ABI library surface (no compute calls:
there is no GPU here).  Also checks that the product never falls back to a CPU path."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    from multipoint_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, 'include', 'multipoint_hip.h')).read()
    declared = set(re.findall(r'\b(mp_[a-z_0-9]+)\s*\(', header))
    assert {'mp_create', 'mp_forward', 'mp_box_nms', 'mp_detect_keypoints', 'mp_sample_descriptors',
            'mp_match_mutual_nn', 'mp_load_weights'} <= declared
    dll = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(dll, name), 'library does not export %s' % name
    assert declared == set(lib.SIGNATURES), 'ctypes binding and header disagree'
    assert b'gfx950' in lib.load_library().mp_version()


def test_library_is_gfx950_only():
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler', '--list', '--type=o',
                          '--input=%s' % os.path.join(ROOT, 'multipoint_amd', 'libmultipoint_hip.so')],
                         capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        assert 'gfx950' in out.stdout and 'gfx942' not in out.stdout and 'sm_' not in out.stdout


def test_no_cpu_fallback(lib):
    """Without a GPU every compute entry raises; nothing silently routes to the oracle / torch CPU."""
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import multipoint_amd.models as M
    import multipoint_amd.utils as U
    net = M.MultiPoint({'multispectral': False, 'descriptor_size': 64})
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net.to('cuda')
    with pytest.raises(RuntimeError):
        U.box_nms(torch.rand(16, 16), 4, 0.015)
    with pytest.raises(RuntimeError):
        U.get_matches(np.eye(4, 64, dtype=np.float32), np.eye(4, 64, dtype=np.float32), 'nnmatcher')
    with pytest.raises(RuntimeError):
        U.interpolate_descriptors(torch.zeros(3, 2, dtype=torch.int64), torch.rand(64, 4, 4), 32, 32)
    h = ctypes.c_void_p()
    rc = lib.load_library().mp_create(ctypes.byref(h), 0)
    assert rc != 0 and not h
    assert lib.load_library().mp_last_error(None)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'multipoint_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('test_oracle', ''), '%s mentions the oracle' % f
    for f in ('predict_align_image_pair.py', 'predict_keypoints.py'):
        assert 'oracle' not in open(os.path.join(ROOT, f)).read()


def test_state_dict_layout_matches_oracle_spec(oracle):
    import multipoint_amd.models as M
    for cfg in (oracle.SHIPPED_MODEL_CONFIG, {'multispectral': True}, {'bn_first': True, 'multispectral': False},
                {'final_batchnorm': False, 'descriptor_head': False}, {'double_convolution': False, 'bn_first': True}):
        ours = [(k, tuple(s), d) for k, s, d in M.MultiPoint(dict(cfg)).state_dict_spec()]
        theirs = [(k, tuple(s), d) for k, s, d in oracle.state_dict_spec(cfg)]
        assert ours == theirs
    n = sum(int(np.prod(s)) if len(s) else 1 for _, s, _ in M.MultiPoint(dict(oracle.SHIPPED_MODEL_CONFIG)).state_dict_spec())
    assert n == 1257169                                           # state-dict elements


def test_load_state_dict_is_strict(oracle):
    import multipoint_amd.models as M
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    sd = oracle.make_weights(0, cfg)
    net = M.MultiPoint(cfg)
    net.load_state_dict(sd)
    assert list(net.state_dict().keys()) == list(sd.keys())
    bad = dict(sd); bad.pop('encoder.5.bias')
    with pytest.raises(RuntimeError, match='Missing key'):
        M.MultiPoint(cfg).load_state_dict(bad)
    bad = dict(sd); bad['encoder.99.weight'] = torch.zeros(1)
    with pytest.raises(RuntimeError, match='Unexpected key'):
        M.MultiPoint(cfg).load_state_dict(bad)
    bad = dict(sd); bad['encoder.5.weight'] = torch.zeros(64, 32, 3, 3)
    with pytest.raises(RuntimeError, match='size mismatch'):
        M.MultiPoint(cfg).load_state_dict(bad)
    # prefixed keys as written by DataParallel checkpoints are fixed by fix_model_weigth_keys
    import multipoint_amd.utils as U
    pre = {'module__' + k: v for k, v in sd.items()}
    M.MultiPoint(cfg).load_state_dict(U.fix_model_weigth_keys(pre))
    with pytest.raises(ValueError):
        M.MultiPoint({'channel_version': 3})
    # double_convolution: false (MultiPoint.py:144-148): the reference module's own key list -- fp32 path only
    single = dict(cfg); single['double_convolution'] = False
    assert [k for k, _, _ in M.MultiPoint(single).state_dict_spec()] == list(oracle.make_weights(0, single).keys())
    assert 'encoder.16.weight' in oracle.make_weights(0, single) and 'encoder.19.weight' not in oracle.make_weights(0, single)
    # MultiPoint.forward wraps ANY config in autocast (MultiPoint.py:99-104): mixed_precision is accepted with every model config
    for upd in ({'double_convolution': False}, {'channel_version': 1}, {'channel_version': 2, 'double_convolution': False}):
        assert M.MultiPoint(dict(upd, mixed_precision=True)).config['mixed_precision'] is True
    with pytest.raises(ValueError):
        net.set_force_return_logits(1)
    with pytest.raises(NotImplementedError):
        net.train()
    rnd = M.MultiPoint(cfg).init_random_weights(3).state_dict()
    assert list(rnd.keys()) == list(sd.keys())


def test_argument_errors_mirror_reference():
    import multipoint_amd.utils as U
    with pytest.raises(ValueError, match='either 2D'):
        U.box_nms(torch.rand(2, 16, 16), 4, 0.015)                # utils.py:90-91
    with pytest.raises(ValueError, match='unknown matching method'):
        U.get_matches(np.zeros((2, 64), np.float32), np.zeros((2, 64), np.float32), 'nope')   # matching.py:18-19
    with pytest.raises(ValueError, match='non-negative'):
        U.get_matches(np.zeros((2, 64), np.float32), np.zeros((2, 64), np.float32), 'nnmatcher', threshold=-0.1)
    with pytest.raises(NotImplementedError):
        U.get_matches(np.zeros((2, 64), np.float32), np.zeros((2, 64), np.float32), 'flann')
    with pytest.raises(ValueError):          # OpenCV: crossCheck supports k = 1 only
        U.get_matches(np.zeros((2, 64), np.float32), np.zeros((2, 64), np.float32), 'bfmatcher', True, crossCheck=True)
    with pytest.raises(AttributeError):      # NNMatcher has no knnMatch (matching.py:21)
        U.get_matches(np.zeros((2, 64), np.float32), np.zeros((2, 64), np.float32), 'nnmatcher', True)
    with pytest.raises(ValueError, match='non-negative'):
        U.get_matches(np.zeros((2, 64), np.float32), np.zeros((2, 64), np.float32), 'thresholdmatcher', threshold=-1.0)
    assert U.get_matches(np.zeros((0, 64), np.float32), np.zeros((2, 64), np.float32), 'thresholdmatcher') == []
    assert U.get_matches(np.zeros((0, 64), np.float32), np.zeros((2, 64), np.float32), 'nnmatcher') == []   # matching.py:46-47


def test_plumbing_helpers():
    import multipoint_amd.utils as U
    d = {'a': torch.zeros(2), 'b': {'c': torch.ones(3), 'name': 'x'}}
    out = U.data_unsqueeze(U.data_to_device(d, 'cpu'), 0)
    assert out['a'].shape == (1, 2) and out['b']['c'].shape == (1, 3) and out['b']['name'] == 'x'
    assert U.dict_update({'a': {'b': 1, 'c': 2}}, {'a': {'b': 3}, 'd': 4}) == {'a': {'b': 3, 'c': 2}, 'd': 4}


def test_synthetic_dataset_schema():
    from multipoint_amd.datasets import SyntheticPairs, ImagePairDataset
    ds = SyntheticPairs({'num_samples': 3, 'height': 64, 'width': 96})
    s = ds[1]
    assert ds.returns_pair() and len(ds) == 3
    for side, flag in (('optical', True), ('thermal', False)):
        e = s[side]
        assert e['image'].shape == (1, 64, 96) and e['image'].dtype == torch.float32
        assert e['valid_mask'].dtype == torch.bool and e['valid_mask'].all()
        assert bool(e['is_optical'][0]) is flag
    assert torch.equal(ds[1]['optical']['image'], s['optical']['image'])         # deterministic
    assert not torch.equal(ds[0]['optical']['image'], s['optical']['image'])
    with pytest.raises(IndexError):
        ds[3]
    with pytest.raises(ValueError):
        SyntheticPairs({'height': 60, 'width': 64})
    with pytest.raises(ValueError):
        ImagePairDataset({'filename': None})
    loader = torch.utils.data.DataLoader(ds, batch_size=2)
    b = next(iter(loader))
    assert b['optical']['image'].shape == (2, 1, 64, 96) and b['thermal']['is_optical'].shape == (2, 1)


def test_cli_surface():
    for script, flags in (('predict_align_image_pair.py', ['-y', '-m', '-v', '-i', '-r', '-p', '-e', '-tk', '-th', '-s']),
                          ('predict_keypoints.py', ['-y', '-m', '-v', '-i', '-r', '-p', '-e', '-b', '-t', '-mask', '-s'])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, script), '--help'], capture_output=True, text=True, cwd=ROOT)
        assert out.returncode == 0
        for f in flags:
            assert re.search(r'(^|[\s\[])%s\b' % re.escape(f), out.stdout), (script, f)
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'configs', 'config_image_pair_dataset_prediction.yaml')))
    assert set(cfg['prediction']) >= {'allow_gpu', 'num_worker', 'batchsize', 'detection_threshold', 'nms', 'cpu_nms',
                                      'topk', 'reprojection_threshold', 'matching'}
    assert set(cfg['prediction']['matching']) == {'method', 'method_kwargs', 'knn_matches'}
    params = yaml.safe_load(open(os.path.join(ROOT, 'model_weights', 'multipoint', 'params.yaml')))['model']
    assert params['type'] == 'MultiPoint' and params['descriptor_size'] == 64 and params['multispectral'] is False


def test_image_pair_dataset_npz_schema(tmp_path):
    """ImagePairDataset.py:88-243 without augmentation (no GPU needed): crop, label maps, random_pairs, names."""
    import random
    import numpy as np
    from multipoint_amd.datasets import ImagePairDataset
    rng = np.random.default_rng(0)
    arrays, labels = {}, {}
    for i in range(2):
        arrays['p%d/optical' % i] = rng.random((40, 56), dtype=np.float32)
        arrays['p%d/thermal' % i] = rng.random((40, 56), dtype=np.float32)
        arrays['p%d/thermal_raw' % i] = rng.random((40, 56), dtype=np.float32)
        labels['p%d/keypoints' % i] = np.stack([rng.integers(0, 40, 25), rng.integers(0, 56, 25)], axis=1)
    fn, kfn = str(tmp_path / 'd.npz'), str(tmp_path / 'k.npz')
    np.savez(fn, **arrays); np.savez(kfn, **labels)
    ds = ImagePairDataset({'filename': fn, 'keypoints_filename': kfn, 'single_image': False})
    s = ds[1]
    assert len(ds) == 2 and ds.get_name(1) == 'p1' and s['name'] == 'p1' and ds.returns_pair()
    assert np.array_equal(s['optical']['image'][0].numpy(), arrays['p1/optical'])
    assert np.array_equal(s['thermal']['image'][0].numpy(), arrays['p1/thermal'])
    assert s['optical']['valid_mask'].all() and s['optical']['valid_mask'].dtype == torch.bool
    assert 'homography' not in s['optical']
    km = np.zeros((40, 56), bool); km[labels['p1/keypoints'][:, 0], labels['p1/keypoints'][:, 1]] = True
    assert np.array_equal(s['optical']['keypoints'].numpy(), km) and np.array_equal(s['thermal']['keypoints'].numpy(), km)
    # raw thermal + crop: the same random.randint draws as the reference (ImagePairDataset.py:122-123)
    dc = ImagePairDataset({'filename': fn, 'keypoints_filename': kfn, 'single_image': False, 'raw_thermal': True,
                           'height': 24, 'width': 32})
    random.seed(3); s = dc[0]
    random.seed(3); i_h = random.randint(0, 16); i_w = random.randint(0, 24)
    assert np.array_equal(s['thermal']['image'][0].numpy(), arrays['p0/thermal_raw'][i_h:i_h + 24, i_w:i_w + 32])
    k = labels['p0/keypoints'] - np.array([[i_h, i_w]])
    k = k[(k[:, 0] >= 0) & (k[:, 0] < 24) & (k[:, 1] >= 0) & (k[:, 1] < 32)]
    km = np.zeros((24, 32), bool); km[k[:, 0], k[:, 1]] = True
    assert np.array_equal(s['optical']['keypoints'].numpy(), km)
    # single image: one spectrum chosen by random.randint(0, 1)
    d1 = ImagePairDataset({'filename': fn, 'single_image': True})
    random.seed(5); s = d1[0]
    random.seed(5); is_optical = bool(random.randint(0, 1))
    assert bool(s['is_optical'][0]) is is_optical and not d1.returns_pair() and 'keypoints' not in s
    assert np.array_equal(s['image'][0].numpy(), arrays['p0/optical' if is_optical else 'p0/thermal'])
    # errors of the reference
    with pytest.raises(ValueError):
        ImagePairDataset({'filename': fn, 'single_image': False, 'height': 100})[0]
    with pytest.raises(IndexError):
        np.savez(str(tmp_path / 'k2.npz'), **{'p0/keypoints': labels['p0/keypoints']})
        ImagePairDataset({'filename': fn, 'keypoints_filename': str(tmp_path / 'k2.npz')})
    with pytest.raises(NotImplementedError):
        ImagePairDataset({'filename': fn, 'augmentation': {'photometric': {'enable': True}}})
    # the homographic augmentation computes on the GPU: no CPU fallback
    if not torch.cuda.is_available():
        da = ImagePairDataset({'filename': fn, 'single_image': False,
                               'augmentation': {'homographic': {'enable': True, 'params': {}}}})
        with pytest.raises((RuntimeError, AssertionError)):
            da[0]


def test_loader_workers_follow_gpu_use():
    import multipoint_amd.datasets as d
    aug = d.SyntheticPairs({'num_samples': 2, 'height': 64, 'width': 96, 'augmentation': {'homographic': {'enable': True}}})
    assert d.loader_num_workers(aug, 4) == 0 and d.loader_num_workers(aug, 0) == 0
    assert d.loader_num_workers(d.SyntheticPairs({}), 4) == 4


def test_c_host_demo_compiles_against_the_header(tmp_path):
    """examples/c_host_demo.c is plain C99 over include/multipoint_hip.h: it must compile (-Wall -Werror) and link
    against the in-tree library without any Python / torch header (run on the GPU box by tests/test_gpu_c_abi.py)."""
    lib = os.path.join(ROOT, 'multipoint_amd', 'libmultipoint_hip.so')
    if not os.path.exists(lib):
        pytest.skip('library not built')
    out = subprocess.run(['gcc', '-std=c99', '-O2', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include',
                          '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'c_host_demo.c'),
                          '-L' + os.path.join(ROOT, 'multipoint_amd'), '-lmultipoint_hip', '-L/opt/rocm/lib', '-lamdhip64',
                          '-o', str(tmp_path / 'demo')], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    src = open(os.path.join(ROOT, 'examples', 'c_host_demo.c')).read()
    includes = [l for l in src.splitlines() if l.startswith('#include')]
    assert includes and not any('torch' in l or 'Python' in l or 'pybind' in l for l in includes)


def test_committed_bench_line_follows_the_contract():
    """profiles/r01_bench_n1.json is the line `python bench.py` printed on the GPU box: the fields the driver parses."""
    import json
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r01_bench_n1.json')))
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert d['metric'].split(' @')[0] == base['metric'].split(' @')[0] and '480x640' in d['metric']
    assert d['unit'] == 'image-pairs/s' and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['n_gpus'] == 1 and d['steps'] > 0 and d['warmup'] >= 0 and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 32 * 1e3 / d['ms_per_step']) / d['value'] < 1e-3          # pairs per step / time per step
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and r['peak'] == 157.3
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert abs(r['achieved'] - r['flop_per_launch'] / (r['ms_per_launch'] * 1e-3) / 1e12) < 0.05
    assert r['traffic'] is None or r['traffic'] > 0
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == d['unit'] and c['sample']
    assert d['parity']['desc_max_abs_err'] <= 1e-4


def test_committed_round6_line_carries_the_exact_entry_and_the_upload():
    """profiles/r06_bench_n1.json, the line of this round's collection: the contract fields, and the `secondary` rates the round-5
    verdict asked to see under the driver's clock -- PairPipeline.run_converged at the headline shape and the PCIe-inclusive rate --
    next to c5 / direct / batch1; `converged` within 3 % of `value`."""
    import json
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r06_bench_n1.json')))
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert d['metric'].split(' @')[0] == base['metric'].split(' @')[0] and d['config']['workload'].startswith('BASELINE configs[2]')
    assert d['n_gpus'] == 1 and d['dtype'] == 'f32' and d['input'] == 'resident in HBM' and d['vs_baseline'] is None
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['peak'] == 157.3 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and r['traffic'] > 0
    sec = d['secondary']
    assert {'converged', 'host_input', 'c5', 'direct', 'batch1'} <= set(sec)
    assert sec['converged']['pairs_per_s'] >= 0.97 * d['value'] and sec['host_input']['pairs_per_s'] >= 0.95 * d['value']
    assert 'run_converged' in sec['converged']['workload'] and 'pinned host memory' in sec['host_input']['workload']
    assert d['parity']['unexplained_keypoints'] == 0 and d['parity']['keypoints_differing'] <= 0.002 * d['parity']['keypoints_total']
    assert d['parity']['structured']['auto']['keypoints_differing'] == 0
    assert d['cpu_baseline']['value'] > 0 and d['cpu_baseline']['kind'] in ('reference', 'port')


def test_bench_self_launches_ranks():
    """`python bench.py --gpus 2` outside torchrun starts the two ranks itself (child torchrun job, 127.0.0.1 rendezvous)
    and fails only because this container has no GPU -- in both ranks, not at a launcher check; a WORLD_SIZE that does
    not match --gpus is an error."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    out = r.stdout + r.stderr
    # every rank that got as far as its device check says so (torchrun tears the job down at the FIRST failing rank, so the second
    # message is not guaranteed), and the failure is reported by the torchrun child job, not by a launcher-side check
    assert 1 <= out.count('bench.py needs an MI355X') <= 2
    assert 'torch.distributed.elastic' in out or 'ChildFailedError' in out
    env.update(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4'], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and 'does not match WORLD_SIZE=2' in (r.stdout + r.stderr)


def test_build_checks_dma_hazards(tmp_path):
    """multipoint_amd/build.py refuses generated code in which a VALU write of an SGPR (spill reload / readfirstlane) sits
    right in front of an LDS-DMA that uses it as its base (the hardware needs five wait states, hipcc pads nothing inside an
    asm statement)."""
    from multipoint_amd import build as b
    ok = tmp_path / 'ok.s'
    ok.write_text('\ts_add_u32 s12, s11, s2\n\ts_addc_u32 s13, s1, s3\n\ts_mov_b32 m0, s17\n\ts_nop 0\n'
                  '\tglobal_load_lds_dwordx4 v48, s[12:13]\n')
    assert b.check_dma_hazards(str(ok)) == 1
    bad = tmp_path / 'bad.s'
    bad.write_text('\tv_readlane_b32 s12, v90, 3\n\ts_mov_b32 m0, s17\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v48, s[12:13]\n')
    with pytest.raises(RuntimeError, match='base SGPR'):
        b.check_dma_hazards(str(bad))
    far = tmp_path / 'far.s'
    far.write_text('\tv_readlane_b32 s12, v90, 3\n' + '\ts_nop 0\n' * 9 + '\tglobal_load_lds_dwordx4 v48, s[12:13]\n')
    assert b.check_dma_hazards(str(far)) == 1
    nop = tmp_path / 'nop.s'                             # s_nop N is N + 1 wait states: 1 + 4 = 5 suffice, 1 + 3 do not
    nop.write_text('\tv_readlane_b32 s13, v90, 3\n\ts_mov_b32 m0, s17\n\ts_nop 3\n\tglobal_load_lds_dwordx4 v48, s[12:13]\n')
    assert b.check_dma_hazards(str(nop)) == 1
    nop.write_text('\tv_readlane_b32 s13, v90, 3\n\ts_mov_b32 m0, s17\n\ts_nop 2\n\tglobal_load_lds_dwordx4 v48, s[12:13]\n')
    with pytest.raises(RuntimeError, match='base SGPR'):
        b.check_dma_hazards(str(nop))
    none = tmp_path / 'none.s'
    none.write_text('\ts_nop 0\n')
    with pytest.raises(RuntimeError, match='no global_load_lds'):
        b.check_dma_hazards(str(none))
    # a DMA fewer than five wait states behind a branch target: a predecessor block the linear walk cannot see may end in
    # the hazardous write, so the wait states must lie inside the DMA's own block
    lab = tmp_path / 'label.s'
    lab.write_text('\tv_readlane_b32 s12, v90, 3\n\ts_cbranch_scc1 .LBB3_7\n' + '\ts_nop 0\n' * 8 + '.LBB3_7:\n\ts_mov_b32 m0, s17\n'
                   '\tglobal_load_lds_dwordx4 v48, s[12:13]\n')
    with pytest.raises(RuntimeError, match='branch target'):
        b.check_dma_hazards(str(lab))
    lab.write_text('.LBB3_7:\n\ts_mov_b32 m0, s17\n\ts_nop 3\n\tglobal_load_lds_dwordx4 v48, s[12:13]\n')
    assert b.check_dma_hazards(str(lab)) == 1


def test_build_refuses_scratch_in_convolution_kernels(tmp_path, monkeypatch):
    """multipoint_amd/build.py reads every convolution kernel's scratch bytes from the generated code-object metadata and fails the
    build above the source's cap (0 for the kernels the benchmark shapes run; round-5 verdict): a spilled register's reload waits for
    every load in flight."""
    from multipoint_amd import build as b
    meta = ('amdhsa.kernels:\n  - .args: []\n    .name:           _Zk1\n    .private_segment_fixed_size: 0\n    .vgpr_count: 200\n'
            '  - .args: []\n    .name:           _Zk2\n    .private_segment_fixed_size: %d\n    .vgpr_count: 256\n')
    f = tmp_path / 'k.s'
    f.write_text(meta % 0)
    assert b.check_scratch(str(f), 0) == {}
    f.write_text(meta % 24)
    assert b.check_scratch(str(f), 64) == {'_Zk2': 24}
    with pytest.raises(RuntimeError, match='_Zk2 24 B'):
        b.check_scratch(str(f), 0)
    f.write_text('\ts_nop 0\n')
    with pytest.raises(RuntimeError, match='no kernel metadata'):
        b.check_scratch(str(f), 0)
    # the kernels of every headline / c5 launch are capped at zero
    for src in ('conv_wino43.hip', 'conv_f16_res.hip', 'conv_mfma.hip', 'head_tail.hip', 'head_tail_f16.hip'):
        assert b.SCRATCH_CAPS[src] == 0
    # and a compile that exceeds its cap leaves no object behind
    monkeypatch.setattr(b, 'OBJ_DIR', str(tmp_path)); monkeypatch.setattr(b, 'CSRC', str(tmp_path))
    monkeypatch.setattr(b, 'DMA_SOURCES', ()); monkeypatch.setattr(b, 'SCRATCH_CAPS', {'s.hip': 0})
    (tmp_path / 's.hip').write_text('#include <hip/hip_runtime.h>\n__global__ void s(float* p, int n) { float a[64]; for (int i = 0; i < 64; ++i) '
                                    'a[i] = p[i * n]; float t = 0; for (int i = 0; i < 64; ++i) t += a[(i * n) & 63]; p[threadIdx.x] = t; }\n')
    with pytest.raises(RuntimeError, match='more scratch'):
        b._compile('s.hip')
    assert [x for x in os.listdir(tmp_path) if x not in ('s.hip', 'k.s')] == []


def test_failed_hazard_check_leaves_no_object(tmp_path, monkeypatch):
    """A source whose generated code fails the LDS-DMA hazard check must not leave `<stem>.o` behind: the next build would
    see it as fresh, skip compile + check and link the hazardous object (round-2 advisor finding)."""
    from multipoint_amd import build as b
    monkeypatch.setattr(b, 'OBJ_DIR', str(tmp_path))
    monkeypatch.setattr(b, 'CSRC', str(tmp_path))
    monkeypatch.setattr(b, 'DMA_SOURCES', ('k.hip',))
    (tmp_path / 'k.hip').write_text('#include <hip/hip_runtime.h>\n__global__ void k(float* p) { p[threadIdx.x] = 1.0f; }\n')
    with pytest.raises(RuntimeError, match='no global_load_lds'):         # the check fails (no DMA in this file at all)
        b._compile('k.hip')
    assert [f for f in os.listdir(tmp_path) if f != 'k.hip'] == []        # no object, no -save-temps leftovers
    monkeypatch.setattr(b, 'DMA_SOURCES', ())
    assert os.path.exists(b._compile('k.hip'))


def test_hardware_queue_default_is_set_by_the_multi_gpu_module_only():
    """GPU_MAX_HW_QUEUES must be in the environment before the runtime initialises for every user of the MULTI-GPU path (an RCCL
    communicator is what makes the default of 4 queues hurt): importing multipoint_amd.dist sets it (an explicit setting wins),
    importing the package alone leaves the host application's runtime configuration alone."""
    cwd = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = lambda code: subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=cwd, timeout=300)
    r = run('import os; os.environ.pop("GPU_MAX_HW_QUEUES", None); import multipoint_amd; print(os.environ.get("GPU_MAX_HW_QUEUES"))')
    assert r.returncode == 0 and r.stdout.strip() == 'None', r.stderr[-500:]
    r = run('import os; os.environ.pop("GPU_MAX_HW_QUEUES", None); import multipoint_amd.dist; print(os.environ["GPU_MAX_HW_QUEUES"])')
    assert r.returncode == 0 and r.stdout.strip() == '8', r.stderr[-500:]
    r = run('import os; os.environ["GPU_MAX_HW_QUEUES"] = "2"; import multipoint_amd.dist; print(os.environ["GPU_MAX_HW_QUEUES"])')
    assert r.returncode == 0 and r.stdout.strip() == '2', r.stderr[-500:]


def test_bench_traffic_figure_only_next_to_the_kernel_it_was_measured_on(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from the committed PMC summary of the NEWEST round, and only when that summary is for the
    kernel instantiation the run timed: an older round's counters, or another instantiation's, must give null -- never a number."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    prof = tmp_path / 'profiles'
    prof.mkdir()
    (prof / 'r02_pmc_hbm_traffic.json').write_text(json.dumps(
        {'dominant_kernel': 'k<true, 8>', 'dominant_kernel_mean_traffic_bytes_per_launch': 14.2e9}))
    (prof / 'r03_pmc_hbm_traffic.json').write_text(json.dumps(
        {'dominant_kernel': 'k<true, 8, false>', 'dominant_kernel_mean_traffic_bytes_per_launch': 1.3e9}))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    assert bench.pmc_traffic('c3', 'k<true,8,false>') == 1.3e9          # spaces do not matter
    assert bench.pmc_traffic('c3', 'k<true,8>') is None                 # the OLD round's kernel: no fall-through to r02
    assert bench.pmc_traffic('c5', 'k<true,8,false>') is None           # no file for that workload
    (prof / 'r03_pmc_hbm_traffic.json').write_text('{"dominant_kernel": "k<true, 8, false>"}')
    assert bench.pmc_traffic('c3', 'k<true,8,false>') is None           # a broken newest file does not expose the older one


def test_conv_algorithm_is_a_model_setting():
    """model.conv_algorithm (multipoint_amd only: the algorithm of the 3x3 convolutions) is validated on the host and is NOT part of
    default_config, which stays equal to the reference's (tests/test_oracle_vs_reference.py)."""
    import multipoint_amd.models as M
    assert 'conv_algorithm' not in M.MultiPoint.default_config
    assert M.MultiPoint({'conv_algorithm': 'direct'}).config['conv_algorithm'] == 'direct'
    assert set(M.MultiPoint.CONV_ALGORITHMS) == {'auto', 'winograd43', 'winograd43_general', 'direct'}
    with pytest.raises(ValueError):
        M.MultiPoint({'conv_algorithm': 'winograd22'})
    assert M.SuperPointMagicLeap({'conv_algorithm': 'direct'}).config['conv_algorithm'] == 'direct'
    # model.batch_invariant: the other multipoint_amd-only key, same rules
    assert 'batch_invariant' not in M.MultiPoint.default_config
    assert M.MultiPoint({'batch_invariant': True}).config['batch_invariant'] is True
    assert M.SuperPointMagicLeap({'batch_invariant': True}).config['batch_invariant'] is True
    from multipoint_amd import _lib
    assert [n for n, _ in _lib.ModelConfig._fields_][-2:] == ['conv_algorithm', 'batch_invariant']


def test_mp_debug_switch_parser(monkeypatch):
    """MP_DEBUG is ONE comma-separated list of `key` / `key=value` developer switches (csrc/api.hip::debug_switch reads the kernel
    selection keys in mp_create; the Python side reads its own -- pipeline stream priorities, post_overlap -- with the same grammar)."""
    from multipoint_amd import _lib
    monkeypatch.delenv('MP_DEBUG', raising=False)
    assert _lib.debug_switch('no_winograd') is None and _lib.debug_switch('wino43', '2') == '2'
    monkeypatch.setenv('MP_DEBUG', 'no_winograd, wino43_gen=2,post_overlap=0 ,ncu=64')
    assert _lib.debug_switch('no_winograd') == '1'
    assert _lib.debug_switch('wino43_gen') == '2' and _lib.debug_switch('wino43') is None      # a key is a whole token, not a prefix
    assert _lib.debug_switch('post_overlap', '1') == '0' and _lib.debug_switch('ncu') == '64'
    assert _lib.debug_switch('fwd_priority', '-1') == '-1'
    # every switch the C side documents is spelled the same way in its parser and in INTEGRATION.md
    api = open(os.path.join(ROOT, 'multipoint_amd', 'csrc', 'api.hip')).read()
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    import re
    keys = set(re.findall(r'debug_switch\("([a-z0-9_]+)"', api))
    assert {'no_winograd', 'wino43', 'wino43_gen', 'no_fuse', 'no_fuse43', 'no_head_fuse', 'no_vin', 'no_planar', 'no_persist',
            'persist_min_items', 'splitk_max', 'f16_no_res', 'f16_no_fuse1', 'f16_res_groups', 'ncu', 'nxcd'} <= keys
    for k in keys:
        assert k in doc, 'MP_DEBUG key %s is not documented in INTEGRATION.md' % k


def test_tie_robust_redo_replaces_only_the_flagged_rows():
    """utils.tie_robust_redo (host logic of the top-k tie guard, CPU tensors + a stand-in model): the rows of prob / desc of the
    flagged images -- and only those -- are replaced in place by the twin model's forward of exactly those images (with their
    is_optical flags); models without a second algorithm (direct_twin() -> None) are left alone."""
    import torch
    from multipoint_amd.utils import utils as U
    calls = []

    class Twin:
        def __call__(self, data):
            calls.append({k: v.clone() for k, v in data.items()})
            n = data['image'].shape[0]
            return {'prob': data['image'] * 0 + 7.0, 'logits': None, 'desc': torch.full((n, 4, 2, 2), 9.0)}

    class Net:
        def __init__(self, twin): self._t = twin
        def direct_twin(self): return self._t

    img = torch.arange(5 * 16, dtype=torch.float32).reshape(5, 1, 4, 4)
    flags_in = torch.tensor([[True], [False], [True], [False], [True]])
    out = {'prob': img.clone(), 'logits': None, 'desc': torch.zeros(5, 4, 2, 2)}
    n = U.tie_robust_redo(Net(Twin()), {'image': img, 'is_optical': flags_in}, out, [False, True, False, True, False])
    assert n == 2 and len(calls) == 1
    assert torch.equal(calls[0]['image'], img[[1, 3]]) and torch.equal(calls[0]['is_optical'], flags_in[[1, 3]])
    for b in range(5):
        if b in (1, 3):
            assert bool((out['prob'][b] == 7.0).all()) and bool((out['desc'][b] == 9.0).all())
        else:
            assert torch.equal(out['prob'][b], img[b]) and bool((out['desc'][b] == 0).all())
    assert U.tie_robust_redo(Net(Twin()), {'image': img}, out, [False] * 5) == 0 and len(calls) == 1      # nothing flagged: no forward
    assert U.tie_robust_redo(Net(None), {'image': img}, out, [True] * 5) == 0                               # no second algorithm


@pytest.mark.parametrize('bn_first', [False, True])
def test_max_pool_commutes_with_the_fp16_activation(bn_first):
    """The identity the fp16 pooled epilogues rest on (conv_f16.hip / conv_f16_res.hip, `pool_first`), checked in torch's own half
    arithmetic: with f(x) = fp16(BN(ReLU(fp16(x + bias)))) (MultiPoint.py:143-148 under autocast; bn_first: ReLU behind the
    BatchNorm) the maximum of a 2x2 window's four activations equals f(max of the four accumulators) for channels whose BatchNorm
    scale is >= 0 and f(min) where it is negative -- bit for bit, ties, zeros and saturating values included -- because every step
    of f is monotonic.  (A fused multiply-add that rounds ONCE to fp16 would break it: the kernels pin the fp32 intermediate.)"""
    g = torch.Generator().manual_seed(5)
    n = 400000
    x = torch.randn(n, 4, generator=g) * torch.tensor([0.01, 1.0, 30.0, 3000.0])[torch.randint(0, 4, (n, 1), generator=g)]
    x[: n // 8] = x[: n // 8].round()                       # exact ties between window members
    x[n // 8: n // 4, 1] = x[n // 8: n // 4, 0]
    bias = torch.randn(n, 1, generator=g).half().float()
    scale = torch.randn(n, 1, generator=g) * 2.0
    scale[: n // 16] = 0.0
    shift = torch.randn(n, 1, generator=g)

    def act(v):
        h = (v + bias).half()
        if not bn_first:
            h = torch.relu(h)
        y = (h.float() * scale + shift).half()              # fp32 affine, THEN fp16: two roundings
        return torch.relu(y) if bn_first else y

    after = act(x).max(dim=1, keepdim=True).values
    pick = torch.where(scale < 0, x.min(dim=1, keepdim=True).values, x.max(dim=1, keepdim=True).values)
    first = act(pick)
    assert torch.equal(after, first)
    assert int((scale < 0).sum()) > n // 3 and int((x[:, 0] == x[:, 1]).sum()) > n // 16


def _fma32(a, b, c):
    """fp32 fused multiply-add: the product of two fp32 values is exact in fp64; the one fp64 addition is rounded to fp32 (the same
    emulation on both sides of the identity below, and monotonic in every argument like the hardware's)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _relu_bits(x):
    """The kernels' ReLU: an INTEGER max of the float's bits with 0 (v_max_i32: negative floats and -0.0 are negative integers)."""
    return np.maximum(x.view(np.int32), 0).view(np.float32)


@pytest.mark.parametrize('bn_first', [False, True])
def test_max_pool_commutes_with_the_fp32_activation_under_the_sign_fold(bn_first):
    """The identity the fp32 pooled F(4x4,3x3) epilogue rests on (conv_wino43.hip, POOL): with g = sign of the channel's BatchNorm
    scale folded into the multiply-add that scales the transform's value and adds the bias (x' = fma(z, g k, g b) = g x exactly),
    ONE max-pool of x' and one activation of the pooled value -- clamp by v_med3_i32 with the bounds (0, INT_MAX) for g = 1 and
    (INT_MIN, 0) for g = -1, then fma with |s| -- give the bits of activating all four values and pooling then."""
    rng = np.random.default_rng(17 + bn_first)
    n = 200000
    z = rng.normal(0, 1, (n, 4)).astype(np.float32)
    z[: n // 10] = np.round(z[: n // 10] * 4) / 4                       # ties inside a window
    z[n // 10: n // 8] = 0.0
    k = rng.choice(np.array([1.0, 0.75, 0.5625, 0.421875, 0.31640625], dtype=np.float32), (n, 4))      # sigma_r sigma_c = a^e
    b = (rng.normal(0, 0.5, (n, 1)) * (rng.random((n, 1)) < 0.9)).astype(np.float32)
    s = (rng.normal(0, 1, (n, 1)) * np.where(rng.random((n, 1)) < 0.02, 0.0, 1.0)).astype(np.float32)    # both signs, some zeros
    t = rng.normal(0, 0.5, (n, 1)).astype(np.float32)
    bb, ss, tt = (np.broadcast_to(v, (n, 4)).copy() for v in (b, s, t))
    # reference order (MultiPoint.py:137-148 then the MaxPool2d of :168-185): activate every value, pool
    x = _fma32(z, k, bb)
    act = _relu_bits(_fma32(x, ss, tt)) if bn_first else _fma32(_relu_bits(x), ss, tt)
    want = act.max(axis=1)
    # kernel order
    g = np.where(s < 0, np.float32(-1), np.float32(1)).astype(np.float32)
    gg = np.broadcast_to(g, (n, 4))
    xp = _fma32(z, k * gg, bb * gg)
    assert np.array_equal(xp, x * gg)                                   # the sign fold is exact
    m = xp.max(axis=1)
    sa, t1, g1 = np.abs(s[:, 0]), t[:, 0], g[:, 0]
    if bn_first:
        got = _relu_bits(_fma32(m, sa, t1))
    else:
        lo = np.where(g1 < 0, np.int32(-2 ** 31), np.int32(0)); hi = np.where(g1 < 0, np.int32(0), np.int32(2 ** 31 - 1))
        c = np.clip(m.view(np.int32), lo, hi).view(np.float32)          # v_med3_i32
        got = _fma32(c, sa, t1)
    assert np.array_equal(got, want)                                    # values ...
    same_bits = got.view(np.int32) == want.view(np.int32)
    assert np.all(same_bits | ((got == 0) & (want == 0)))               # ... and bits, up to the sign of an exact zero

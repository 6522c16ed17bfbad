"""GPU (MI355X, ONE device): what of the multi-GPU path (reference analogue export_keypoints.py:52-53) can be
tested without a second GPU.

  * the rank program itself under torchrun with RCCL: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`
    -- init_process_group('nccl'), barrier, all_reduce(MAX) of the step time and the metric all_gather on device tensors;
  * world size 2 EMULATED sequentially on the one device: shard 0 and shard 1 of 8 pairs (dist.shard_pairs) run as two
    independent single-rank pipelines; their `pair_metric_records`, concatenated the way `gather_pair_metrics` orders them
    (by rank), must equal the rows of ONE rank processing all 8 pairs -- "gathered metrics equal to a single-rank run".
    (The gather itself is covered over gloo, world size 2, in tests/test_dist_gloo.py.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_rank_program_under_torchrun_rccl():
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--pairs-per-gpu', '4', '--no-cpu-baseline']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['scaling'] == 'weak' and d['unit'] == 'image-pairs/s'
    assert d['config']['pairs_per_gpu'] == 4 and d['config']['parallelism'] == 'dp1'
    assert d['value'] > 0 and abs(d['value'] - 4 * 2 / (d['ms_per_step'] * 2e-3)) <= 0.02 * d['value']
    assert d['pair_metrics']['pairs'] == 4                       # the gathered records: one row per pair of the job
    assert d['roofline']['frac'] > 0 and d['roofline']['frac_algorithmic'] > 0
    assert d['cpu_baseline'] is None
    # the job is what the line says: RCCL saw one rank, its own loop time is the job's, every pair's record arrived
    assert d['ranks']['ranks_seen'] == 1 and d['ranks']['gathered_records'] == 4
    assert len(d['ranks']['per_rank_ms_per_step']) == 1 and abs(d['ranks']['per_rank_ms_per_step'][0] - d['ms_per_step']) <= 0.01 * d['ms_per_step']
    assert d['step_ms']['min'] <= d['step_ms']['median'] <= d['step_ms']['max']


def test_value_under_torchrun_equals_the_plain_line():
    """N = 1 under torchrun (RCCL initialised, barrier + all-reduce around the timed loop) must report the same rate as the plain
    N = 1 run within the run-to-run noise: the distributed plumbing costs the timed region nothing."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    common = ['--gpus', '1', '--steps', '12', '--warmup', '4', '--pairs-per-gpu', '16', '--no-cpu-baseline']
    plain = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + common, env=env, capture_output=True, text=True,
                           timeout=600, cwd=ROOT)
    assert plain.returncode == 0, plain.stderr[-2000:]
    tr = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                         '127.0.0.1', '--master-port', '29534', os.path.join(ROOT, 'bench.py')] + common, env=env,
                        capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert tr.returncode == 0, tr.stderr[-2000:]
    a = json.loads([l for l in plain.stdout.splitlines() if l.startswith('{')][-1])
    b = json.loads([l for l in tr.stdout.splitlines() if l.startswith('{')][-1])
    assert 'ranks' not in a and b['ranks']['ranks_seen'] == 1 and b['ranks']['gathered_records'] == b['config']['pairs_per_gpu']
    assert b['ranks']['record_fields'] == ['pair_id', 'n_kp_a', 'n_kp_b', 'n_matches', 't_forward', 't_nms', 't_match', 'desc_err']
    assert b['pair_metrics']['t_forward_ms'] > 0 and b['pair_metrics']['t_nms_ms'] > 0 and b['pair_metrics']['t_match_ms'] > 0
    # (two processes a few seconds apart on a part whose clock drifts by 2-3 %: the per-step MEDIAN of the hipEvent times is the robust
    # comparison; the wall-clock rates of 12 steps get a wider band -- a 6 % band on them failed once in ~10 suite runs of round 6)
    assert abs(a['step_ms']['median'] - b['step_ms']['median']) <= 0.08 * a['step_ms']['median'], (a['step_ms'], b['step_ms'])
    assert abs(a['value'] - b['value']) <= 0.15 * a['value'], (a['value'], b['value'])


def test_emulated_world2_equals_single_rank(oracle):
    import multipoint_amd.models as M
    from multipoint_amd.dist import pair_metric_records, shard_pairs
    from multipoint_amd.pipeline import PairPipeline
    sys.path.insert(0, ROOT)
    import bench
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    net = M.MultiPoint(dict(cfg)); net.load_state_dict(oracle.make_weights(0, cfg)); net.to('cuda'); net.eval()
    pred = dict(bench.PRED_CFG)
    n_pairs, world, H, W = 8, 2, 240, 320
    dev = torch.device('cuda', 0)

    def run(pair_ids):
        pipe = PairPipeline(net, pred, capacity=pred['topk'])
        images = bench.make_batch(pair_ids, dev, H, W)
        res = pipe.run_converged(images)
        return pair_metric_records(res, pair_ids).cpu(), res.to_host()

    single, single_host = run(list(range(n_pairs)))
    gathered, hosts = [], {}
    for rank in range(world):
        ids = shard_pairs(n_pairs, rank, world)
        rec, host = run(ids)
        gathered.append(rec)
        for p, h in zip(ids, host):
            hosts[p] = h
    gathered = torch.cat(gathered, dim=0)                        # gather_pair_metrics orders by rank
    assert sorted(gathered[:, 0].tolist()) == list(range(n_pairs))
    assert torch.equal(gathered[torch.argsort(gathered[:, 0])][:, :4], single[:, :4]) and single.shape[1] == 8
    assert int(single[:, 3].min()) > 0                           # real matches, not empty rows
    # a pair's results do not depend on which batch (shard) it was processed in: same keypoints, descriptors, matches
    for p in range(n_pairs):
        a, b = single_host[p], hosts[p]
        for k in ('kp_optical', 'kp_thermal', 'match_query', 'match_train'):
            assert np.array_equal(a[k], b[k]), (p, k)
        assert np.array_equal(a['desc_optical'], b['desc_optical']) and np.array_equal(a['match_dist'], b['match_dist'])


def test_numa_binding_resolves_the_gpu_on_this_box():
    """dist.bind_rank_to_numa_node on real hardware: the KFD topology lists this box's GPU(s) with a PCI id that exists under
    /sys/bus/pci/devices, and binding rank 0 either takes effect (a non-empty subset of the CPUs the process may run on) or reports
    None because the container does not expose the topology -- never a wrong or empty set.  The affinity is restored afterwards."""
    from multipoint_amd import dist as D
    before = os.sched_getaffinity(0)
    try:
        nodes = D._kfd_gpu_nodes()
        if os.path.isdir('/sys/class/kfd/kfd/topology/nodes'):
            assert len(nodes) >= 1, 'KFD topology is exposed but lists no GPU node'
            for _, bdf in nodes:
                assert len(bdf.split(':')) == 3 and '.' in bdf
        got = D.bind_rank_to_numa_node(0)
        print('\n[numa] kfd gpu nodes %s -> rank 0 bound to %s' % (nodes, None if got is None else '%d CPUs %d-%d' % (len(got), got[0], got[-1])))
        if got is not None:
            assert len(got) > 0 and set(got) <= before and set(got) == os.sched_getaffinity(0)
            assert nodes and os.path.exists(os.path.join('/sys/bus/pci/devices', nodes[0][1], 'local_cpulist'))
        else:
            assert os.sched_getaffinity(0) == before
    finally:
        os.sched_setaffinity(0, before)

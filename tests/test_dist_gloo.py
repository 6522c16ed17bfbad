"""CPU, world_size 2 over gloo: the multi-GPU path (pair sharding + the metric gather, which is the
only collective of the hot path)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_pairs, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from multipoint_amd.dist import shard_pairs, gather_pair_metrics
    mine = shard_pairs(n_pairs, rank, world)
    # a per-pair record that only depends on the pair id (stands in for kp/match counts)
    rec = torch.tensor([[p, 3 * p + 1, 7 * p % 5, p * p] for p in mine], dtype=torch.int32).reshape(-1, 4)
    allrec = gather_pair_metrics(rec)
    torch.save(allrec, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    n_pairs, world = 7, 2          # ragged: rank 0 owns 4 pairs, rank 1 owns 3
    mp.spawn(_worker, args=(world, _free_port(), n_pairs, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(tmp_path / 'rank0.pt'); b = torch.load(tmp_path / 'rank1.pt')
    assert torch.equal(a, b)
    assert a.shape == (n_pairs, 4)
    ids = sorted(a[:, 0].tolist())
    assert ids == list(range(n_pairs))                      # every pair exactly once
    for row in a.tolist():
        p = row[0]
        assert row == [p, 3 * p + 1, 7 * p % 5, p * p]      # identical to a single-rank run


def _results_for(pair_ids, K=50):
    """A PairResults with the tensor shapes/dtypes PairPipeline produces (pipeline.py: kp_yx [2P,K,2] i32, kp_count [2P] i32
    -- may exceed the capacity K --, desc [2P,K,D], match_idx/dist [P,K], match_count [P]); contents depend on the pair id
    only, so a sharded run must reproduce the single-rank rows."""
    from multipoint_amd.pipeline import PairResults
    P = len(pair_ids)
    kp_count = torch.tensor([[(11 * p) % 70, (7 * p + 3) % 70] for p in pair_ids], dtype=torch.int32).reshape(-1)
    match_count = torch.tensor([min((11 * p) % 70, (7 * p + 3) % 70, K) // 2 for p in pair_ids], dtype=torch.int32)
    return PairResults(torch.zeros((2 * P, K, 2), dtype=torch.int32), torch.zeros((2 * P, K)), kp_count,
                       torch.zeros((2 * P, K, 8)), torch.full((P, K), -1, dtype=torch.int32), torch.zeros((P, K)),
                       match_count, 64, 64)


def _stage_ms(rank):
    return {'forward': 9.0 + rank, 'nms': 0.25 + rank, 'match': 0.125 + rank}


def _worker_results(rank, world, port, n_pairs, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from multipoint_amd.dist import shard_pairs, gather_pair_metrics, pair_metric_records
    mine = shard_pairs(n_pairs, rank, world)
    allrec = gather_pair_metrics(pair_metric_records(_results_for(mine), mine, stage_ms=_stage_ms(rank),
                                                     desc_err=[1e-6 * p if p % 2 else None for p in mine]))      # exactly bench.py's gather
    torch.save(allrec, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_of_pair_results_world2(tmp_path):
    """The record bench.py gathers, built from PairResults-shaped tensors by the product's own pair_metric_records()."""
    from multipoint_amd.dist import pair_metric_records
    n_pairs, world = 9, 2
    mp.spawn(_worker_results, args=(world, _free_port(), n_pairs, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(tmp_path / 'rank0.pt'); b = torch.load(tmp_path / 'rank1.pt')
    from multipoint_amd.dist import RECORD_FIELDS
    assert RECORD_FIELDS == ('pair_id', 'n_kp_a', 'n_kp_b', 'n_matches', 't_forward', 't_nms', 't_match', 'desc_err')
    assert torch.equal(a.nan_to_num(-1.0), b.nan_to_num(-1.0)) and a.dtype == torch.float64 and a.shape == (n_pairs, 8)
    single = pair_metric_records(_results_for(list(range(n_pairs))), list(range(n_pairs)))
    order = torch.argsort(a[:, 0])
    assert torch.equal(a[order][:, :4], single[:, :4])       # ids and counts: same rows as one rank processing every pair
    assert torch.isnan(single[:, 4:]).all()                  # nothing timed, nothing checked: NaN, not zero
    assert int(a[:, 1].max()) <= 50 and int(a[:, 2].max()) <= 50      # counts clamped to the list capacity
    for row in a.tolist():
        p = int(row[0]); r = p % world                      # the stage times are those of the rank (batch) that owned the pair
        assert row[4:7] == [9.0 + r, 0.25 + r, 0.125 + r]
        assert (row[7] == 1e-6 * p) if p % 2 else (row[7] != row[7])


def test_shard_pairs_partition():
    from multipoint_amd.dist import shard_pairs, gather_pair_metrics
    for world in (1, 2, 4, 8):
        parts = [shard_pairs(37, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(37))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    x = torch.arange(6).reshape(3, 2)
    assert torch.equal(gather_pair_metrics(x), x)            # no process group: identity


def _worker_rank_program(rank, world, port, n_pairs, out_dir):
    """The collective part of bench.py's rank program (max step time, per-rank times, the count of ranks the communicator has,
    the metric gather) over gloo."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from multipoint_amd.dist import (bind_rank_to_numa_node, gather_pair_metrics, gather_scalar, pair_metric_records,
                                     ranks_seen, shard_pairs)
    bound = bind_rank_to_numa_node(rank)                    # no GPU topology in the CPU container: silently None
    mine = shard_pairs(n_pairs, rank, world)
    dt = 0.010 * (rank + 1)                                 # this rank's loop time
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rec = gather_pair_metrics(pair_metric_records(_results_for(mine), mine)) if mine else \
        gather_pair_metrics(torch.zeros((0, 8), dtype=torch.float64))
    torch.save({'max': float(t.item()), 'per_rank': gather_scalar(dt, 'cpu'), 'seen': ranks_seen('cpu'), 'rec': rec,
                'bound': bound}, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_program_world4_ragged(tmp_path):
    """World size 4 with ragged shards (6 pairs: ranks 0, 1 own two, ranks 2, 3 one): every rank ends up with the same
    gathered rows = the single-rank rows, the max over the ranks' times, all four per-rank times and a rank count of 4."""
    from multipoint_amd.dist import pair_metric_records
    n_pairs, world = 6, 4
    mp.spawn(_worker_rank_program, args=(world, _free_port(), n_pairs, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / ('rank%d.pt' % r), weights_only=False) for r in range(world)]
    single = pair_metric_records(_results_for(list(range(n_pairs))), list(range(n_pairs)))
    for o in outs:
        assert o['seen'] == world and abs(o['max'] - 0.040) < 1e-12
        assert [round(v, 6) for v in o['per_rank']] == [0.01, 0.02, 0.03, 0.04]
        assert torch.equal(o['rec'].nan_to_num(-1.0), outs[0]['rec'].nan_to_num(-1.0)) and o['rec'].shape == (n_pairs, 8)
        assert o['rec'].dtype == torch.float64
        assert torch.equal(o['rec'][torch.argsort(o['rec'][:, 0])].nan_to_num(-1.0), single.nan_to_num(-1.0))
        assert o['bound'] is None or len(o['bound']) > 0


def test_rank_helpers_without_a_process_group():
    from multipoint_amd.dist import gather_scalar, hw_queues_note, ranks_seen
    assert ranks_seen('cpu') == 1 and gather_scalar(1.5, 'cpu') == [1.5]
    assert hw_queues_note()['GPU_MAX_HW_QUEUES'] is not None      # importing multipoint_amd.dist sets the deployment default


def test_numa_binding_resolves_the_visible_device(monkeypatch):
    """bind_rank_to_numa_node maps the local rank through the *_VISIBLE_DEVICES filters (integer lists) and refuses what it cannot
    map -- binding to another GPU's node would be worse than not binding (round-4 advisor)."""
    from multipoint_amd import dist as D
    for k in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(k, raising=False)
    assert D._visible_device_index(3) == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '4,5,6,7')
    assert D._visible_device_index(1) == 5 and D._visible_device_index(4) is None
    monkeypatch.setenv('CUDA_VISIBLE_DEVICES', '4,5,6,7')
    assert D._visible_device_index(0) == 4                       # the HIP / CUDA aliases agree: fine
    monkeypatch.setenv('CUDA_VISIBLE_DEVICES', '0,1,2,3')
    assert D._visible_device_index(0) is None                    # the aliases disagree: not ours to guess
    monkeypatch.delenv('CUDA_VISIBLE_DEVICES')
    # ROCR_VISIBLE_DEVICES is another layer: HIP indexes into the ROCr-filtered list (round-5 advisor)
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '1,0')
    assert D._visible_device_index(0) is None                    # HIP index 4 of a two-entry ROCr list: unmappable
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '2,3'); monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1')
    assert D._visible_device_index(0) == 2 and D._visible_device_index(1) == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '2,3')             # identical strings are still two layers
    assert D._visible_device_index(0) is None
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '4,5,6,7'); monkeypatch.setenv('HIP_VISIBLE_DEVICES', '3,1')
    assert D._visible_device_index(0) == 7 and D._visible_device_index(1) == 5 and D._visible_device_index(2) is None
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    assert D._visible_device_index(1) == 5                       # ROCr filter alone
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', 'GPU-deadbeef')
    assert D._visible_device_index(0) is None                    # UUIDs cannot be mapped without a GPU call
    assert D.bind_rank_to_numa_node(0) is None                   # and then nothing is bound


def test_c4_split_world8_gloo(tmp_path):
    """BASELINE configs[3]: 256 pairs over 8 ranks, 32 each, pair p on rank p mod 8 -- the exact split the 8-GPU line runs -- and a
    ragged total (250 pairs: ranks 0-1 own 32, the rest 31); every record gathered once, identical on every rank."""
    from multipoint_amd.dist import shard_pairs
    for n in (256, 250):
        owned = [shard_pairs(n, r, 8) for r in range(8)]
        assert sorted(p for o in owned for p in o) == list(range(n))
        assert all(all(p % 8 == r for p in o) for r, o in enumerate(owned))
        assert [len(o) for o in owned] == [n // 8 + (1 if r < n % 8 else 0) for r in range(8)]
    n_pairs, world = 250, 8
    mp.spawn(_worker, args=(world, _free_port(), n_pairs, str(tmp_path)), nprocs=world, join=True)
    recs = [torch.load(tmp_path / ('rank%d.pt' % r)) for r in range(world)]
    assert all(torch.equal(recs[0], r) for r in recs[1:]) and recs[0].shape == (n_pairs, 4)
    assert sorted(recs[0][:, 0].tolist()) == list(range(n_pairs))


def test_bench_rank_program_decisions(monkeypatch):
    """bench.py's rank program, the parts that decide WHAT the N > 1 job is (pure functions, no GPU): the device of a rank under
    torchrun (all GPUs visible: LOCAL_RANK) and under a launcher that pre-sets HIP_VISIBLE_DEVICES per rank (one visible: 0);
    `config.workload` names BASELINE configs[3] for --gpus 8 --pairs-per-gpu 32; the CPU baseline / parity / secondary legs run
    on N = 1 only, so that the timed N > 1 job is the ranks and nothing else."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    assert bench.rank_env({}) == (0, 1, 0)
    assert bench.rank_env({'RANK': '5', 'WORLD_SIZE': '8', 'LOCAL_RANK': '5'}) == (5, 8, 5)
    assert [bench.local_device_index(r, 8) for r in range(8)] == list(range(8))       # torchrun: every GPU visible to every rank
    assert [bench.local_device_index(r, 1) for r in range(8)] == [0] * 8              # HIP_VISIBLE_DEVICES=<one GPU> per rank
    import pytest
    with pytest.raises(RuntimeError):
        bench.local_device_index(5, 4)
    with pytest.raises(RuntimeError):
        bench.local_device_index(0, 0)
    w8 = bench.workload_name(32, 8, 480, 640, 1000)
    assert w8.startswith('BASELINE configs[3]: 256 pairs 480x640 sharded over 8 GPUs (32 per GPU')
    for world in (1, 2, 4):
        assert bench.workload_name(32, world, 480, 640, 1000).startswith('BASELINE configs[2]: 32 pairs (=64 images) 480x640 per GPU')
    assert bench.workload_name(16, 8, 480, 640, 1000).startswith('BASELINE configs[2]')          # not C4's split
    assert bench.workload_name(8, 8, 1024, 1280, 2000, c5=True).startswith('BASELINE configs[4] (64 pairs sharded over 8 GPUs)')
    assert bench.workload_name(8, 1, 1024, 1280, 2000, c5=True).startswith('BASELINE configs[4] (per-GPU share)')
    assert bench.runs_extra_legs(1, False) == (True, True)
    assert bench.runs_extra_legs(1, True) == (True, False)                                       # torchrun at world 1: no secondary
    for world in (2, 4, 8):
        assert bench.runs_extra_legs(world, True) == (False, False)
    assert bench.runs_extra_legs(1, False, no_cpu_baseline=True) == (False, False)
    assert bench.runs_extra_legs(1, False, c5=True) == (True, False)
    # the NUMA binding sees the same device: a rank with one visible GPU resolves ITS filter entry, not entry LOCAL_RANK
    from multipoint_amd import dist as D
    for k in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '5')
    assert D._visible_device_index(bench.local_device_index(5, 1)) == 5

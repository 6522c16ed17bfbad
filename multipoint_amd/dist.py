"""Multi-GPU plumbing: image pairs are independent end to end, so ranks shard
pairs round-robin and never exchange data on the compute path.  The only collective is a gather of
small per-pair metric records -- RCCL (`backend="nccl"` on ROCm) over xGMI on the GPU box, gloo in
the CPU tests.

Deployment setting of the multi-GPU path, applied when THIS module is imported (not by `import multipoint_amd`: a host
application's ROCm runtime is not ours to configure) and only if the variable is unset: `GPU_MAX_HW_QUEUES=8`.  An RCCL
communicator creates streams of its own, and with the runtime's default of 4 hardware queues the pipeline's post-processing
stream then shares a queue with the convolution stream -- the two serialise and a step gets 4 % longer.
The runtime reads the variable when it initialises (the first GPU call), so import this module -- or set the variable --
before that; `hw_queues_note()` says whether it took effect."""
import os

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def hw_queues_note():
    """The GPU_MAX_HW_QUEUES value this process runs with, and whether the ROCm runtime was already initialised when this
    module set its default (in which case the setting has no effect until the next process)."""
    return {'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES'),
            'runtime_initialised_at_query': bool(torch.cuda.is_initialized())}


def bind_rank_to_numa_node(local_rank):
    """Pin this rank's threads to the CPUs of its GPU's NUMA node (eight Python launch loops otherwise wander over the
    sockets of a 256-CPU host).  Reads /sys/class/drm/card*/device/{numa_node,local_cpulist} -- no GPU call, so it can and
    should run before the first one.  Returns the CPU set it bound to, or None when the topology is not exposed (then
    nothing changes): binding is an optimisation, never a requirement."""
    import glob
    try:
        cards = sorted(glob.glob('/sys/class/drm/card[0-9]*/device/local_cpulist'),
                       key=lambda q: int(''.join(ch for ch in q.split('/')[4] if ch.isdigit()) or 0))
        # render nodes without a compute device (ASPEED etc.) have no 'vendor' 0x1002: keep AMD GPUs only
        gpus = []
        for q in cards:
            d = os.path.dirname(q)
            try:
                if open(os.path.join(d, 'vendor')).read().strip() == '0x1002':
                    gpus.append(d)
            except OSError:
                pass
        if local_rank >= len(gpus):
            return None
        cpus = set()
        for part in open(os.path.join(gpus[local_rank], 'local_cpulist')).read().strip().split(','):
            if not part:
                continue
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return sorted(cpus)
    except (OSError, ValueError, AttributeError):
        return None


def ranks_seen(device):
    """An all-reduce of ones over the default group: the number of ranks the communicator really has (proof, in a bench line,
    that RCCL saw N members)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    t = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(t)
    return int(t.item())


def gather_scalar(value, device):
    """One float per rank, in rank order, on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def shard_pairs(num_pairs, rank, world_size):
    """Pair ids owned by `rank`: p with p % world_size == rank (evaluation order is irrelevant:
    the reference's DataLoader is shuffle=False and every pair is independent)."""
    return list(range(rank, num_pairs, world_size))


def pair_metric_records(res, pair_ids):
    """Per-pair metric rows [pair_id, n_kp_optical, n_kp_thermal, n_matches] (int32, on the device of `res`) of one
    PairResults batch whose pair i has global id pair_ids[i]: keypoint counts are clamped to the list capacity like
    PairResults.to_host().  This is the record the ranks gather."""
    K = res.kp_yx.shape[1]
    dev = res.kp_count.device
    ids = torch.as_tensor(list(pair_ids), dtype=torch.int32, device=dev)
    if ids.numel() != res.match_count.shape[0]:
        raise ValueError('pair_ids has %d entries for %d pairs' % (ids.numel(), res.match_count.shape[0]))
    return torch.stack([ids, res.kp_count[0::2].clamp(max=K).to(torch.int32), res.kp_count[1::2].clamp(max=K).to(torch.int32),
                        res.match_count.to(torch.int32)], dim=1)


def gather_pair_metrics(records):
    """records: [n_local, F] int32/float tensor of per-pair metric rows (same n_local on every rank
    for weak scaling; ragged counts are padded).  Returns the [sum n, F] concatenation on every rank,
    ordered by rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return records
    world = dist.get_world_size()
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    nmax = int(max(c.item() for c in counts))
    pad = torch.zeros((nmax,) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
    pad[:records.shape[0]] = records
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:int(c.item())] for b, c in zip(bufs, counts)], dim=0)

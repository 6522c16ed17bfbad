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


def _visible_device_index(local_rank):
    """The PHYSICAL index (KFD / ROCr enumeration order) of the GPU HIP device `local_rank` is, honouring the device filters when
    they are plain integer lists.  HIP_VISIBLE_DEVICES and CUDA_VISIBLE_DEVICES are aliases (HIP reads either): they must agree.
    ROCR_VISIBLE_DEVICES is a different LAYER -- it filters what the ROCr runtime exposes, and HIP then indexes into that
    filtered list -- so the two COMPOSE: physical = rocr_ids[hip_ids[local_rank]] (ROCR='2,3' with HIP='0,1' is GPUs 2 and 3,
    also when the two strings happen to be equal).  None when a filter cannot be mapped by hand (UUIDs, disagreeing aliases,
    an index out of range): the caller then does not bind at all -- a wrong node is worse than none."""
    def ints(key):
        v = os.environ.get(key, '')
        if v == '':
            return None
        return [int(x) for x in v.split(',') if x.strip() != '']          # ValueError: not an integer list
    try:
        rocr, hip, cuda = ints('ROCR_VISIBLE_DEVICES'), ints('HIP_VISIBLE_DEVICES'), ints('CUDA_VISIBLE_DEVICES')
    except ValueError:
        return None
    if hip is not None and cuda is not None and hip != cuda:
        return None
    hip = hip if hip is not None else cuda
    idx = local_rank
    if hip is not None:
        if idx >= len(hip):
            return None
        idx = hip[idx]
    if rocr is not None:
        if idx < 0 or idx >= len(rocr):
            return None
        idx = rocr[idx]
    return idx if idx >= 0 else None


def _kfd_gpu_nodes():
    """GPUs in KFD enumeration order (the order ROCr, and with it HIP, numbers them): [(node, 'dddd:bb:dd.f')] from
    /sys/class/kfd/kfd/topology/nodes/*/properties (CPU nodes have simd_count 0).  No GPU context needed."""
    import glob
    out = []
    for d in sorted(glob.glob('/sys/class/kfd/kfd/topology/nodes/[0-9]*'), key=lambda q: int(q.rsplit('/', 1)[1])):
        props = {}
        try:
            for line in open(os.path.join(d, 'properties')):
                k, _, v = line.strip().partition(' ')
                props[k] = v
        except OSError:
            continue
        if int(props.get('simd_count', '0') or 0) == 0:
            continue
        loc = int(props.get('location_id', '0') or 0)
        dom = int(props.get('domain', '0') or 0)
        out.append((int(d.rsplit('/', 1)[1]), '%04x:%02x:%02x.%x' % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)))
    return out


def bind_rank_to_numa_node(local_rank):
    """Pin this rank to the CPUs of its GPU's NUMA node (eight Python launch loops otherwise wander over the sockets of a
    256-CPU host).  The GPU is resolved through its PCI bus id -- KFD topology order = HIP device order, filtered by the
    *_VISIBLE_DEVICES variables when they are integer lists -- and /sys/bus/pci/devices/<bdf>/local_cpulist: no GPU call, so it
    can and should run before the first one.  Returns the CPU set it bound to, or None when the topology is not exposed or a
    device filter cannot be mapped (then nothing changes): binding is an optimisation, never a requirement.
    `os.sched_setaffinity(0, ...)` moves the CALLING thread and the threads it creates afterwards; threads that already exist
    (an imported library's pools) keep their affinity -- call this first thing in the rank program."""
    try:
        phys = _visible_device_index(local_rank)
        gpus = _kfd_gpu_nodes()
        if phys is None or phys >= len(gpus):
            return None
        path = os.path.join('/sys/bus/pci/devices', gpus[phys][1], 'local_cpulist')
        cpus = set()
        for part in open(path).read().strip().split(','):
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


# the per-pair record the ranks gather, one float64 row per pair
RECORD_FIELDS = ('pair_id', 'n_kp_a', 'n_kp_b', 'n_matches', 't_forward', 't_nms', 't_match', 'desc_err')


def pair_metric_records(res, pair_ids, stage_ms=None, desc_err=None):
    """Per-pair metric rows `RECORD_FIELDS` (float64, on the device of `res`; ids and counts are exact in a double) of one
    PairResults batch whose pair i has global id pair_ids[i].  n_kp_a / n_kp_b: keypoints of the optical / thermal image,
    clamped to the list capacity like PairResults.to_host(); t_forward / t_nms / t_match: milliseconds the BATCH this pair was
    in spent in the forward, in box-NMS + top-k + keypoint lists, and in descriptor sampling + matching (`stage_ms`, e.g.
    PairResults.stage_ms() of a run with `timings=True`; pairs of one batch share them; NaN: not timed); desc_err: this pair's
    largest descriptor deviation from the CPU reference path when a parity leg checked it (NaN: not checked)."""
    K = res.kp_yx.shape[1]
    dev = res.kp_count.device
    n = res.match_count.shape[0]
    ids = torch.as_tensor(list(pair_ids), dtype=torch.float64, device=dev)
    if ids.numel() != n:
        raise ValueError('pair_ids has %d entries for %d pairs' % (ids.numel(), n))
    nan = float('nan')
    st = [nan, nan, nan] if stage_ms is None else [float(stage_ms[k]) for k in ('forward', 'nms', 'match')]
    if desc_err is None:
        de = torch.full((n,), nan, dtype=torch.float64, device=dev)
    else:
        de = torch.as_tensor([nan if v is None else float(v) for v in desc_err], dtype=torch.float64, device=dev)
        if de.numel() != n:
            raise ValueError('desc_err has %d entries for %d pairs' % (de.numel(), n))
    cols = [ids, res.kp_count[0::2].clamp(max=K).to(torch.float64), res.kp_count[1::2].clamp(max=K).to(torch.float64),
            res.match_count.to(torch.float64)]
    cols += [torch.full((n,), v, dtype=torch.float64, device=dev) for v in st] + [de]
    return torch.stack(cols, dim=1)


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

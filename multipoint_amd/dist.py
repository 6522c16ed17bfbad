"""Multi-GPU plumbing: image pairs are independent end to end, so ranks shard
pairs round-robin and never exchange data on the compute path.  The only collective is a gather of
small per-pair metric records -- RCCL (`backend="nccl"` on ROCm) over xGMI on the GPU box, gloo in
the CPU tests."""
import torch
import torch.distributed as dist


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

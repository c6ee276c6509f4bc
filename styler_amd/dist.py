"""Data-parallel plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).

The reference's only parallelism is nn.DataParallel (train.py:33), de-facto single-GPU.  Here
utterances are sharded across ranks; the forward has no data-path collective (every sequence is
independent; PostNet BatchNorm statistics stay per-rank exactly as per-replica in DataParallel)."""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world, lengths=None):
    """Utterance indices of `rank`: sorted by length (descending) and dealt in snake order
    (0..w-1, w-1..0, ...) so that padded work is balanced across ranks (the reference's collate already
    sorts by text length, dataset.py:189-198)."""
    order = list(range(n_items))
    if lengths is not None:
        order.sort(key=lambda i: (-int(lengths[i]), i))
    out = []
    for pos, idx in enumerate(order):
        rnd, k = divmod(pos, world)
        if (k if rnd % 2 == 0 else world - 1 - k) == rank:
            out.append(idx)
    return out


def aggregate_throughput(elapsed_s, units, device="cpu"):
    """MAX of the per-rank elapsed time and SUM of the per-rank work units over the default group
    (bench.py's whole-job rate = sum(units) / max(elapsed))."""
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


BUCKET_BYTES = 32 << 20


def allreduce_sum_(flat_grads, bucket_bytes=BUCKET_BYTES):
    """Bucketed in-place SUM all-reduce over a FLAT gradient buffer (29.48 M fp32 = 117.9 MB -> 4 buckets of 32 MB),
    launched back to back on the collective stream.  Returns the async work handles so the caller can overlap the range
    with other work; call `.wait()` on each before the optimizer step.  The division by the world size is NOT a pass
    over the buffer: `TrainState.step` folds 1 / world into the clip + Adam kernel (`grad_scale`)."""
    if world_size() == 1:
        return []
    n = max(1, bucket_bytes // flat_grads.element_size())
    return [dist.all_reduce(flat_grads[start:start + n], op=dist.ReduceOp.SUM, async_op=True)
            for start in range(0, flat_grads.numel(), n)]


def allreduce_mean_(flat_grads, bucket_bytes=BUCKET_BYTES):
    """`allreduce_sum_` followed (after the waits, by the caller's stream order) by one in-place division: for callers
    that want the mean in the buffer itself (logging scalars; the train step uses the sum + grad_scale instead)."""
    works = allreduce_sum_(flat_grads, bucket_bytes)
    if works:
        for w in works:
            w.wait()
        flat_grads.div_(world_size())
    return []

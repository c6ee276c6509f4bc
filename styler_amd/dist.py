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
# STYLER_FORCE_ALLREDUCE=1: issue the gradient all-reduce even on ONE rank (an identity, but every call of the N > 1 path -- RCCL
# communicator, bucketed async collectives on the flat buffer, the launch point inside backward, the waits in step() -- runs on the
# real backend): how a 1-GPU box exercises the "nccl" transport (tests/test_15_dist_gpu.py::test_one_rank_rccl_step)
FORCE_COLLECTIVES = __import__("os").environ.get("STYLER_FORCE_ALLREDUCE", "0") == "1"


def allreduce_sum_(flat_grads, bucket_bytes=BUCKET_BYTES):
    """Bucketed in-place SUM all-reduce over a FLAT gradient buffer (29.48 M fp32 = 117.9 MB -> 4 buckets of 32 MB),
    launched back to back on the collective stream.  Returns the async work handles so the caller can overlap the range
    with other work; call `.wait()` on each before the optimizer step.  The division by the world size is NOT a pass
    over the buffer: `TrainState.step` folds 1 / world into the clip + Adam kernel (`grad_scale`)."""
    if world_size() == 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized()):
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


class Bf16Reducer:
    """Optional bf16 transport for the gradient all-reduce (STYLER_ALLREDUCE_BF16=1; off by default: it changes the update
    arithmetic the parity tests pin).  The fp32 range is cast into a persistent bf16 shadow (one pass: 4 B read + 2 B
    written per element), the shadow is all-reduced in `BUCKET_BYTES` buckets -- half the bytes on every xGMI link -- and
    cast back into the fp32 buffer after the waits (exact).  The sum itself is formed in bf16 by the collective: with N
    ranks each element carries up to N - 1 bf16 roundings (2^-9 relative each), the same order as the bf16 operand rounding
    of the throughput mode's GEMMs; tests/test_01_host_cpu.py pins the two-rank delta."""

    def __init__(self, flat_grads):
        self.flat = flat_grads
        self.shadow = torch.empty(flat_grads.numel(), device=flat_grads.device, dtype=torch.bfloat16)
        self.pending = []

    def _cast_down(self, lo, hi):
        if self.flat.is_cuda:
            from . import ops
            ops._chk(ops.lib.styler_cast_bf16(self.flat[lo:hi].data_ptr(), self.shadow[lo:hi].data_ptr(), hi - lo, ops._stream()),
                     "styler_cast_bf16")
        else:
            self.shadow[lo:hi].copy_(self.flat[lo:hi])

    def _cast_up(self, lo, hi):
        if self.flat.is_cuda:
            from . import ops
            ops._chk(ops.lib.styler_cast_from_bf16(self.shadow[lo:hi].data_ptr(), self.flat[lo:hi].data_ptr(), hi - lo,
                                                   ops._stream()), "styler_cast_from_bf16")
        else:
            self.flat[lo:hi].copy_(self.shadow[lo:hi])

    def start(self, lo, hi):
        """Begin the all-reduce of flat[lo:hi] (lo, hi multiples of 4 elements); returns the async work handles."""
        if world_size() == 1:
            return []
        self._cast_down(lo, hi)
        self.pending.append((lo, hi))
        return allreduce_sum_(self.shadow[lo:hi])

    def finish(self):
        """After the handles have been waited for: the reduced values back into the fp32 buffer."""
        for lo, hi in self.pending:
            self._cast_up(lo, hi)
        self.pending = []


def allreduce_preflight(device, nbytes=117_930_804, reps=5):
    """What one gradient all-reduce costs on this job's links, measured before the timed steps (bench.py prints it in
    `config`): ranks seen, milliseconds and bus bandwidth (algbw * 2 (N - 1) / N) of an fp32 SUM all-reduce of `nbytes` in
    BUCKET_BYTES buckets, and of the bf16 transport of the same gradient (half the bytes)."""
    import time
    n = world_size()
    out = {"ranks": n}
    if n == 1:
        return out
    def agree(err, ms=0.0):
        """(anyone failed?, MAX ms) -- reached by EVERY rank, the failing one from its except branch (a rank
        that failed alone used to skip the collective the others then blocked in forever)."""
        t = torch.tensor([0.0 if err is None else 1.0, ms], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item()) > 0, float(t[1].item())

    for name, dtype, elems in (("fp32", torch.float32, nbytes // 4), ("bf16", torch.bfloat16, nbytes // 4)):
        err, buf = None, None
        try:                                         # the allocation alone: a rank that fails HERE has entered no collective yet
            buf = torch.zeros(elems, device=device, dtype=dtype)
        except RuntimeError as e:
            err = str(e)[:120]
        # every rank agrees on the allocation BEFORE any data collective, so all ranks issue the same collective sequence (round-5
        # advisor: a rank that died allocating used to answer the others' bucketed all-reduce with agree()'s 2-element one)
        failed, _ = agree(err)
        if failed:
            out[name] = {"error": err if err is not None else "another rank failed"}
            continue
        try:                                         # one warm-up all-reduce (dtype support of the backend): entered by every rank
            for w in allreduce_sum_(buf):
                w.wait()
            if buf.is_cuda:
                torch.cuda.synchronize()
        except RuntimeError as e:                    # (a backend without this dtype fails on every rank alike: say so instead of dying)
            err = str(e)[:120]
        failed, _ = agree(err)                       # doubles as the barrier in front of the timed repetitions
        if failed:
            out[name] = {"error": err if err is not None else "another rank failed"}
            continue
        ms = 0.0
        try:
            t0 = time.perf_counter()
            for _ in range(reps):
                for w in allreduce_sum_(buf):
                    w.wait()
            if buf.is_cuda:
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
        except RuntimeError as e:
            err = str(e)[:120]
        failed, ms = agree(err, ms)
        if failed:
            out[name] = {"error": err if err is not None else "another rank failed"}
            continue
        b = elems * buf.element_size()
        out[name] = {"bytes": b, "ms": round(ms, 3), "busbw_GBs": round(b / (ms * 1e-3) * 2 * (n - 1) / n / 1e9, 1)}
    return out

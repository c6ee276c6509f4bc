"""Tensor-level wrappers over the C ABI (include/styler_hip.h).

PyTorch supplies device memory (caching allocator), the current HIP stream and nothing else: every
function here hands raw device pointers + sizes to libstyler_hip.so.  Activations are fp32
channels-last [B, L, C]; a channel slice `x[..., a:b]` of a wider buffer is a legal argument (the row
stride travels as `ld`), which is how torch.cat / torch.split copies of the reference disappear."""
import os

import torch

from ._lib import lib

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
ACT_LEAKY = 4                      # leaky_relu(0.1), the vocoder's LRELU_SLOPE
PREC_F32, PREC_BF16 = 0, 1
PREC_BF16X3 = 2                    # host-level arithmetic: fp32-class products as three bf16 MFMA products (split3 / lo_part)


class StylerHipError(RuntimeError):
    pass


def _chk(rc, name):
    if rc != 0:
        # a failed entry point may have returned before consuming this thread's one-shot registrations (the bf16x3 split output,
        # the split-K workspace): drop them, so the NEXT call cannot write into a buffer the unwinding frees (round-5 advisor)
        lib.styler_set_x3_out(None, 0)
        lib.styler_gemm_set_workspace(None, 0)
        lib.styler_gemm_set_counters(None, 0)
        raise StylerHipError(f"{name} failed with code {rc}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    """Row stride (elements) of a [B, L, C] or [rows, C] fp32 view whose channels are contiguous."""
    assert t.stride(-1) == 1 or t.shape[-1] == 1, "channel dim must be contiguous"
    ld = t.stride(-2)
    if t.dim() == 3 and t.shape[0] > 1:
        assert t.stride(0) == t.shape[1] * ld, "batch stride must equal L * row stride"
    return ld


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda, (t.dtype, t.device)
    return t


# the loss-only side stream of the current step (modules.StyleModeling._loss_only_stream, rt.pred_stream), or None
loss_side_stream = None


class WgradArena:
    """Workspace arena for one backward pass: every styler_wgrad call takes a slice and defers its split-K
    reduction; `flush()` folds all partials into the gradients with ONE launch.  The first pass only measures
    (calls reduce immediately); the arena is sized from it."""

    _DESC = None
    GROUP_TABLE_BYTES = 1 << 16

    def __init__(self):
        self.buf = None
        self.used = 0
        self.total = 0                               # floats requested by the current pass (taken or not)
        self.descs = []
        self._cache = {}                             # descriptor tuple -> (device table, total blocks)
        # bf16 mode: weight gradients are not launched one by one -- they are collected and run as ONE grouped launch per
        # kernel variant at flush (the group fills the chip together, so a member needs 1-3 split-K partials, not 8-37)
        self.group = []                              # WgradGroupDesc of the pending members
        self.group_keep = []                         # operand tensors, alive until the grouped launches have been enqueued
        self.flush_no = 0                            # flushes so far in this pass (the decoder-side one comes first)
        self.hist = {}                               # (flush number, variant) -> tiles of that group in the previous pass
        self._gcache = {}                            # descriptor bytes -> (pinned host table, device table)
        self._pinned_pool = []
        # weight gradients on a side stream (rt.wgrad_stream): the dX chain of backward never reads a weight gradient, so
        # the two chains only meet at flush(); operands stay referenced until then
        self.side = None
        self.side_used = False
        self.side_keep = []
        # an arena whose buffer / tables are addressed by a captured hipGraph (training.GraphedTrainStep owns one per
        # graph): descriptor tables are never evicted, and once `frozen` the buffer must not be re-sized
        self.owned_by_graph = False
        self.frozen = False

    def reset_plan(self):
        """Forget everything sized / cached from the engine's split plan (ops.wgrad_tune flipped a knob): the next pass measures
        again.  Refused for an arena a captured hipGraph addresses -- that step has to be captured anew."""
        if self.frozen:
            raise StylerHipError("WgradArena: a captured hipGraph addresses this arena; re-capture the step after wgrad_tune")
        self.buf, self.used, self.total = None, 0, 0
        self._cache.clear(); self._gcache.clear(); self.hist.clear()
        self.descs, self.group, self.group_keep = [], [], []

    def begin(self, device=None):
        """Start of a backward pass.  The buffer is (re)sized HERE, from what the previous pass asked for in total --
        never inside a pass, where slices are in use (a pass may flush more than once: the decoder-side flush of the
        overlapped all-reduce comes first)."""
        if self.total > (self.buf.numel() if self.buf is not None else 0) and device is not None:
            if self.frozen:
                raise StylerHipError("WgradArena: a captured hipGraph addresses this buffer; it cannot be re-sized")
            self.buf = torch.empty(self.total, device=device, dtype=torch.float32)
        self.used = 0
        self.total = 0
        self.descs = []
        self.group, self.group_keep = [], []
        self.flush_no = 0
        self.side_used, self.side_keep = False, []

    def side_stream(self, device):
        """The weight-gradient stream, ordered behind everything enqueued so far on the current one."""
        if self.side is None:
            self.side = torch.cuda.Stream(device=device)
        self.side.wait_stream(torch.cuda.current_stream())
        self.side_used = True
        return self.side

    def take(self, nfloats, device):
        nfloats = (nfloats + 3) & ~3
        self.total += nfloats
        if self.buf is None or self.used + nfloats > self.buf.numel():
            return None                              # measuring pass (or overflow): caller reduces immediately
        out = self.buf[self.used:self.used + nfloats]
        self.used += nfloats
        return out

    # EXPERIMENT (measured, not the default): ALL bf16 weight gradients of a backward pass as members of one grouped launch
    # per kernel variant, with a split count derived from the group's size (GROUP_BLOCKS / tiles of the group in the previous
    # pass).  Fewer partial tiles, but at 1-3 splits a group is only ~1.5 rounds of blocks at 2 blocks per CU, and the tail
    # costs more than the partial-tile traffic saves: 15.76 ms (768 blocks), 15.59 (1536), 15.49 (stand-alone split counts)
    # vs 15.1-15.2 ms for the interleaved stand-alone launches on the same box type.
    group_all = __import__("os").environ.get("STYLER_WGRAD_GROUP_ALL", "0") == "1"
    GROUP_BLOCKS = int(__import__("os").environ.get("STYLER_WGRAD_GROUP_TOTAL", "1536"))   # blocks one grouped launch aims for

    def want_splits(self, variant):
        """Split-K count for a new member of the group (current flush, `variant`): GROUP_BLOCKS over the tiles that group
        had in the previous pass; 0 (= the stand-alone policy) while there is no history."""
        tiles = self.hist.get((self.flush_no, variant))
        if not tiles:
            return 0
        return max(1, (self.GROUP_BLOCKS + tiles // 2) // tiles)

    def _top_up_pinned(self, capturing):
        """Pinned staging buffers cannot be allocated while a hipGraph is being captured (and a captured memcpy node
        re-reads its buffer at every replay, so each table owns one): keep a reserve from the eager steps -- a capture
        may flush more than once (split step) and may be retried."""
        if not capturing:
            while len(self._pinned_pool) < 8:
                self._pinned_pool.append(torch.empty(self.GROUP_TABLE_BYTES, dtype=torch.uint8).pin_memory())

    def flush(self, device):
        import ctypes
        import numpy as np
        if loss_side_stream is not None:             # rt.pred_stream: weight-gradient partials written on the loss-only side
            torch.cuda.current_stream().wait_stream(loss_side_stream)   # stream are folded below (a no-op edge once joined)
        if self.side_used:                           # join: the partial tiles written on the side stream are read below
            torch.cuda.current_stream().wait_stream(self.side)
            self.side_used, self.side_keep = False, []
        if self.group:
            from ._lib import WgradGroupDesc
            # one launch per kernel variant; inside a launch the heavy members (long K loops) come first, members with
            # >= 8 splits start on a multiple of 8 blocks (their split -> XCD map, gemm_bwd.hip)
            members = sorted(self.group, key=lambda d: (d.variant, -d.cps * d.kw))
            launches, tiles = [], {}
            start, first = 0, 0
            for i, d in enumerate(members):
                if i and d.variant != members[i - 1].variant:
                    launches.append((members[first].variant, first, i - first, start))
                    start, first = 0, i
                if d.splits >= 8:
                    start = (start + 7) & ~7
                d.block_start = start
                start += d.nblocks
                tiles[d.variant] = tiles.get(d.variant, 0) + d.tiles
            launches.append((members[first].variant, first, len(members) - first, start))
            for v, t in tiles.items():
                self.hist[(self.flush_no, v)] = t
            arr = (WgradGroupDesc * len(members))(*members)
            key = bytes(arr)
            capturing = torch.cuda.is_current_stream_capturing()
            self._top_up_pinned(capturing)
            if not capturing:
                # eager step: operand addresses differ from step to step, so the table is simply uploaded (a blocking
                # copy of a few KB); nothing is cached and no pinned buffer is consumed (pinning memory costs ~10 ms a piece)
                table = torch.frombuffer(bytearray(key), dtype=torch.uint8).to(device)
            else:
                if key not in self._gcache:
                    if len(key) > self.GROUP_TABLE_BYTES or not self._pinned_pool:
                        raise StylerHipError("grouped wgrad: no pinned staging buffer (run one eager step before capturing)")
                    host = self._pinned_pool.pop()   # owned by this cache entry from now on: the captured memcpy node
                    host[:len(key)].copy_(torch.frombuffer(bytearray(key), dtype=torch.uint8))   # re-reads it at every replay
                    dev_t = torch.empty(len(key), device=device, dtype=torch.uint8)
                    dev_t.copy_(host[:len(key)], non_blocking=True)
                    self._gcache[key] = (host, dev_t)
                table = self._gcache[key][1]
            esz = ctypes.sizeof(WgradGroupDesc)
            for variant, first, count, blocks in launches:
                _chk(lib.styler_wgrad_group(table.data_ptr() + first * esz, count, blocks, variant, _stream()),
                     "styler_wgrad_group")
            self.group, self.group_keep = [], []
        self.flush_no += 1
        if not self.descs:
            return
        descs, self.descs = self.descs, []
        self._reduce(descs, device)

    def _reduce(self, all_descs, device):
        import numpy as np
        # Descriptors that add into the SAME gradient from adjacent workspace slices (the three calls of a bf16x3 weight
        # gradient) become one descriptor with the split counts summed; other repeats (a module applied twice in one step:
        # the augmentation classifiers of the main and the DAT pass) go to a FOLLOWING launch -- two blocks of one launch
        # must never read-modify-write the same dw elements (until round 4 such pairs shared a launch and were correct
        # only because their blocks ran thousands of blocks apart).
        merged = []
        for d in all_descs:
            ws, dw, sn, sc, sj, n, cin, kw, splits = d
            if merged:
                pw, pdw, psn, psc, psj, pn, pcin, pkw, psp = merged[-1]
                if (pdw, psn, psc, psj, pn, pcin, pkw) == (dw, sn, sc, sj, n, cin, kw) and ws == pw + psp * n * cin * kw * 4:
                    merged[-1] = (pw, pdw, psn, psc, psj, pn, pcin, pkw, psp + splits)
                    continue
            merged.append(d)
        rounds = []
        for d in merged:
            for r in rounds:
                if d[1] not in r[1]:
                    r[0].append(d); r[1].add(d[1])
                    break
            else:
                rounds.append(([d], {d[1]}))
        for descs, _ in rounds:
            key = tuple(descs)
            if key not in self._cache:
                dt = np.dtype([("ws", np.uint64), ("dw", np.uint64), ("sn", np.int64), ("sc", np.int64), ("sj", np.int64),
                               ("block_start", np.int64), ("n", np.int32), ("cin", np.int32), ("kw", np.int32),
                               ("splits", np.int32)])
                arr = np.zeros(len(descs), dtype=dt)
                start = 0
                owners = []                          # which descriptor owns which block (the kernel would search for it)
                for i, (ws, dw, sn, sc, sj, n, cin, kw, splits) in enumerate(descs):
                    arr[i] = (ws, dw, sn, sc, sj, start, n, cin, kw, splits)
                    nb = int(lib.styler_wgrad_reduce_blocks(n, cin, kw, sc, sj))
                    owners.append(np.full(nb, i, dtype=np.int32))
                    start += nb
                if len(self._cache) > 8 and not self.owned_by_graph:
                    self._cache.clear()              # eager steps only: a graph's tables live as long as its arena
                self._cache[key] = (torch.from_numpy(arr.view(np.uint8).copy()).to(device), start,
                                    torch.from_numpy(np.concatenate(owners)).to(device))
            table, blocks, bmap = self._cache[key]
            _chk(lib.styler_wgrad_reduce_multi_map(table.data_ptr(), len(descs), blocks, bmap.data_ptr() if block_maps else None, _stream()),
                 "styler_wgrad_reduce_multi")


wgrad_arena = None
# round 6: the multi-descriptor launches (derived-layout refresh, fold of the split-K partials) get the owner of every block from the
# host instead of searching the descriptor table per block (STYLER_BLOCKMAP=0: the search)
block_maps = os.environ.get("STYLER_BLOCKMAP", "1") != "0"


class GemmProfiler:
    """Optional HIP-event bracket around every styler_conv_gemm launch (bench.py's live roofline
    measurement).  Events are recorded on the launch stream; elapsed times are read after a sync."""

    def __init__(self):
        self.records = []          # (variant, flops, start_event, end_event, packed)
        self.shapes = []           # per record: (B, L, cin, n, kw, io, act) -- tools/step_gemms.py

    def summary(self, packed_fraction=1.0):
        """`packed_fraction` = valid rows / row capacity of the packed decoder tensors: launches on packed rows are
        recorded with the capacity (the valid count lives on the device) and scaled here to ALGORITHMIC flops."""
        out = {}
        for var, flops, e0, e1, packed in self.records:
            d = out.setdefault(var, {"launches": 0, "flops": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops * (packed_fraction if packed else 1.0)
            d["ms"] += e0.elapsed_time(e1)
        return out


gemm_profiler = None


class ZeroSlab:
    """One fp64 buffer cleared ONCE per step, from which the normalisation kernels take their (must-be-zero) statistics
    workspaces: 68 memset launches per training step become one.  Sized from the previous step's demand."""

    def __init__(self):
        self.buf = None
        self.used = 0
        self.total = 0
        self.frozen = False                          # set once a captured hipGraph addresses `buf`

    def begin(self, device, fill=True):
        """`fill=False`: the caller clears `buf` itself (training.forward_backward: one launch with the gradient clear)."""
        if self.total > (self.buf.numel() if self.buf is not None else 0):
            if self.frozen:
                raise StylerHipError("ZeroSlab: a captured hipGraph addresses this buffer; it cannot be re-sized")
            self.buf = torch.empty(self.total, device=device, dtype=torch.float64)
        if self.buf is not None and fill:
            self.buf.zero_()
        self.used = 0
        self.total = 0
        return self.buf

    def take(self, n):
        n = (n + 1) & ~1
        self.total += n
        if self.buf is None or self.used + n > self.buf.numel():
            return None
        out = self.buf[self.used:self.used + n]
        self.used += n
        return out


zero_slab = None            # set by training.forward_backward for the duration of a step


def _norm_ws(n, device):
    """(workspace of n doubles, already-zero flag)"""
    if zero_slab is not None:
        ws = zero_slab.take(n)
        if ws is not None:
            return ws, 1
    return torch.empty(n, device=device, dtype=torch.float64), 0


class PackPlan:
    """Index tables of the packed-rows layout (styler_pack_plan): the valid rows of all items back to back in a
    [1, B*T, C] tensor.  `nrows` (int64 [1], device) is the `lens` argument of the row-wise ops on packed tensors."""

    def __init__(self, lens, B, T):
        dev = lens.device
        assert lens.dtype == torch.int64 and lens.is_contiguous() and lens.numel() == B
        self.B, self.T, self.rows, self.lens = B, T, B * T, lens
        self.cu = torch.empty(B + 1, device=dev, dtype=torch.int32)
        self.rowinfo = torch.empty(B * T, 2, device=dev, dtype=torch.int32)
        self.chunktab = torch.empty(B * T // 64 + B + 1, 4, device=dev, dtype=torch.int32)
        self.counts = torch.empty(2, device=dev, dtype=torch.int64)
        _chk(lib.styler_pack_plan(lens.data_ptr(), B, T, self.cu.data_ptr(), self.rowinfo.data_ptr(),
                                  self.chunktab.data_ptr(), self.counts.data_ptr(), _stream()), "styler_pack_plan")
        self.nrows = self.counts[0:1]


IO_X16, IO_Y16, IO_MASK16, IO_RES16 = 1, 2, 4, 8


def _pk_io(padded, packed):
    return (IO_X16 if padded.dtype == torch.bfloat16 else 0) | (IO_Y16 if packed.dtype == torch.bfloat16 else 0)


def pack_rows(x, plan, add=None, out_bf16=False):
    """[B, T, C] padded -> [1, B*T, C] packed (+ add[t] per row, the positional table).  `out_bf16`: the packed tensor is
    stored as bf16 (the decoder's residual stream in throughput mode); x may be fp32 or bf16."""
    B, T, C = x.shape
    assert (B, T) == (plan.B, plan.T) and (add is None or (add.shape[0] >= T and add.shape[1] == C and add.is_contiguous()))
    out = torch.empty(1, B * T, C, device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    _chk(lib.styler_pack_rows(x.data_ptr(), _ld(x), out.data_ptr(), C, _ptr(add), plan.cu.data_ptr(), B, T, C, _pk_io(x, out),
                              _stream()), "styler_pack_rows")
    return out


def pack_rows_pair(xa, xb, plan, add=None, out_bf16=False):
    """Two padded [B, T, C] tensors -> one packed [1, 2B*T, C] tensor whose items 0..B-1 come from `xa` and B..2B-1 from
    `xb` (`plan` is the PackPlan of the 2B items): the clean and the noisy decode of styler.py:52,55 as ONE batch."""
    B, T, C = xa.shape
    assert xb.shape == xa.shape and (2 * B, T) == (plan.B, plan.T)
    assert add is None or (add.shape[0] >= T and add.shape[1] == C and add.is_contiguous())
    out = torch.empty(1, 2 * B * T, C, device=xa.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    for half, x in enumerate((xa, xb)):
        _chk(lib.styler_pack_rows(x.data_ptr(), _ld(x), out.data_ptr(), C, _ptr(add),
                                  plan.cu.data_ptr() + 4 * B * half, B, T, C, _pk_io(x, out), _stream()), "styler_pack_rows")
    return out


def unpack_rows_pair(xp, plan):
    """Inverse of pack_rows_pair (also its backward): packed [1, 2B*T, C] (fp32 or bf16) -> two padded fp32 [B, T, C] tensors."""
    C = xp.shape[-1]
    xp = _rows_view(xp)
    B = plan.B // 2
    outs = []
    for half in range(2):
        out = torch.empty(B, plan.T, C, device=xp.device, dtype=torch.float32)
        _chk(lib.styler_unpack_rows(xp.data_ptr(), _ld(xp), out.data_ptr(), C, plan.cu.data_ptr() + 4 * B * half, B,
                                    plan.T, C, _pk_io(out, xp), _stream()), "styler_unpack_rows")
        outs.append(out)
    return outs


def unpack_rows(xp, plan):
    """[1, B*T, C] packed (fp32 or bf16) -> fp32 [B, T, C] padded with zeros at t >= len[b]."""
    C = xp.shape[-1]
    xp = _rows_view(xp)
    out = torch.empty(plan.B, plan.T, C, device=xp.device, dtype=torch.float32)
    _chk(lib.styler_unpack_rows(xp.data_ptr(), _ld(xp), out.data_ptr(), C, plan.cu.data_ptr(), plan.B, plan.T, C,
                                _pk_io(out, xp), _stream()), "styler_unpack_rows")
    return out


# bf16x3 mode, inside a training step (training.forward_backward sets a dict): the split of a tensor is kept until the step
# ends, so the activation a GEMM split in the forward is not split again for that layer's weight gradient in backward (a
# third of the step's split passes).  An entry holds the source tensor too: its memory cannot be handed to another tensor
# while the entry lives, so (address, shape, strides, version) identifies the VALUES that were split.
x3_cache = None
# the compact [hi | lo] split where the channel count allows it (STYLER_X3_COMPACT=0: always the triple form)
x3_compact = os.environ.get("STYLER_X3_COMPACT", "1") != "0"
# bf16x3 mode: independent small Linears as grouped launches (STYLER_X3_GROUPED=0: one launch per Linear)
x3_grouped = os.environ.get("STYLER_X3_GROUPED", "1") != "0"


def split3(x, plan=None):
    """[.., C] fp32 -> the bf16x3 split of the rows (styler_split3_bf16), the activation operand of a bf16x3 GEMM: the compact
    [.., 2C] = [hi | lo] when C % 64 == 0 (the GEMM reads hi a second time for the third product, STYLER_IO_X3A), else the
    triple [.., 3C] = [hi | lo | hi].  With `plan` (packed rows) only the valid prefix is written."""
    cache = x3_cache
    if cache is not None:
        # (the plan's identity is part of the key: the valid prefix written depends on it.  Invariant: a tensor that has been
        #  split is never an `out=` target of a raw-pointer kernel later in the same step -- the library's kernels do not
        #  bump `_version`; every x3 operand of the tape is a freshly allocated activation / gradient)
        key = (x.data_ptr(), tuple(x.shape), x.stride(), plan.counts.data_ptr() if plan is not None else 0)
        hit = cache.get(key)
        if hit is not None and hit[1] == x._version:
            return hit[2]
    C = x.shape[-1]
    rows = x.numel() // C
    parts = 2 if (C % 64 == 0 and x3_compact) else 3
    y = torch.empty(*x.shape[:-1], parts * C, device=x.device, dtype=torch.bfloat16)
    _chk(lib.styler_split3_bf16(_f32(x).data_ptr(), _ld(x), y.data_ptr(), rows, C,
                                plan.counts.data_ptr() if plan is not None else None, parts, _stream()), "styler_split3_bf16")
    if cache is not None:
        cache[key] = (x, x._version, y)
    return y


def split3_multi(xs):
    """split3 of several tensors (no packed plan) with ONE launch per eight that are not in the step's split cache."""
    from ._lib import CopySeg
    outs, todo = [None] * len(xs), []
    cache = x3_cache
    for i, x in enumerate(xs):
        key = (x.data_ptr(), tuple(x.shape), x.stride(), 0)
        hit = cache.get(key) if cache is not None else None
        if hit is not None and hit[1] == x._version:
            outs[i] = hit[2]
            continue
        C = x.shape[-1]
        parts = 2 if (C % 64 == 0 and x3_compact) else 3
        y = torch.empty(*x.shape[:-1], parts * C, device=x.device, dtype=torch.bfloat16)
        outs[i] = y
        todo.append((_f32(x), y, parts, key))
    for j in range(0, len(todo), 8):
        part = todo[j:j + 8]
        if len(part) == 1:
            x, y, parts, _ = part[0]
            C = x.shape[-1]
            _chk(lib.styler_split3_bf16(x.data_ptr(), _ld(x), y.data_ptr(), x.numel() // C, C, None, parts, _stream()),
                 "styler_split3_bf16")
            continue
        arr = (CopySeg * len(part))()
        for k, (x, y, parts, _) in enumerate(part):
            C = x.shape[-1]
            arr[k].src, arr[k].dst, arr[k].ld_src, arr[k].ld_dst = x.data_ptr(), y.data_ptr(), _ld(x), parts * C
            arr[k].rows, arr[k].C, arr[k]._pad = x.numel() // C, C, parts
        _chk(lib.styler_split3_multi(arr, len(part), _stream()), "styler_split3_multi")
    if cache is not None:
        for x, y, _, key in todo:
            cache[key] = (x, x._version, y)
    return outs


# Round 5: producers write the split of their fp32 output themselves (GEMM epilogues and norms: styler_set_x3_out) and file it in the step's split cache, so the consumer's split3() is a lookup -- no pass re-reads the tensor.
# STYLER_X3_PRODUCERS=0: every split is a separate pass again (A/B switch, the tests compare both).
x3_producers = os.environ.get("STYLER_X3_PRODUCERS", "1") != "0"


def x3_out_for(shape_prefix, C, device):
    """(split tensor [.., parts * C] bf16, parts) a producer should fill for an fp32 output [.., C] -- or (None, 0) outside a
    bf16x3 training step (no cache to file it in) or with STYLER_X3_PRODUCERS=0."""
    if x3_cache is None or not x3_producers or C % 4:
        return None, 0
    parts = 2 if (C % 64 == 0 and x3_compact) else 3
    return torch.empty(*shape_prefix, parts * C, device=device, dtype=torch.bfloat16), parts


def x3_register(y, y3, plan=None):
    """Files `y3`, the split a producer wrote for its fp32 output `y`, under the key split3(y, plan) looks up."""
    if y3 is not None and x3_cache is not None:
        key = (y.data_ptr(), tuple(y.shape), y.stride(), plan.counts.data_ptr() if plan is not None else 0)
        x3_cache[key] = (y, y._version, y3)


def _x3_begin(out, enable=True, allowed_parts=(2, 3)):
    """Registers a split output for the NEXT producer call (styler_set_x3_out) when `out` (fp32, contiguous rows) is going to be
    a bf16x3 GEMM operand in this training step; returns the split tensor to hand to `_x3_end`, or None.
    `allowed_parts`: the storage forms THIS producer can write -- the x3 attention kernels and layernorm_bwd only know the
    compact [hi | lo] form (parts = 2); with STYLER_X3_COMPACT=0 (triple form everywhere) they do not register, and the
    consumer's split3() makes the pass instead (round-5 advisor: they used to return STYLER_EINVAL there)."""
    if not enable or out is None or out.dtype != torch.float32 or not out.is_contiguous():
        return None
    y3, parts = x3_out_for(out.shape[:-1], out.shape[-1], out.device)
    if y3 is None or parts not in allowed_parts:
        return None
    _chk(lib.styler_set_x3_out(y3.data_ptr(), parts), "styler_set_x3_out")
    return y3


def _x3_end(out, y3, plan=None):
    if y3 is not None:
        lib.styler_set_x3_out(None, 0)               # (consumed by the producer; cleared again in case it never got there)
        x3_register(out, y3, plan)


def _x3_abort(y3):
    """Error path of a producer call: drop the thread-local registration, so that the NEXT producer call of this thread cannot
    write a split into a buffer that is about to be freed (round-5 advisor)."""
    if y3 is not None:
        lib.styler_set_x3_out(None, 0)


def lo_part(x, plan=None):
    """bf16(x - float(bf16(x))) stored as fp32 (styler_lo_part): the low operand of a bf16x3 weight gradient."""
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)   # (rows behind a packed prefix are never read: the
    # weight-gradient kernels address an item through a descriptor that ends with the item)
    _chk(lib.styler_lo_part(_f32(x).data_ptr(), _ld(x), y.data_ptr(), rows, C,
                            plan.counts.data_ptr() if plan is not None else None, _stream()), "styler_lo_part")
    return y


def conv_gemm(x, w, bias=None, *, kw=1, n=None, act=ACT_NONE, prec=PREC_F32, scale=None, res=None,
              out=None, lens=None, plan=None, mask=None, out_bf16=False, _x3a=False, x3_out=False):
    """y = act(scale * conv1d_same(x, w) + bias) (+ res); x [B, L, cin] -> y [B, L, n].
    `w` is the kernel-layout weight [n, kw*cin] (fp32, or bf16 when prec == PREC_BF16).
    Throughput mode only: x, the output (`out_bf16` / a bf16 `out`) and `mask` may be bf16 tensors (the FFN hidden
    activation and its gradient are stored that way; no residual with a bf16 output)."""
    if prec == PREC_BF16X3:
        # fp32-class products on the bf16 engines: x -> activation blocks (hi, lo, hi) (3 cin channels; compact [hi | lo] + STYLER_IO_X3A) against the weight's [w_hi | w_hi | w_lo]
        # rows (runtime.gemm_weight); everything behind the contraction -- bias, activation, residual, masks -- is unchanged
        x3 = x if x.dtype == torch.bfloat16 else split3(x, plan)      # (a bf16 x is a split3 tensor the caller shares)
        cin3 = w.shape[-1] // kw                                       # 3 C
        if w.dtype != torch.bfloat16 or w.shape[-1] != kw * cin3 or cin3 % 3 or x3.shape[-1] not in (cin3, cin3 // 3 * 2):
            raise StylerHipError("bf16x3 GEMM needs the x3 weight layout [n, kw * 3 cin] and a split3 activation")
        return conv_gemm(x3, w, bias, kw=kw, n=n, act=act, prec=PREC_BF16, scale=scale, res=res, out=out,
                         lens=lens, plan=plan, mask=mask, out_bf16=False, _x3a=x3.shape[-1] != cin3, x3_out=x3_out)
    B, L, cin = x.shape
    if _x3a:                                           # compact [hi | lo] rows: the contraction still runs over 3 C channels
        cin = cin // 2 * 3
    n = w.shape[0] if n is None else n
    if prec == PREC_BF16 and (w.dtype != torch.bfloat16 or cin % 8):
        raise StylerHipError("bf16 GEMM needs a bf16 weight shadow and cin % 8 == 0")
    if out is None:
        out = torch.empty(B, L, n, device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    io = (1 if x.dtype == torch.bfloat16 else 0) | (2 if out.dtype == torch.bfloat16 else 0) | \
         (4 if mask is not None and mask.dtype == torch.bfloat16 else 0) | \
         (8 if res is not None and res.dtype == torch.bfloat16 else 0) | (IO_X3A if _x3a else 0)
    if io:
        if prec != PREC_BF16:
            raise StylerHipError("bf16 activations only in throughput mode")
    else:
        _f32(x)
    prof = gemm_profiler
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = None
    if prec == PREC_BF16 and ((io & 1) or kw >= 3):  # split-K launches (256 x 256 engine / small-M 64 x 64 tile) want scratch for their partial tiles
        need = lib.styler_conv_gemm_workspace_bytes(B, L, cin, n, kw, act, prec, io, _ld(x), int(plan is not None),
                                                    int(mask is not None))
        if need and lens is None:
            ws = torch.empty(need, device=x.device, dtype=torch.uint8)
            lib.styler_gemm_set_workspace(ws.data_ptr(), need)
            if (io & 1) and n % 256 == 0:            # the 256 x 256 engine finishes its split-K = 2 launches in the kernel when it
                ncnt = 2 * ((B * L + 255) // 256) * (n // 256)     # gets zeroed tile counters (it leaves them zero)
                cnt = zero_slab.take((ncnt + 1) // 2) if zero_slab is not None else None
                if cnt is None:
                    cnt = torch.zeros(ncnt, device=x.device, dtype=torch.int32)
                lib.styler_gemm_set_counters(cnt.data_ptr(), ncnt)
    y3 = None
    if x3_out and prec == PREC_BF16 and out.dtype == torch.float32 and out.is_contiguous() and prof is None:
        # (bf16x3: this output is the activation operand of a following GEMM -- its split leaves with the epilogue)
        y3, parts3 = x3_out_for(out.shape[:-1], n, x.device)
        if y3 is not None:
            _chk(lib.styler_set_x3_out(y3.data_ptr(), parts3), "styler_set_x3_out")
    if plan is not None:                             # packed rows: taps stay inside their item, tiles behind the data skip
        assert B == 1 and L == plan.rows
        _chk(lib.styler_conv_gemm_packed(x.data_ptr(), _ld(x), w.data_ptr(), _ptr(scale), _ptr(bias), _ptr(res),
                                         _ld(res) if res is not None else 0, out.data_ptr(), _ld(out), L, cin, n, kw,
                                         act, prec, plan.counts.data_ptr(), plan.rowinfo.data_ptr(), _ptr(mask),
                                         _ld(mask) if mask is not None else 0, io, _stream()),
             "styler_conv_gemm_packed")
    else:
        _chk(lib.styler_conv_gemm(x.data_ptr(), _ld(x), w.data_ptr(), _ptr(scale), _ptr(bias), _ptr(res),
                                  _ld(res) if res is not None else 0, out.data_ptr(), _ld(out), B, L, cin, n,
                                  kw, act, prec, _ptr(lens), _ptr(mask), _ld(mask) if mask is not None else 0, io,
                                  _stream()), "styler_conv_gemm")
    if ws is not None:
        lib.styler_gemm_set_workspace(None, 0)       # (consumed by the call above; cleared again in case it never got there)
        lib.styler_gemm_set_counters(None, 0)
    if y3 is not None:
        lib.styler_set_x3_out(None, 0)
        x3_register(out, y3, plan)
    if prof is not None:
        e1.record()
        prof.records.append((lib.styler_conv_gemm_engine2(B, L, cin, n, kw, prec, io, _ld(x), int(plan is not None), act,
                                                          int(mask is not None)) if ws is not None or not (io & 1)
                             else lib.styler_conv_gemm_engine(B, L, cin, n, kw, prec, io, _ld(x), int(plan is not None)),
                             2.0 * B * L * n * kw * cin, e0, e1, plan is not None))
        prof.shapes.append((B, L, cin, n, kw, io, act))
    return out


def conv_gemm_multi(calls):
    """Several INDEPENDENT Linear-shaped GEMMs -> list of outputs.  `calls`: dicts with the keyword arguments of `conv_gemm`
    (x, w, bias, n, act, prec, scale, res, out).  The members that take the 64 x 64 bf16 tile (k = 1) run as ONE launch
    (styler_conv_gemm_group, <= 8 members each): in throughput mode those with fp32 activations in and out, in the bf16x3
    arithmetic every member (its activation is split first; the grouped kernel reads the split, STYLER_IO_X3A for the
    compact form); anything else -- fp32 mode, large shapes, bf16 storage -- is launched one by one."""
    from ._lib import GemmProblem
    outs = [None] * len(calls)
    group = []
    if x3_grouped and gemm_profiler is None and x3_cache is not None:   # bf16x3 members: their activation splits in one launch (kept in the step's cache)
        need = [c["x"] for c in calls if c.get("prec") == PREC_BF16X3 and c["x"].dtype == torch.float32 and c["x"].dim() == 3]
        if len(need) > 1:
            split3_multi(need)
    for i, c in enumerate(calls):
        x, w = c["x"], c["w"]
        B, L, cin = x.shape
        n = c.get("n") or w.shape[0]
        x3m = c.get("prec") == PREC_BF16X3 and w.dtype == torch.bfloat16 and x3_grouped and gemm_profiler is None
        if x3m:
            xs = x if x.dtype == torch.bfloat16 else split3(x)        # (a bf16 x is a split3 tensor the caller shares)
            cin = w.shape[-1]                                          # 3 C: the contraction the kernel walks
            flags = 1 | (2 if xs.shape[-1] != cin else 0)
            ok = (cin % 3 == 0 and xs.shape[-1] in (cin, cin // 3 * 2) and cin % 8 == 0 and _ld(xs) % 8 == 0
                  and (c.get("res") is None or c["res"].dtype == torch.float32)
                  and not (lib.styler_conv_gemm_variant(B, L, cin, n, 1, PREC_BF16) & 1))
            if ok:
                x = xs
        else:
            flags = 0
            ok = (c.get("prec") == PREC_BF16 and w.dtype == torch.bfloat16 and x.dtype == torch.float32 and cin % 8 == 0
                  and w.shape[-1] == cin and gemm_profiler is None
                  and (c.get("res") is None or c["res"].dtype == torch.float32)
                  and not (lib.styler_conv_gemm_variant(B, L, cin, n, 1, PREC_BF16) & 1))
        if not ok:
            outs[i] = conv_gemm(c["x"], w, c.get("bias"), n=n, act=c.get("act", ACT_NONE), prec=c.get("prec", PREC_F32),
                                scale=c.get("scale"), res=c.get("res"), out=c.get("out"))
            continue
        out = c.get("out")
        if out is None:
            out = torch.empty(B, L, n, device=x.device, dtype=torch.float32)
        outs[i] = out
        group.append((c, out, B, L, cin, n, x, flags))
    for j in range(0, len(group), 8):
        part = group[j:j + 8]
        if len(part) == 1:
            c, out, B, L, cin, n, x, flags = part[0]
            conv_gemm(c["x"], c["w"], c.get("bias"), n=n, act=c.get("act", ACT_NONE), prec=c.get("prec", PREC_BF16),
                      scale=c.get("scale"), res=c.get("res"), out=out)
            continue
        arr = (GemmProblem * len(part))()
        for k, (c, out, B, L, cin, n, x, flags) in enumerate(part):
            x, res = (x if flags else _f32(x)), c.get("res")
            m = arr[k]
            m.x, m.w, m.scale, m.shift, m.res, m.y, m.len = (x.data_ptr(), c["w"].data_ptr(), _ptr(c.get("scale")),
                                                             _ptr(c.get("bias")), _ptr(res), out.data_ptr(), None)
            m.ldx, m.ldres, m.ldy = _ld(x), (_ld(res) if res is not None else 0), _ld(out)
            m.B, m.L, m.cin, m.n, m.act, m.flags = B, L, cin, n, c.get("act", ACT_NONE), flags
        _chk(lib.styler_conv_gemm_group(arr, len(part), _stream()), "styler_conv_gemm_group")
    return outs


ACT_CRELU = 5                      # min(max(v, 0), 20): DeepSpeaker's clipped ReLU
ACT_RES_FIRST = 0x100              # y = act(scale * (acc + res) + shift): last call of a conv summed over several calls


def conv_gemm_pad(x, w, bias=None, *, kw, pad, act=ACT_NONE, prec=PREC_F32, res=None, out=None, scale=None):
    """y = act(sum_j x[t + j - pad] w[:, j, :] + bias) (+ res) with an explicit left padding and any kw <= 9; x, res and
    out may be strided row views ([B, L, C] with row stride d*C: the phase views of a dilated conv)."""
    B, L, cin = x.shape
    n = w.shape[0]
    if out is None:
        out = torch.empty(B, L, n, device=x.device, dtype=torch.float32)
    _f32(x), _f32(out)
    if prec == PREC_BF16 and w.dtype != torch.bfloat16:
        raise StylerHipError("bf16 GEMM needs a bf16 weight shadow")
    _chk(lib.styler_conv_gemm_pad(x.data_ptr(), _ld(x), w.data_ptr(), _ptr(scale), _ptr(bias), _ptr(res),
                                  _ld(res) if res is not None else 0, out.data_ptr(), _ld(out), B, L, cin, n, kw, pad,
                                  act, prec, _stream()), "styler_conv_gemm_pad")
    return out


def leaky_sum(a, b=None, c=None, *, scale=1.0, slope=0.1, out=None):
    """out = leaky_relu(scale * (a + b + c), slope) on contiguous fp32 tensors (out may alias a)."""
    assert a.is_contiguous() and (b is None or b.is_contiguous()) and (c is None or c.is_contiguous())
    if out is None:
        out = torch.empty_like(a)
    _chk(lib.styler_leaky_sum(_f32(a).data_ptr(), _ptr(b), _ptr(c), out.data_ptr(), a.numel(), float(scale),
                              float(slope), _stream()), "styler_leaky_sum")
    return out


_dropout_counter_ptr = None


def set_dropout_counter(counter):
    """Register (or, with None, unregister) the int64 device step counter mixed into every dropout seed.  The library
    keeps the raw address: whoever registers a tensor unregisters it before the tensor dies (TrainState.close)."""
    global _dropout_counter_ptr
    assert counter is None or (counter.dtype == torch.int64 and counter.is_cuda)
    _chk(lib.styler_set_dropout_counter(_ptr(counter)), "styler_set_dropout_counter")
    _dropout_counter_ptr = _ptr(counter)


def dropout_counter_is(counter):
    return counter is not None and _dropout_counter_ptr == counter.data_ptr()


def cast_bf16(src):
    dst = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
    _chk(lib.styler_cast_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "styler_cast_bf16")
    return dst


def gemm256_config(enabled=-1, min_tiles=-1, split=-1, take_all=-1):
    """Test / tuning hook of the 256 x 256 LDS-DMA GEMM engine (csrc/gemm256.hip): switch it on / off, set the tile count
    from which it takes a launch, and the test-only policy overrides `split` (0 policy, 1 never split-K, 2 split-K wherever
    the epilogue allows it) and `take_all` (1: no tile bound, no short-K guard); -1 keeps a value.  Returns the previous
    (enabled, min_tiles, split, take_all)."""
    prev = lib.styler_gemm256_config(-1, -1)
    pol = lib.styler_gemm256_policy(-1, -1)
    lib.styler_gemm256_config(int(enabled), int(min_tiles))
    lib.styler_gemm256_policy(int(split), int(take_all))
    return prev & 1, prev >> 1, pol & 3, pol >> 2


def wgrad_tune(knob, value, arenas=()):
    """styler_wgrad_tune with the host-side bookkeeping it needs: knob 1 (tall k = 5 tile) changes split counts and workspace
    bytes, so every WgradArena in `arenas` (training.TrainState.arena, ...) forgets its sizing and descriptor tables when the
    value actually changes (round-5 advisor).  Returns the previous value."""
    prev = lib.styler_wgrad_tune(int(knob), int(value))
    if prev < 0:
        raise StylerHipError(f"styler_wgrad_tune: unknown knob {knob}")
    if prev != value and value in (0, 1, 2):
        for a in arenas:
            a.reset_plan()
    return prev


def gemm256_height(ht=-1, min_tiles3=-1):
    """Tile height of the 256-column LDS-DMA engine: ht 4 = 256 rows, 3 = 192 rows, 0 = per-launch policy (-1 keeps);
    `min_tiles3`: smallest 192 x 256 tile count the policy hands to the 192-row tile.  Returns the previous (ht, min_tiles3)."""
    prev = lib.styler_gemm256_height(int(ht), int(min_tiles3))
    return prev & 7, prev >> 3


def gemm_small_split_config(enabled=-1):
    """Test / tuning hook of the 64 x 64 tile's split-K (small-M, long-K conv GEMMs in bf16 mode): on / off (-1 keeps).
    Returns the previous value."""
    return lib.styler_gemm_small_split_config(int(enabled))


def gemm_n96_config(enabled=-1, min_rows=-1):
    """Test / tuning hook of the narrow-output GEMM tile (128 x 96, bf16 mode, 64 < n <= 96): on / off and the smallest row
    count it takes (-1 keeps a value).  Returns the previous (enabled, min_rows)."""
    prev = lib.styler_gemm_n96_config(-1, -1)
    lib.styler_gemm_n96_config(int(enabled), int(min_rows))
    return prev & 1, prev >> 1


def _prec(prec):
    if prec is None:
        from .runtime import rt
        return rt.prec
    return prec


# bf16x3 mode: attention on three-product bf16 MFMAs (attention_*_x3_kernel); STYLER_ATTN_X3=0: the exact-fp32 MFMA kernels
rt_attn_x3 = os.environ.get("STYLER_ATTN_X3", "1") != "0"


def attention_fwd(qkv, lens, lse=None, prec=None, plan=None, out_bf16=False, x3=False):
    """qkv [B, L, 768] (or, with `plan`, the packed [1, B*T, 768]); lse (optional) [B, 4, L] (packed: [B, 4, T]).
    out_bf16 (throughput mode): the output is stored as bf16."""
    assert qkv.is_contiguous() and qkv.shape[2] == 768
    out_bf16 = out_bf16 and _prec(prec) == PREC_BF16
    out = torch.empty(qkv.shape[0], qkv.shape[1], 256, device=qkv.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    if qkv.dtype == torch.bfloat16 or out_bf16:         # stored bf16 (throughput mode, rt.bf16_qkv / bf16_att): bf16 kernels only
        assert _prec(prec) == PREC_BF16
        io = (1 if qkv.dtype == torch.bfloat16 else 0) | (2 if out_bf16 else 0)
        fn = lambda *a: lib.styler_attention_fwd_bf16_io(*a[:-1], io, a[-1])
    else:
        fn = {PREC_BF16: lib.styler_attention_fwd_bf16, PREC_BF16X3: lib.styler_attention_fwd_x3}.get(_prec(prec), lib.styler_attention_fwd)
        if not rt_attn_x3 and _prec(prec) == PREC_BF16X3:
            fn = lib.styler_attention_fwd
    # x3 (bf16x3, the x3 kernels only): the output feeds the out-projection GEMM -- its split leaves with it
    y3 = _x3_begin(out, x3 and fn is lib.styler_attention_fwd_x3, allowed_parts=(2,))
    try:
        if plan is not None:
            _chk(fn(qkv.data_ptr(), out.data_ptr(), _ptr(lse), plan.B, plan.T, plan.lens.data_ptr(), plan.cu.data_ptr(),
                    _stream()), "styler_attention_fwd")
        else:
            B, L, _ = qkv.shape
            _chk(fn(qkv.data_ptr(), out.data_ptr(), _ptr(lse), B, L, _ptr(lens), None, _stream()), "styler_attention_fwd")
    except Exception:
        _x3_abort(y3)
        raise
    _x3_end(out, y3, plan)
    return out


def add_layernorm(x, gamma, beta, *, res=None, lens=None, out=None, dot_w=None, dot_b=None, drop_p=0.0, drop_seed=0,
                  in_drop_p=0.0, in_drop_seed=0, sum_out=None, out16=None, out_bf16=False, x3=False, plan=None):
    """LayerNorm(dropout(x) + res) with pad-mask; with dot_w returns the [B, L] scalar head instead.  `sum_out`
    (optional) receives the pre-norm sum (what layernorm_bwd needs); `out16` (optional, a bf16 [B, L, C] tensor) a second
    copy of the output rounded to bf16, for the GEMMs that take it as their activation operand."""
    B, L, C = x.shape
    dot_out = None
    if dot_w is not None:
        dot_out = torch.empty(B, L, device=x.device, dtype=torch.float32)
    elif out is None:
        # the output follows the residual stream's storage format (a bf16 residual = the decoder's bf16 stream)
        out_bf16 = out_bf16 or (res is not None and res.dtype == torch.bfloat16)
        out = torch.empty(B, L, C, device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    _f32(x)
    ln_io = (1 if res is not None and res.dtype == torch.bfloat16 else 0) | \
            (2 if out is not None and out.dtype == torch.bfloat16 else 0) | \
            (4 if sum_out is not None and sum_out.dtype == torch.bfloat16 else 0)
    # x3: the output is the activation operand of a bf16x3 GEMM (filed under the key split3(out, plan) looks up)
    y3 = _x3_begin(out if dot_w is None else None, x3)
    _chk(lib.styler_add_layernorm(x.data_ptr(), _ld(x), _ptr(res), _ld(res) if res is not None else 0,
                                  gamma.data_ptr(), beta.data_ptr(), _ptr(out),
                                  _ld(out) if out is not None else 0, _ptr(dot_w), _ptr(dot_b),
                                  _ptr(dot_out), B, L, C, _ptr(lens), float(drop_p), int(drop_seed), float(in_drop_p),
                                  int(in_drop_seed), _ptr(sum_out), _ld(sum_out) if sum_out is not None else 0,
                                  _ptr(out16), _ld(out16) if out16 is not None else 0, ln_io, _stream()),
         "styler_add_layernorm")
    _x3_end(out, y3, plan)
    return dot_out if dot_w is not None else out


def linear_ln_ok(a, n):
    """Does csrc/linear_ln.hip take this Linear (bf16 rows `a` [B, L, K] -> n channels)?"""
    return (a.dtype == torch.bfloat16 and a.dim() == 3 and a.stride(-1) == 1 and
            bool(lib.styler_linear_ln_ok(a.shape[0] * a.shape[1], a.shape[2], int(n), _ld(a))))


def linear_ln(a, w, bias, res, gamma, beta, *, lens=None, drop_p=0.0, drop_seed=0, sum_out=None, out16=None, out=None,
              packed=False):
    """LayerNorm(dropout(a W^T + bias) + res) with pad mask as one launch (styler_linear_ln): a bf16 [B, L, K], w bf16
    [256, K]; the output follows the residual's storage format (as add_layernorm); `sum_out` receives the pre-norm sum."""
    B, L, K = a.shape
    if out is None:
        out = torch.empty(B, L, 256, device=a.device,
                          dtype=torch.bfloat16 if (res is not None and res.dtype == torch.bfloat16) else torch.float32)
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or tuple(w.shape) != (256, K):
        raise StylerHipError("linear_ln needs bf16 rows and a bf16 [256, K] weight")
    ln_io = (1 if res is not None and res.dtype == torch.bfloat16 else 0) | \
            (2 if out.dtype == torch.bfloat16 else 0) | \
            (4 if sum_out is not None and sum_out.dtype == torch.bfloat16 else 0)
    prof = gemm_profiler
    if prof is not None:                              # bench.py's live brackets: filed under its own key (not a conv_gemm engine)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _chk(lib.styler_linear_ln(a.data_ptr(), _ld(a), K, w.data_ptr(), _ptr(bias), _ptr(res),
                              _ld(res) if res is not None else 0, gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), _ld(out),
                              _ptr(sum_out), _ld(sum_out) if sum_out is not None else 0, _ptr(out16),
                              _ld(out16) if out16 is not None else 0, B, L, _ptr(lens), float(drop_p), int(drop_seed), ln_io,
                              _stream()), "styler_linear_ln")
    if prof is not None:
        e1.record()
        prof.records.append(("linear_ln", 2.0 * B * L * 256 * K, e0, e1, packed))
    return out


IO_X3A = 1024         # STYLER_IO_X3A: the activation operand of a bf16x3 GEMM is the compact [hi | lo] split
IO_PARAM_SLOTS = 512  # STYLER_IO_PARAM_SLOTS: parameter gradients leave the kernel as per-block slots (see _param_slots)
IO_Z_BF16 = 16       # STYLER_IO_Z_BF16: the tensor a norm kernel normalises (a convolution's output) is stored as bf16


def groupnorm_z_bf16_ok(L):
    """A bf16 convolution output can go through GroupNorm forward AND backward (single-pass kernels only)."""
    return 0 < L <= lib.styler_groupnorm_fused_rows(1)


def groupnorm_relu(x, gamma, beta, out=None, stats=None, x3=False):
    """`stats` (optional, [B, C/16, 2] fp32) receives mean / rstd of every group for the backward.  x may be bf16
    (groupnorm_z_bf16_ok) -- then `out` must be given."""
    B, L, C = x.shape
    if out is None:
        out = x
    ws, z = _norm_ws(B * (C // 16) * 2, x.device)
    io = (2 if out.dtype == torch.bfloat16 else 0) | (IO_Z_BF16 if x.dtype == torch.bfloat16 else 0)
    y3 = _x3_begin(out, x3)
    _chk(lib.styler_groupnorm_relu(x.data_ptr(), _ld(x), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                   _ld(out), _ptr(stats), ws.data_ptr(), z, B, L, C, io, _stream()), "styler_groupnorm_relu")
    _x3_end(out, y3)
    return out


def bn_fold(gamma, beta, running_mean, running_var, conv_bias):
    C = gamma.numel()
    scale = torch.empty(C, device=gamma.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    _chk(lib.styler_bn_fold(gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                            _ptr(conv_bias), scale.data_ptr(), shift.data_ptr(), C, _stream()), "styler_bn_fold")
    return scale, shift


# scratch replicas of LayerNorm's parameter gradients: one per block of styler_layernorm_bwd (<= STYLER_LNBWD_BLOCKS = 256 blocks),
# so the blocks STORE instead of adding atomically -- never fewer slots than that cap (with fewer the kernel would fall back to
# fp32 atomics into slot block % replicas)
LN_REPLICAS = max(int(os.environ.get("STYLER_LN_REPLICAS", "256")), int(os.environ.get("STYLER_LNBWD_BLOCKS", "256")))
def _bn_ws(rows, C, segs, device):
    """BatchNorm's column-statistics workspace: one [2C]-double slot per (segment, 128-row chunk), STORED by the statistics
    kernels and folded in a fixed order (round 5: no fp64 atomics, nothing to zero -- not taken from the zero slab)."""
    return torch.empty(int(lib.styler_bn_workspace_doubles(rows, C, segs)), device=device, dtype=torch.float64), 0


def batchnorm_train(x, gamma, beta, running_mean, running_var, act, drop_p=0.0, drop_seed=0, segs=1, out_bf16=False,
                    x3=False):
    """x [B, L, C] contiguous (conv output incl. bias). Returns y (= dropout(act(BN(x))) with drop_p > 0), save_mean,
    save_rstd ([segs, C]: `segs` equal row ranges, each normalised with its own batch statistics)."""
    assert x.is_contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    mean = torch.empty(segs, C, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    ws, z = _bn_ws(rows, C, segs, x.device)
    y3 = _x3_begin(y, x3)
    _chk(lib.styler_batchnorm_train(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                    mean.data_ptr(), rstd.data_ptr(), _ptr(running_mean), _ptr(running_var),
                                    ws.data_ptr(), z, rows, C, act, float(drop_p), int(drop_seed), int(segs),
                                    (2 if out_bf16 else 0) | (IO_Z_BF16 if x.dtype == torch.bfloat16 else 0), _stream()),
         "styler_batchnorm_train")
    _x3_end(y, y3)
    return y, mean, rstd


def embed_pos(text, emb, pe):
    B, L = text.shape
    out = torch.empty(B, L, emb.shape[1], device=emb.device, dtype=torch.float32)
    assert text.dtype == torch.int64 and text.is_contiguous() and pe.shape[0] >= L
    _chk(lib.styler_embed_pos(text.data_ptr(), emb.data_ptr(), pe.data_ptr(), out.data_ptr(), B, L,
                              emb.shape[1], _stream()), "styler_embed_pos")
    return out


def add_pos(x, pe):
    B, L, C = x.shape
    assert pe.shape[0] >= L
    out = torch.empty(B, L, C, device=x.device, dtype=torch.float32)
    _chk(lib.styler_add_pos(x.data_ptr(), _ld(x), pe.data_ptr(), out.data_ptr(), B, L, C, _stream()),
         "styler_add_pos")
    return out


def gather_rows(ids, table):
    """nn.Embedding lookup: ids int32 [..] -> [.., C] rows of `table` [V, C]."""
    assert ids.dtype == torch.int32 and ids.is_contiguous() and table.is_contiguous()
    out = torch.empty(*ids.shape, table.shape[1], device=table.device, dtype=torch.float32)
    _chk(lib.styler_gather_rows(ids.data_ptr(), _f32(table).data_ptr(), out.data_ptr(), ids.numel(), table.shape[1],
                                _stream()), "styler_gather_rows")
    return out


def sinusoid_table(L, C, device):
    pe = torch.empty(L, C, device=device, dtype=torch.float32)
    _chk(lib.styler_sinusoid_table(pe.data_ptr(), L, C, _stream()), "styler_sinusoid_table")
    return pe


def onehot_conv5(v, wt, bias, out, err_flag=None, idx_out=None):
    B, L = v.shape
    C = bias.numel()
    assert v.is_contiguous()
    _chk(lib.styler_onehot_conv5(v.data_ptr(), wt.data_ptr(), bias.data_ptr(), out.data_ptr(), _ld(out),
                                 _ptr(idx_out), _ptr(err_flag), B, L, C, _stream()), "styler_onehot_conv5")
    return out


def mel_calibrate(x, mel_len, src_len, S):
    B, T, C = x.shape
    y = torch.empty(B, S, C, device=x.device, dtype=torch.float32)
    _chk(lib.styler_mel_calibrate_io(x.data_ptr(), _ld(x), y.data_ptr(), _ld(y), mel_len.data_ptr(),
                                     src_len.data_ptr(), B, T, S, C, 1 if x.dtype == torch.bfloat16 else 0, _stream()),
         "styler_mel_calibrate")
    return y


def lstm_bidir(gx, w_hh, H, cell_out=None, gates_out=None):
    B, S, _ = gx.shape
    assert gx.is_contiguous() and gx.shape[2] == 8 * H and w_hh.is_contiguous()
    out = torch.empty(B, S, 2 * H, device=gx.device, dtype=torch.float32)
    _chk(lib.styler_lstm_bidir(gx.data_ptr(), w_hh.data_ptr(), out.data_ptr(), _ptr(cell_out), _ptr(gates_out),
                               B, S, H, _stream()), "styler_lstm_bidir")
    return out


def lstm_bidir_multi(gxs, w_hhs, Hs, save=False, parts=0):
    """Up to 4 independent BiLSTM layers in one launch.  Returns outs (and, with save=True, cells and gates).
    `parts`: 0 = the VALU recurrence (fp32 arithmetic), 1 / 3 = the MFMA recurrence with bf16 / bf16x3 products (round 6)."""
    from ._lib import LstmDesc
    n = len(gxs)
    B, S, _ = gxs[0].shape
    dev = gxs[0].device
    outs = [torch.empty(B, S, 2 * H, device=dev, dtype=torch.float32) for H in Hs]
    cells = [torch.empty(B, S, 2 * H, device=dev, dtype=torch.float32) for H in Hs] if save else [None] * n
    gates = [torch.empty(B, S, 8 * H, device=dev, dtype=torch.float32) for H in Hs] if save else [None] * n
    descs = (LstmDesc * n)()
    for i in range(n):
        assert gxs[i].is_contiguous() and w_hhs[i].is_contiguous() and gxs[i].shape[2] == 8 * Hs[i]
        descs[i] = LstmDesc(gxs[i].data_ptr(), w_hhs[i].data_ptr(), outs[i].data_ptr(), _ptr(cells[i]), _ptr(gates[i]),
                            Hs[i], 0)
    if parts:
        _chk(lib.styler_lstm_bidir_multi_mfma(descs, n, B, S, parts, _stream()), "styler_lstm_bidir_multi_mfma")
    else:
        _chk(lib.styler_lstm_bidir_multi(descs, n, B, S, _stream()), "styler_lstm_bidir_multi")
    return (outs, cells, gates) if save else outs


def lstm_bidir_bwd_multi(douts, gates, cells, w_hhs, Hs, parts=0):
    from ._lib import LstmBwdDesc
    n = len(douts)
    douts = [d.contiguous() for d in douts]
    B, S, _ = douts[0].shape
    dgps = [torch.empty(B, S, 8 * H, device=douts[0].device, dtype=torch.float32) for H in Hs]
    descs = (LstmBwdDesc * n)()
    for i in range(n):
        descs[i] = LstmBwdDesc(douts[i].data_ptr(), gates[i].data_ptr(), cells[i].data_ptr(), w_hhs[i].data_ptr(),
                               dgps[i].data_ptr(), Hs[i], 0)
    if parts:
        _chk(lib.styler_lstm_bidir_bwd_multi_mfma(descs, n, B, S, parts, _stream()), "styler_lstm_bidir_bwd_multi_mfma")
    else:
        _chk(lib.styler_lstm_bidir_bwd_multi(descs, n, B, S, _stream()), "styler_lstm_bidir_bwd_multi")
    return dgps


def aug_classifier_tail(h, ln_g, ln_b, w2, b2):
    B, S, _ = h.shape
    assert h.is_contiguous()
    out = torch.empty(B, 2, device=h.device, dtype=torch.float32)
    _chk(lib.styler_aug_classifier_tail(h.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), w2.data_ptr(),
                                        b2.data_ptr(), out.data_ptr(), B, S, _stream()),
         "styler_aug_classifier_tail")
    return out


def duration_scan(B, S, device, dur=None, log_d=None, d_control=1.0, want_dur=False):
    """-> (csum int32 [B,S], mel_len int64 [B], dur_out fp32 [B,S] | None)"""
    csum = torch.empty(B, S, device=device, dtype=torch.int32)
    mel_len = torch.empty(B, device=device, dtype=torch.int64)
    dur_out = torch.empty(B, S, device=device, dtype=torch.float32) if (want_dur and log_d is not None) else None
    is_float = 0
    if dur is not None:
        assert dur.is_contiguous() and dur.dtype in (torch.int64, torch.float32)
        is_float = int(dur.dtype == torch.float32)
    if log_d is not None:
        assert log_d.is_contiguous()
    _chk(lib.styler_duration_scan(_ptr(dur), is_float, _ptr(log_d), float(d_control), _ptr(dur_out),
                                  csum.data_ptr(), mel_len.data_ptr(), B, S, _stream()), "styler_duration_scan")
    return csum, mel_len, dur_out


def length_regulate(x, csum, T, frame_idx=None):
    B, S, C = x.shape
    out = torch.empty(B, T, C, device=x.device, dtype=torch.float32)
    _chk(lib.styler_length_regulate(x.data_ptr(), _ld(x), csum.data_ptr(), out.data_ptr(), _ld(out),
                                    _ptr(frame_idx), B, S, T, C, _stream()), "styler_length_regulate")
    return out


def bucket_embed_add(text, speaker, p, p_scale, e, e_scale, pitch_bins, energy_bins, pitch_emb, energy_emb,
                     noise=None, p_ids=None, e_ids=None):
    B, T, _ = text.shape
    out = torch.empty(B, T, 256, device=text.device, dtype=torch.float32)
    out2 = torch.empty_like(out) if noise is not None else None
    assert p.is_contiguous() and e.is_contiguous()
    _chk(lib.styler_bucket_embed_add(text.data_ptr(), _ld(text), speaker.data_ptr(), _ld(speaker), p.data_ptr(),
                                     float(p_scale), e.data_ptr(), float(e_scale), pitch_bins.data_ptr(),
                                     energy_bins.data_ptr(), pitch_emb.data_ptr(), energy_emb.data_ptr(),
                                     out.data_ptr(), _ptr(noise), _ld(noise) if noise is not None else 0,
                                     _ptr(out2), _ptr(p_ids), _ptr(e_ids), B, T, _stream()),
         "styler_bucket_embed_add")
    return out, out2


def add2(a, b, out=None):
    """out = a + b over [.., C] views (channel slices allowed)."""
    C = a.shape[-1]
    rows = a.numel() // C
    if out is None:
        out = torch.empty(a.shape, device=a.device, dtype=torch.float32)
    _chk(lib.styler_add2(a.data_ptr(), _ld(a), _ptr(b), _ld(b) if b is not None else 0, out.data_ptr(), _ld(out),
                         rows, C, _stream()), "styler_add2")
    return out


def style_cat(te, pu, tnu, spk, eu, ru, du):
    """-> (enc [B, S, 1280], dp [B, S, 256]): styler_style_cat (all operands contiguous fp32)."""
    B, S, _ = te.shape
    enc = torch.empty(B, S, 1280, device=te.device, dtype=torch.float32)
    dp = torch.empty(B, S, 256, device=te.device, dtype=torch.float32)
    _chk(lib.styler_style_cat(te.data_ptr(), pu.data_ptr(), tnu.data_ptr(), spk.data_ptr(), eu.data_ptr(), ru.data_ptr(), du.data_ptr(),
                              enc.data_ptr(), dp.data_ptr(), B, S, _stream()), "styler_style_cat")
    return enc, dp


def add3(a, b, c):
    """(a + b) + c over [.., C] fp32 views with contiguous channels."""
    a, b, c = _rows_view(a), _rows_view(b), _rows_view(c)
    C = a.shape[-1]
    out = torch.empty(a.shape, device=a.device, dtype=torch.float32)
    _chk(lib.styler_add3(a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), c.data_ptr(), _ld(c), out.data_ptr(), C, a.numel() // C, C, _stream()),
         "styler_add3")
    return out


def fill_zero(t):
    """Zeros into a [.., C] view with contiguous channels (a channel / item slice of a wider buffer; fp32, or bf16 with an even
    channel count, offset and row stride): styler_copy_rows_multi with a null source -- no torch fill on the step."""
    if t.dtype == torch.bfloat16:
        t = t.view(torch.float32)
    copy_rows_multi([(None, t)])


def copy_rows_multi(pairs):
    """[(src or None, dst), ...] (<= 8 per launch): dst[..] = src[..] (zeros for None) over [.., C] fp32 views with contiguous
    channels -- ONE launch for all parts of a concatenation / gathered slice gradient (styler_copy_rows_multi)."""
    import ctypes
    from ._lib import CopySeg
    for i in range(0, len(pairs), 8):
        part = pairs[i:i + 8]
        arr = (CopySeg * len(part))()
        for k, (src, dst) in enumerate(part):
            C = dst.shape[-1]
            assert dst.dtype == torch.float32 and (src is None or (src.dtype == torch.float32 and src.shape == dst.shape))
            arr[k].src = src.data_ptr() if src is not None else None
            arr[k].dst = dst.data_ptr()
            arr[k].ld_src = (_ld(src) if src.dim() > 1 else C) if src is not None else 0
            arr[k].ld_dst = _ld(dst) if dst.dim() > 1 else C
            arr[k].rows = dst.numel() // C
            arr[k].C = C
        _chk(lib.styler_copy_rows_multi(arr, len(part), _stream()), "styler_copy_rows_multi")


def add_rowvec(a, v, L, out=None):
    """out[b,t,:] = (a[b,t,:] if a is not None else 0) + v[b,:]"""
    B, C = v.shape
    if out is None:
        out = torch.empty(B, L, C, device=v.device, dtype=torch.float32)
    _chk(lib.styler_add_rowvec(_ptr(a), _ld(a) if a is not None else 0, v.data_ptr(), v.stride(0), out.data_ptr(),
                               _ld(out), B, L, C, _stream()), "styler_add_rowvec")
    return out


def length_mask2(lens0, L0, lens1, L1):
    """(mask [B0, L0], mask [B1, L1]) bool, True = padding: the two masks of a forward in one launch."""
    m0 = torch.empty(lens0.shape[0], L0, device=lens0.device, dtype=torch.bool)
    m1 = torch.empty(lens1.shape[0], L1, device=lens1.device, dtype=torch.bool)
    _chk(lib.styler_length_mask2(lens0.data_ptr(), m0.data_ptr(), lens0.shape[0], L0, lens1.data_ptr(), m1.data_ptr(), lens1.shape[0], L1,
                                 _stream()), "styler_length_mask2")
    return m0, m1


def length_mask(lens, L):
    B = lens.shape[0]
    mask = torch.empty(B, L, device=lens.device, dtype=torch.bool)
    _chk(lib.styler_length_mask(lens.data_ptr(), mask.data_ptr(), B, L, _stream()), "styler_length_mask")
    return mask


def masked_err_sum(a, b, acc, kind, lens):
    """acc (double[2], zeroed by caller) += (sum err over valid, count).  a, b: [B, L] or [B, L, C]."""
    if a.dim() == 2:
        B, L = a.shape
        C, lda, ldb = 1, 1, 1
    else:
        B, L, C = a.shape
        lda, ldb = _ld(a), _ld(b)
    _chk(lib.styler_masked_err_sum(a.data_ptr(), lda, b.data_ptr(), ldb, acc.data_ptr(), kind, B, L, C,
                                   _ptr(lens), _stream()), "styler_masked_err_sum")
    return acc


# ======================================================================================
# backward / training wrappers
# ======================================================================================
def _rows_view(t):
    """Make a [.., C] gradient usable by the kernels: contiguous channels and uniform row stride."""
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        return t.contiguous()
    if t.dim() == 3 and t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1):
        return t.contiguous()
    if t.dim() == 3 and (t.stride(1) & 3 or t.data_ptr() & 15):
        return t.contiguous()
    return t


def act_bwd(dy, y, act, lens=None):
    dy = _rows_view(dy)
    B, L, C = dy.shape
    dz = torch.empty(B, L, C, device=dy.device, dtype=torch.float32)
    _chk(lib.styler_act_bwd(dy.data_ptr(), _ld(dy), _ptr(y), _ld(y) if y is not None else 0, dz.data_ptr(), C, B, L, C,
                            act, _ptr(lens), _stream()), "styler_act_bwd")
    return dz


def act_bwd_multi(items):
    """[(dy, y, act), ...] -> [dz, ...]: the activation backward of several tensors in one launch (<= 8 per launch)."""
    from ._lib import ActSeg
    outs = []
    for j in range(0, len(items), 8):
        part = items[j:j + 8]
        arr = (ActSeg * len(part))()
        for k, (dy, y, act) in enumerate(part):
            dy = _rows_view(dy)
            C = dy.shape[-1]
            dz = torch.empty(dy.shape, device=dy.device, dtype=torch.float32)
            outs.append(dz)
            m = arr[k]
            m.dy, m.y, m.dz = _f32(dy).data_ptr(), _f32(y).data_ptr(), dz.data_ptr()
            m.lddy, m.ldy, m.rows, m.C, m.act = _ld(dy), _ld(y), dy.numel() // C, C, act
        _chk(lib.styler_act_bwd_multi(arr, len(part), _stream()), "styler_act_bwd_multi")
    return outs


LIN128_SPLITS = int(os.environ.get("STYLER_WGRAD_LIN128_SPLITS", "8"))


def split3_parts(t3, C):
    """(hi, lo) bf16 views [.., C] of a split3 tensor ([.., 2C] = [hi | lo] or [.., 3C] = [hi | lo | hi])."""
    return t3[..., 0:C], t3[..., C:2 * C]


IO_X_LO, IO_DZ_LO, IO_X3CAT, IO_DB_SLOTS = 32, 64, 128, 256      # STYLER_IO_X_LO / _DZ_LO / _X3CAT / _DB_SLOTS
# inside a training step the bias gradients leave the weight-gradient kernels as per-split slots that the step's multi-tensor
# reduce folds in split order (bit-reproducible; STYLER_BIAS_SLOTS=0: fp32 atomics as before round 4)
bias_slots = os.environ.get("STYLER_BIAS_SLOTS", "1") != "0"


def _param_slots(nslots, length, target):
    """Inside a training step: [nslots][length] floats from the step's arena, their in-order fold into `target` queued on the
    step's multi-tensor reduce; None outside a step, with STYLER_BIAS_SLOTS=0, or while the arena is being sized (the caller
    then takes its atomics path)."""
    arena = wgrad_arena
    if arena is None or not bias_slots:
        return None
    sl = arena.take(nslots * length, target.device)
    if sl is None:
        return None
    arena.descs.append((sl.data_ptr(), target.data_ptr(), 1, 0, 0, length, 1, 1, nslots))
    return sl


def _param_slots_all(reqs):
    """All-or-nothing form for kernels that need SEVERAL slot arrays: `reqs` = [(nslots, length, target), ...] -> list of
    slot tensors, or None (atomics for all).  Every request is made even after one has failed, so that ONE measuring pass
    sizes the arena for all of them (the short-circuit under-counted `arena.total`, the arena then needed one
    pass per array to converge and a capture could bake the atomic path in)."""
    arena = wgrad_arena
    if arena is None or not bias_slots:
        return None
    n0 = len(arena.descs)
    out = [_param_slots(ns, ln, t) for ns, ln, t in reqs]
    if any(o is None for o in out):
        del arena.descs[n0:]                         # drop the folds queued for the arrays that did fit
        return None
    return out


def _bias_slots(arena, splits, n, db, db2, device):
    """Takes [splits][n] floats from the arena and queues their fold into db (and db2); None when the arena has no room (the
    measuring pass of the first step)."""
    bs = arena.take(splits * n, device)
    if bs is None:
        return None
    arena.descs.append((bs.data_ptr(), db.data_ptr(), 1, 0, 0, n, 1, 1, splits))
    if db2 is not None:
        arena.descs.append((bs.data_ptr(), db2.data_ptr(), 1, 0, 0, n, 1, 1, splits))
    return bs
# bf16x3 weight gradients on bf16-resident splits as one launch over the three parts (STYLER_WGRAD_X3CAT=0: three launches)
x3cat = os.environ.get("STYLER_WGRAD_X3CAT", "1") != "0"


def _is_split3(hi, lo, C):
    """(hi, lo) are the views split3_parts makes of ONE split tensor (lo right behind hi in every row)."""
    return (hi.dtype == torch.bfloat16 and lo.dtype == torch.bfloat16 and hi.shape == lo.shape and hi.stride() == lo.stride()
            and hi.stride(-1) == 1 and hi.stride(-2) >= 2 * C and lo.data_ptr() - hi.data_ptr() == 2 * C)


def wgrad(dz, x, dw, n, cin, kw=1, db=None, pad_left=None, strides=None, prec=None, db2=None, plan=None, dz_parts=None,
          x_parts=None, x_exact=False, parts=0):
    """dw (fp32, parameter layout [n, cin] or [n, cin, kw]) += dz^T x over all taps; db (and db2) += colsum(dz)."""
    B, L = dz.shape[0], dz.shape[1]
    if strides is None:
        strides = (cin * kw, kw, 1) if kw > 1 else (cin, 1, 0)
    if pad_left is None:
        pad_left = kw // 2
    if prec is None:
        from .runtime import rt
        prec = rt.prec
    if prec == PREC_BF16X3:
        # dz^T x = dz_hi^T x_hi + dz_hi^T x_lo + dz_lo^T x_hi: three calls of the bf16 engine accumulating into dw.
        if n % 4:
            prec = PREC_F32
        else:
            kwargs = dict(kw=kw, pad_left=pad_left, strides=strides, prec=PREC_BF16, plan=plan)
            tiles = ((n + 63) // 64) * ((cin + 63) // 64)
            resident = (n % 8 == 0 and cin % 8 == 0 and not x_exact and
                        ((kw == 1 and pad_left == 0 and tiles >= 16 and n > 64 and cin > 64) or kw in (5, 9)))
            if resident:
                # both operands as bf16 views of their [hi | lo (| hi)] splits (ops.split3) (the dX GEMM of the same node needs dz's split
                # anyway: callers hand it over as `dz_parts`): the LDS-DMA kernels take them.  The bias sums need both
                # parts of dz: colsum(dz_hi) from the first call, colsum(dz_lo) from the third.
                dzh, dzl = dz_parts if dz_parts is not None else split3_parts(split3(dz, plan), n)
                xh, xl = x_parts if x_parts is not None else split3_parts(split3(x, plan), cin)
                if (x3cat and db2 is None and lib.styler_wgrad_x3cat_ok(n, cin, kw, pad_left) and _is_split3(dzh, dzl, n)
                        and _is_split3(xh, xl, cin)):
                    # ONE launch over the three parts (STYLER_IO_X3CAT): a third of the partial tiles and of the launches
                    wgrad(dzh, xh, dw, n, cin, db=db, parts=IO_X3CAT, **kwargs)
                    return
                wgrad(dzh, xh, dw, n, cin, db=db, db2=db2, **kwargs)
                wgrad(dzh, xl, dw, n, cin, **kwargs)
                wgrad(dzl, xh, dw, n, cin, db=db, db2=db2, **kwargs)
                return
            # any other shape: fp32-typed operands (the kernels round an fp32 operand to bf16 while staging = its high part,
            # or, with STYLER_IO_X_LO / STYLER_IO_DZ_LO, stage its low part).  The bias sums come from the first call, which
            # adds the fp32 values of dz before rounding them.
            if dz.dtype != torch.float32 or x.dtype != torch.float32:
                raise StylerHipError("bf16x3 weight gradient: fp32 operands expected")
            wgrad(dz, x, dw, n, cin, db=db, db2=db2, **kwargs)
            if not x_exact:                           # (a one-hot x is exact in bf16: its low part is zero)
                wgrad(dz, x, dw, n, cin, parts=IO_X_LO, **kwargs)
            wgrad(dz, x, dw, n, cin, parts=IO_DZ_LO, **kwargs)
            return
    prof = gemm_profiler
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    arena = wgrad_arena
    io = (2 if dz.dtype == torch.bfloat16 else 0) | (1 if x.dtype == torch.bfloat16 else 0)    # bf16-resident operands
    if parts:
        assert prec == PREC_BF16 and io == (3 if parts == IO_X3CAT else 0), "low-part flags: fp32-typed operands; x3cat: bf16 splits"
    if (arena is not None and arena.buf is not None and prec == PREC_BF16 and prof is None and (plan is None or db2 is None)
            and (arena.group_all or (kw == 1 and io in (0, 2, 3)))):
        # grouped path (default: the small Linear gradients, variant 0; STYLER_WGRAD_GROUP_ALL=1: every bf16 gradient):
        # plan the member first (variant, tiles, split count), then give it its slice of the arena
        from ._lib import WgradGroupDesc
        import ctypes
        d = WgradGroupDesc()
        args = (dz.data_ptr(), _ld(dz), x.data_ptr(), _ld(x), _ptr(db), _ptr(db2), B, L, n, cin, kw, pad_left, prec)
        packed = (plan.counts.data_ptr(), plan.chunktab.data_ptr()) if plan is not None else (None, None)
        nb = lib.styler_wgrad_group_desc(ctypes.byref(d), *args, None, *packed, io | parts, 0)
        if nb < 0:
            _chk(nb, "styler_wgrad_group_desc")
        small = ((n + 63) // 64) * ((cin + 63) // 64) < 48
        if nb > 0 and (arena.group_all or d.variant == 0 or (d.variant in (1, 7, 8) and small)):
            want = arena.want_splits(d.variant) if arena.group_all else (LIN128_SPLITS if d.variant in (1, 7, 8) else 0)
            if want:
                nb = lib.styler_wgrad_group_desc(ctypes.byref(d), *args, None, *packed, io | parts, want)
            ws = arena.take(d.splits * n * kw * cin, dz.device)
            if ws is not None:
                d.ws = ws.data_ptr()
                arena.descs.append((ws.data_ptr(), dw.data_ptr(), strides[0], strides[1], strides[2], n, cin, kw, d.splits))
                if db is not None and bias_slots:
                    bs = _bias_slots(arena, d.splits, n, db, db2, dz.device)
                    if bs is not None:               # the member stores its column sums into the slots (0x800: see gemm_bwd.hip)
                        d.db, d.db2, d.pad_left = bs.data_ptr(), 0, d.pad_left | 0x800
                arena.group.append(d)
                arena.group_keep.append((dz, x, plan))
                return
            arena.total -= (d.splits * n * kw * cin + 3) & ~3      # did not fit: the stand-alone request below is counted
    nfloats = int(lib.styler_wgrad_workspace_bytes_io(B, L, n, cin, kw, pad_left, prec, io | parts)) // 4
    ws, defer = None, 0
    if arena is not None:
        ws = arena.take(nfloats, dz.device)
        if ws is not None:
            defer = 1
            arena.descs.append((ws.data_ptr(), dw.data_ptr(), strides[0], strides[1], strides[2], n, cin, kw,
                                int(lib.styler_wgrad_splits_io(B, L, n, cin, kw, pad_left, prec, io | parts))))
    if arena is not None and db is not None and bias_slots:
        if defer:
            bs = _bias_slots(arena, arena.descs[-1][8], n, db, db2, dz.device)
            if bs is not None:
                db, db2, parts = bs, None, parts | IO_DB_SLOTS
        else:                                        # measuring pass: the slots count towards the next step's arena
            arena.total += (int(lib.styler_wgrad_splits_io(B, L, n, cin, kw, pad_left, prec, io | parts)) * n + 3) & ~3
    if ws is None:
        ws = torch.empty(nfloats, device=dz.device, dtype=torch.float32)
    grouped = False
    from .runtime import rt as _rt
    side = None
    if (not grouped) and defer and _rt.wgrad_stream and prof is None and prec == PREC_BF16:
        side = arena.side_stream(dz.device)
        arena.side_keep.append((dz, x, plan))
    if side is not None:
        with torch.cuda.stream(side):
            _wgrad_launch(dz, x, dw, db, db2, strides, B, L, n, cin, kw, pad_left, prec, ws, defer, io | parts, plan)
    elif not grouped:
        _wgrad_launch(dz, x, dw, db, db2, strides, B, L, n, cin, kw, pad_left, prec, ws, defer, io | parts, plan)
    if prof is not None:
        e1.record()
        prof.records.append(("wgrad_bf16" if prec == PREC_BF16 else "wgrad", 2.0 * B * L * n * kw * cin, e0, e1,
                             plan is not None))


def _wgrad_launch(dz, x, dw, db, db2, strides, B, L, n, cin, kw, pad_left, prec, ws, defer, io, plan):
    if plan is not None:
        assert B == 1 and L == plan.rows and db2 is None and pad_left == kw // 2
        _chk(lib.styler_wgrad_packed(dz.data_ptr(), _ld(dz), x.data_ptr(), _ld(x), dw.data_ptr(), _ptr(db), strides[0],
                                     strides[1], strides[2], L, n, cin, kw, prec, ws.data_ptr(), defer,
                                     plan.rowinfo.data_ptr(), plan.chunktab.data_ptr(), plan.counts.data_ptr(), io,
                                     _stream()), "styler_wgrad_packed")
    else:
        _chk(lib.styler_wgrad(dz.data_ptr(), _ld(dz), x.data_ptr(), _ld(x), dw.data_ptr(), _ptr(db), _ptr(db2), strides[0], strides[1],
                              strides[2], B, L, n, cin, kw, pad_left, prec, ws.data_ptr(), defer, io, _stream()),
             "styler_wgrad")


def colsum(dz, out, out2=None):
    C = dz.shape[-1]
    rows = dz.numel() // C
    _chk(lib.styler_colsum(dz.data_ptr(), _ld(dz), out.data_ptr(), _ptr(out2), rows, C, _stream()), "styler_colsum")


def attention_bwd(qkv, out, dout, lse, lens, prec=None, plan=None, out_bf16=False, x3=False):
    """out_bf16 (throughput mode only): dqkv comes back as bf16 -- what the QKV dX GEMM and the weight gradients round it to."""
    B, L = (plan.B, plan.T) if plan is not None else (qkv.shape[0], qkv.shape[1])
    dout = dout.contiguous()
    bf16 = _prec(prec) == PREC_BF16
    dqkv = torch.empty_like(qkv, dtype=torch.bfloat16 if (out_bf16 and bf16) else torch.float32)
    ws = torch.empty(B * 4 * L, device=qkv.device, dtype=torch.float32)
    if plan is not None:
        lens, cu = plan.lens, plan.cu.data_ptr()
    else:
        cu = None
    if bf16:
        _chk(lib.styler_attention_bwd_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                           ws.data_ptr(), B, L, _ptr(lens), cu,
                                           (2 if dqkv.dtype == torch.bfloat16 else 0) | (1 if qkv.dtype == torch.bfloat16 else 0) |
                                           (4 if out.dtype == torch.bfloat16 else 0) | (8 if dout.dtype == torch.bfloat16 else 0),
                                           _stream()), "styler_attention_bwd_bf16")
    else:
        fn = lib.styler_attention_bwd_x3 if (_prec(prec) == PREC_BF16X3 and rt_attn_x3) else lib.styler_attention_bwd
        y3 = _x3_begin(dqkv, x3 and fn is lib.styler_attention_bwd_x3, allowed_parts=(2,))   # (dqkv feeds the QKV dX GEMM + weight gradients)
        try:
            _chk(fn(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                    ws.data_ptr(), B, L, _ptr(lens), cu, _stream()), "styler_attention_bwd")
        except Exception:
            _x3_abort(y3)
            raise
        _x3_end(dqkv, y3, plan)
    return dqkv


def layernorm_bwd(x, dy, gamma, beta, dgamma, dbeta, lens=None, need_dx=True, dot_w=None, dout=None, ddot_w=None,
                  ddot_b=None, drop_p=0.0, drop_seed=0, in_drop_p=0.0, in_drop_seed=0, relu_input=False, x3=False, plan=None):
    """Returns dx, or (dx, dx_drop) when in_drop_p > 0 (dx_drop = dx through the forward's input-dropout mask).
    relu_input: x is a ReLU output and dx comes back as the gradient w.r.t. the ReLU's input."""
    B, L, C = x.shape
    # bf16 residual stream (x = the saved pre-norm sum is bf16): the gradients that continue along it are bf16 as well
    gdt = torch.bfloat16 if x.dtype == torch.bfloat16 else torch.float32
    dx = torch.empty(B, L, C, device=x.device, dtype=gdt) if need_dx else None
    dxd = torch.empty(B, L, C, device=x.device, dtype=gdt) if in_drop_p > 0 else None
    if dy is not None:
        dy = _rows_view(dy)
    lnb_io = (2 if x.dtype == torch.bfloat16 else 0) | (4 if dy is not None and dy.dtype == torch.bfloat16 else 0) | \
             (8 if dx is not None and dx.dtype == torch.bfloat16 else 0) | (16 if dxd is not None and dxd.dtype == torch.bfloat16 else 0)
    if dout is not None:
        dout = dout.contiguous()
    # inside a training step the three parameter-gradient vectors go to zeroed 16-replica scratch (zero slab) and are
    # folded by the backward's single multi-tensor reduce launch (WgradArena)
    rep, pg, pb, pw = 1, dgamma, dbeta, ddot_w
    arena, slab = wgrad_arena, zero_slab
    if slab is not None:
        nvec = 3 if ddot_w is not None else 2
        sc = slab.take(nvec * LN_REPLICAS * 128)     # doubles = nvec * 16 * 256 floats (asked for in every pass: sizing)
        want_db = ddot_w is not None and ddot_b is not None and bias_slots
        sdb = slab.take(LN_REPLICAS // 2) if want_db else None           # (likewise)
        if sc is not None and arena is not None and arena.buf is not None:
            sc = sc.view(torch.float32).view(nvec, LN_REPLICAS, 256)
            rep, pg, pb = LN_REPLICAS, sc[0], sc[1]
            arena.descs.append((pg.data_ptr(), dgamma.data_ptr(), 1, 0, 0, 256, 1, 1, rep))
            arena.descs.append((pb.data_ptr(), dbeta.data_ptr(), 1, 0, 0, 256, 1, 1, rep))
            if ddot_w is not None:
                pw = sc[2]
                arena.descs.append((pw.data_ptr(), ddot_w.data_ptr(), 1, 0, 0, 256, 1, 1, rep))
                if sdb is not None:                  # the tail's scalar bias: one slot per block as well
                    sdb = sdb.view(torch.float32)
                    arena.descs.append((sdb.data_ptr(), ddot_b.data_ptr(), 1, 0, 0, 1, 1, 1, rep))
                    ddot_b, lnb_io = sdb, lnb_io | 64        # STYLER_LNB_DOTB_SLOTS
    fold = None
    if rep == 1:
        # stand-alone call (no training step around it): the blocks still store into per-block slots -- never fp32 atomics
        # into one vector, whose order (and therefore the last bits of the sum) changes from launch to launch -- and a
        # fixed-order fold adds them to the gradients
        nvec = 3 if ddot_w is not None else 2
        fold = torch.zeros(nvec, LN_REPLICAS, 256, device=x.device, dtype=torch.float32)
        rep, pg, pb = LN_REPLICAS, fold[0], fold[1]
        fold_db = None
        if ddot_w is not None:
            pw = fold[2]
            if ddot_b is not None:                       # the tail's scalar bias: per-block slots too (round-3 advisor)
                fold_db, fold_db_dst = torch.zeros(LN_REPLICAS, device=x.device, dtype=torch.float32), ddot_b
                ddot_b, lnb_io = fold_db, lnb_io | 64    # STYLER_LNB_DOTB_SLOTS
    # x3: the gradient that continues into the sublayer's GEMMs (dx_drop when the forward dropped its input, else dx) leaves
    # with its bf16x3 split, filed under the key split3(., plan) looks up
    d_o = dxd if dxd is not None else dx
    y3 = _x3_begin(d_o, x3, allowed_parts=(2,))
    try:
        _chk(lib.styler_layernorm_bwd(x.data_ptr(), _ld(x), _ptr(dy), _ld(dy) if dy is not None else 0, gamma.data_ptr(),
                                      _ptr(beta), _ptr(dx), C, pg.data_ptr(), pb.data_ptr(), _ptr(dot_w),
                                      _ptr(dout), _ptr(pw), _ptr(ddot_b), B, L, C, _ptr(lens), float(drop_p),
                                      int(drop_seed), float(in_drop_p), int(in_drop_seed), _ptr(dxd), C, rep, (1 if relu_input else 0) | lnb_io,
                                      _stream()),
             "styler_layernorm_bwd")
    except Exception:
        _x3_abort(y3)
        raise
    _x3_end(d_o, y3, plan)
    if fold is not None:
        _chk(lib.styler_fold_replicas(fold[0].data_ptr(), fold[1].data_ptr(), _ptr(fold[2]) if ddot_w is not None else None,
                                      dgamma.data_ptr(), dbeta.data_ptr(), _ptr(ddot_w), LN_REPLICAS, 256, _stream()),
             "styler_fold_replicas")
        if fold_db is not None:
            _chk(lib.styler_fold_replicas(fold_db.data_ptr(), None, None, fold_db_dst.data_ptr(), None, None, LN_REPLICAS, 1,
                                          _stream()), "styler_fold_replicas")
    return (dx, dxd) if dxd is not None else dx


def groupnorm_relu_bwd(x, dy, gamma, beta, stats, dgamma, dbeta, dx_bf16=False, x3=False):
    B, L, C = x.shape
    dy = _rows_view(dy)
    dx = torch.empty(B, L, C, device=x.device, dtype=torch.bfloat16 if dx_bf16 else torch.float32)
    ws, z = _norm_ws(B * (C // 16) * 2, x.device)
    # gamma / beta gradients: per-item slots folded in item order by the step's reduce (single-pass kernel; else atomics)
    sg = sb = None
    if L <= lib.styler_groupnorm_fused_rows(1):
        both = _param_slots_all([(B, C, dgamma), (B, C, dbeta)])     # (no room for both: atomics for both)
        if both is not None:
            sg, sb = both
    pg, pb, slots = (sg, sb, IO_PARAM_SLOTS) if sb is not None else (dgamma, dbeta, 0)
    y3 = _x3_begin(dx, x3)
    _chk(lib.styler_groupnorm_relu_bwd(x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), gamma.data_ptr(), beta.data_ptr(),
                                       stats.data_ptr(), dx.data_ptr(), C, pg.data_ptr(), pb.data_ptr(),
                                       ws.data_ptr(), z, B, L, C,
                                       (2 if dx_bf16 else 0) | (1 if dy.dtype == torch.bfloat16 else 0) |
                                       (IO_Z_BF16 if x.dtype == torch.bfloat16 else 0) | slots, _stream()),
         "styler_groupnorm_relu_bwd")
    _x3_end(dx, y3)
    return dx


def batchnorm_bwd(x, y, dy, gamma, mean, rstd, dgamma, dbeta, act, beta=None, drop_p=0.0, drop_seed=0, segs=1,
                  dx_bf16=False, x3=False):
    """`y` may be None when `beta` is given (the activation output is recomputed from x)."""
    C = x.shape[-1]
    rows = x.numel() // C
    dy = dy.contiguous()
    dx = torch.empty_like(x, dtype=torch.bfloat16 if dx_bf16 else torch.float32)
    ws, z = _bn_ws(rows, C, segs, x.device)
    y3 = _x3_begin(dx, x3)
    _chk(lib.styler_batchnorm_bwd(x.data_ptr(), _ptr(y), dy.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                  rstd.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), z,
                                  rows, C, act, _ptr(beta), float(drop_p), int(drop_seed), int(segs),
                                  (2 if dx_bf16 else 0) | (1 if dy.dtype == torch.bfloat16 else 0) |
                                  (IO_Z_BF16 if x.dtype == torch.bfloat16 else 0), _stream()),
         "styler_batchnorm_bwd")
    _x3_end(dx, y3)
    return dx


def embed_bwd(text, dy, demb):
    """demb [V, C] += the dy rows of every token, per table row in token order (no atomics: bit-reproducible)."""
    dy = _rows_view(dy)
    B, L = text.shape
    if demb.shape[1] <= 1024 and bias_slots:
        _chk(lib.styler_embed_bwd_det(text.data_ptr(), dy.data_ptr(), _ld(dy), demb.data_ptr(), B, L, demb.shape[1],
                                      demb.shape[0], _stream()), "styler_embed_bwd_det")
        return
    _chk(lib.styler_embed_bwd(text.data_ptr(), dy.data_ptr(), _ld(dy), demb.data_ptr(), B, L, demb.shape[1], _stream()),
         "styler_embed_bwd")


def onehot_conv5_bwd(v, dy, dw, db):
    """dw [C, 257, 5] += onehot(v)^T-conv dy, db += colsum(dy): one-hot expansion + the wgrad GEMM."""
    dy = _rows_view(dy)
    B, L = v.shape
    C = db.numel()
    oh = torch.empty(B, L, 260, device=v.device, dtype=torch.float32)
    _chk(lib.styler_onehot_expand(v.data_ptr(), oh.data_ptr(), B * L, _stream()), "styler_onehot_expand")
    wgrad(dy, oh, dw, C, 257, kw=5, db=db, strides=(257 * 5, 5, 1), x_exact=True)


def mel_calibrate_bwd(dy, mel_len, src_len, T, out_bf16=False):
    dy = _rows_view(dy)
    B, S, C = dy.shape
    dx = torch.empty(B, T, C, device=dy.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    _chk(lib.styler_mel_calibrate_bwd_io(dy.data_ptr(), _ld(dy), dx.data_ptr(), C, mel_len.data_ptr(), src_len.data_ptr(),
                                         B, T, S, C, 2 if out_bf16 else 0, _stream()), "styler_mel_calibrate_bwd")
    return dx


def lstm_bidir_bwd(dout, gates, cell, w_hh, H):
    dout = dout.contiguous()
    B, S, _ = dout.shape
    dgp = torch.empty(B, S, 8 * H, device=dout.device, dtype=torch.float32)
    _chk(lib.styler_lstm_bidir_bwd(dout.data_ptr(), gates.data_ptr(), cell.data_ptr(), w_hh.data_ptr(), dgp.data_ptr(),
                                   B, S, H, _stream()), "styler_lstm_bidir_bwd")
    return dgp


def aug_classifier_tail_bwd(h, ln_g, ln_b, w2, b2, dout, dln_g, dln_b, dw2, db2):
    B, S, _ = h.shape
    dh = torch.empty_like(h)
    dout = dout.contiguous()
    ns = lib.styler_aug_classifier_tail_slots(B, S)
    tg = [(dln_g, 256), (dln_b, 256), (dw2, 512), (db2, 2)]
    sl = _param_slots_all([(ns, n, t) for t, n in tg])    # not all four fit: atomics for all
    if sl is not None:
        _chk(lib.styler_aug_classifier_tail_bwd_io(h.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                                   dout.data_ptr(), dh.data_ptr(), sl[0].data_ptr(), sl[1].data_ptr(),
                                                   sl[2].data_ptr(), sl[3].data_ptr(), B, S, IO_PARAM_SLOTS, _stream()),
             "styler_aug_classifier_tail_bwd_io")
        return dh
    _chk(lib.styler_aug_classifier_tail_bwd(h.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                            dout.data_ptr(), dh.data_ptr(), dln_g.data_ptr(), dln_b.data_ptr(),
                                            dw2.data_ptr(), db2.data_ptr(), B, S, _stream()),
         "styler_aug_classifier_tail_bwd")
    return dh


def length_regulate_bwd(dy, csum, S):
    dy = _rows_view(dy)
    B, T, C = dy.shape
    dx = torch.empty(B, S, C, device=dy.device, dtype=torch.float32)
    _chk(lib.styler_length_regulate_bwd(dy.data_ptr(), _ld(dy), csum.data_ptr(), dx.data_ptr(), C, B, S, T, C,
                                        _stream()), "styler_length_regulate_bwd")
    return dx


def bucket_embed_bwd(dy, p_ids, e_ids, dpitch_emb, denergy_emb):
    dy = dy.contiguous()
    B, T, _ = dy.shape
    ns = lib.styler_bucket_embed_slices()
    sp = se = None
    if dpitch_emb.numel() == 256 * 256 and denergy_emb.numel() == 256 * 256:
        both = _param_slots_all([(ns, dpitch_emb.numel(), dpitch_emb), (ns, denergy_emb.numel(), denergy_emb)])
        if both is not None:
            sp, se = both
    if se is not None:                               # per-slice slots, folded in slice order by the step's reduce
        _chk(lib.styler_bucket_embed_bwd_slots(dy.data_ptr(), p_ids.data_ptr(), e_ids.data_ptr(), sp.data_ptr(), se.data_ptr(),
                                               B, T, _stream()), "styler_bucket_embed_bwd_slots")
        return
    _chk(lib.styler_bucket_embed_bwd(dy.data_ptr(), p_ids.data_ptr(), e_ids.data_ptr(), dpitch_emb.data_ptr(),
                                     denergy_emb.data_ptr(), B, T, _stream()), "styler_bucket_embed_bwd")


def rowsum(x, out=None, accumulate=False):
    x = _rows_view(x)
    B, L, C = x.shape
    if out is None:
        out = torch.empty(B, C, device=x.device, dtype=torch.float32)
    _chk(lib.styler_rowsum(x.data_ptr(), _ld(x), out.data_ptr(), out.stride(0), B, L, C, int(accumulate), _stream()),
         "styler_rowsum")
    return out


MASKED_ACC_DOUBLES = 2052     # STYLER_MASKED_ACC_DOUBLES: totals, ticket and one slot pair per block (<= 1024 blocks per term)


def masked_err_bwd(a, b, acc, gscale, kind, lens):
    if a.dim() == 2:
        B, L = a.shape
        C, lda, ldb = 1, 1, 1
    else:
        B, L, C = a.shape
        lda, ldb = _ld(a), _ld(b)
    da = torch.empty(a.shape, device=a.device, dtype=torch.float32)
    _chk(lib.styler_masked_err_bwd(a.data_ptr(), lda, b.data_ptr(), ldb, acc.data_ptr(), gscale.data_ptr(),
                                   da.data_ptr(), kind, B, L, C, _ptr(lens), _stream()), "styler_masked_err_bwd")
    return da


def masked_err_mean(a, b, kind, lens):
    """-> (mean [1] fp32, acc [4] fp64): masked MSE (kind 0) / L1 (kind 1) over valid positions, mean taken in the kernel.
    Inside a training step the accumulator comes from the per-step zero slab (no memset of its own)."""
    if a.dim() == 2:
        B, L = a.shape
        C, lda, ldb = 1, 1, 1
    else:
        B, L, C = a.shape
        lda, ldb = _ld(a), _ld(b)
    acc = None
    if zero_slab is not None:
        acc = zero_slab.take(MASKED_ACC_DOUBLES)
    if acc is None:
        acc = torch.zeros(MASKED_ACC_DOUBLES, dtype=torch.float64, device=a.device)
    out = torch.empty(1, device=a.device, dtype=torch.float32)
    _chk(lib.styler_masked_err_mean(_f32(a).data_ptr(), lda, _f32(b).data_ptr(), ldb, acc.data_ptr(), out.data_ptr(), kind, B, L,
                                    C, _ptr(lens), _stream()), "styler_masked_err_mean")
    return out, acc


def _masked_dims(a, b):
    if a.dim() == 2:
        return a.shape[0], a.shape[1], 1, 1, 1
    return a.shape[0], a.shape[1], a.shape[2], _ld(a), _ld(b)


def masked_err_mean_multi(terms):
    """[(a, b, kind, lens), ...] (<= 8) -> ([mean [1] fp32, ...], [acc [4] fp64, ...]): the masked MSE / L1 terms of one loss
    call in ONE launch (styler_masked_err_mean_multi)."""
    from ._lib import MaskedTerm
    arr = (MaskedTerm * len(terms))()
    means = torch.empty(len(terms), device=terms[0][0].device, dtype=torch.float32)
    accs = []
    for k, (a, b, kind, lens) in enumerate(terms):
        B, L, C, lda, ldb = _masked_dims(a, b)
        acc = zero_slab.take(MASKED_ACC_DOUBLES) if zero_slab is not None else None
        if acc is None:
            acc = torch.zeros(MASKED_ACC_DOUBLES, dtype=torch.float64, device=a.device)
        accs.append(acc)
        m = arr[k]
        m.a, m.b, m.acc, m.mean, m.len = _f32(a).data_ptr(), _f32(b).data_ptr(), acc.data_ptr(), means[k:k + 1].data_ptr(), _ptr(lens)
        m.lda, m.ldb, m.B, m.L, m.C, m.kind = lda, ldb, B, L, C, kind
    _chk(lib.styler_masked_err_mean_multi(arr, len(terms), _stream()), "styler_masked_err_mean_multi")
    return [means[k:k + 1] for k in range(len(terms))], accs


def masked_err_bwd_multi(terms):
    """[(a, b, acc, gscale [1], kind, lens), ...] -> [da, ...] in one launch."""
    from ._lib import MaskedTerm
    arr = (MaskedTerm * len(terms))()
    outs = []
    for k, (a, b, acc, g, kind, lens) in enumerate(terms):
        B, L, C, lda, ldb = _masked_dims(a, b)
        da = torch.empty(a.shape, device=a.device, dtype=torch.float32)
        outs.append(da)
        m = arr[k]
        m.a, m.b, m.acc, m.gscale, m.da, m.len = a.data_ptr(), b.data_ptr(), acc.data_ptr(), g.data_ptr(), da.data_ptr(), _ptr(lens)
        m.lda, m.ldb, m.B, m.L, m.C, m.kind = lda, ldb, B, L, C, kind
    _chk(lib.styler_masked_err_bwd_multi(arr, len(terms), _stream()), "styler_masked_err_bwd_multi")
    return outs


def nll3(lps, label, gscale=None, want_grad=False):
    """Three NLLLoss(mean) terms summed.  `label`: int64 [B] tensor, or the python int 0 / 1 (all labels equal)."""
    B = lps[0].shape[0]
    const = label if isinstance(label, int) else 0
    lab = None if isinstance(label, int) else label
    loss = torch.empty(1, device=lps[0].device, dtype=torch.float32) if not want_grad else None
    d3 = torch.empty(3, B, 2, device=lps[0].device, dtype=torch.float32) if want_grad else None
    _chk(lib.styler_nll3(lps[0].data_ptr(), lps[1].data_ptr(), lps[2].data_ptr(), _ptr(lab), const, _ptr(loss),
                         _ptr(gscale), _ptr(d3), B, _stream()), "styler_nll3")
    return d3 if want_grad else loss


def weighted_sum(terms, weights):
    """sum_i weights[i] * terms[i] over scalar fp32 device tensors -> [1]."""
    import ctypes
    n = len(terms)
    ptrs = (ctypes.c_void_p * n)(*[_f32(t).data_ptr() for t in terms])
    w = (ctypes.c_float * n)(*[float(x) for x in weights])
    out = torch.empty(1, device=terms[0].device, dtype=torch.float32)
    _chk(lib.styler_weighted_sum(ptrs, w, n, out.data_ptr(), _stream()), "styler_weighted_sum")
    return out


def scale_weights(g, weights):
    import ctypes
    n = len(weights)
    w = (ctypes.c_float * n)(*[float(x) for x in weights])
    out = torch.empty(n, device=g.device, dtype=torch.float32)
    _chk(lib.styler_scale_weights(_f32(g).data_ptr(), w, n, out.data_ptr(), _stream()), "styler_scale_weights")
    return out


def _loss_tail_args(weights, lps, labels):
    import ctypes
    n = len(weights) - 2
    w = (ctypes.c_float * len(weights))(*[float(x) for x in weights])
    lp = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in lps])
    lab = [None if isinstance(l, int) else l for l in labels]
    const = [l if isinstance(l, int) else 0 for l in labels]
    return n, w, lp, lab, const


def loss_tail(means, weights, lps, labels):
    """The tail of the train step's loss head in one launch (styler_loss_tail): `means` = n <= 8 scalar fp32 device tensors,
    `weights` = n + 2 floats, `lps` = six [B, 2] log-probability tensors (main pass d, p, e, then DAT pass d, p, e), `labels` =
    two int64 [B] tensors or python ints -> out [3] = (total, nll3 of the main pass, nll3 of the DAT pass)."""
    import ctypes
    n, w, lp, lab, const = _loss_tail_args(weights, lps, labels)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[_f32(t).data_ptr() for t in means])
    B = lps[0].shape[0]
    out = torch.empty(3, device=lps[0].device, dtype=torch.float32)
    _chk(lib.styler_loss_tail(ptrs, w, n, lp, _ptr(lab[0]), const[0], _ptr(lab[1]), const[1], B, out.data_ptr(), _stream()),
         "styler_loss_tail")
    return out


def loss_tail_bwd(g, weights, lps, labels):
    """-> (gw [n] = g * weights[:n], d6 [6, B, 2]): the backward of loss_tail in one launch."""
    n, w, lp, lab, const = _loss_tail_args(weights, lps, labels)
    B = lps[0].shape[0]
    gw = torch.empty(max(n, 1), device=g.device, dtype=torch.float32)
    d6 = torch.empty(6, B, 2, device=g.device, dtype=torch.float32)
    _chk(lib.styler_loss_tail_bwd(_f32(g).data_ptr(), w, n, lp, _ptr(lab[0]), const[0], _ptr(lab[1]), const[1], B,
                                  gw.data_ptr(), d6.data_ptr(), _stream()), "styler_loss_tail_bwd")
    return gw, d6


def nll(logp, label, gscale=None, want_grad=False):
    B = logp.shape[0]
    loss = torch.empty(1, device=logp.device, dtype=torch.float32) if not want_grad else None
    dlogp = torch.empty_like(logp) if want_grad else None
    _chk(lib.styler_nll(logp.data_ptr(), label.data_ptr(), _ptr(loss), _ptr(gscale), _ptr(dlogp), B, _stream()),
         "styler_nll")
    return dlogp if want_grad else loss[0]


def dropout(x, p, seed):
    x = _rows_view(x)
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    _chk(lib.styler_dropout(x.data_ptr(), _ld(x), y.data_ptr(), C, rows, C, float(p), int(seed), _stream()),
         "styler_dropout")
    return y


def step_begin(grad, slab, counter):
    """One launch: zeros into `grad` and `slab` (either may be None), counter[0] += 1 (int64 [1] tensor or None)."""
    nbytes = lambda t: 0 if t is None else t.numel() * t.element_size()
    ok = all(t is None or (t.is_contiguous() and t.data_ptr() % 16 == 0 and nbytes(t) % 16 == 0) for t in (grad, slab))
    if not ok:                                       # (not the step's case)
        for t in (grad, slab):
            if t is not None:
                t.zero_()
        if counter is not None:
            counter.add_(1)
        return
    _chk(lib.styler_step_begin(_ptr(grad), nbytes(grad), _ptr(slab), nbytes(slab), _ptr(counter), _stream()), "styler_step_begin")


def sumsq(flat, out):
    _chk(lib.styler_sumsq(flat.data_ptr(), flat.numel(), out.data_ptr(), _stream()), "styler_sumsq")
    return out


def adam_step(p, g, m, v, sumsq_buf, max_norm, lr, beta1, beta2, eps, step, grad_scale=1.0):
    _chk(lib.styler_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), _ptr(sumsq_buf),
                              float(max_norm), float(lr), float(beta1), float(beta2), float(eps), int(step),
                              float(grad_scale), _stream()), "styler_adam_step")

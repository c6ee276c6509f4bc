#!/usr/bin/env python
"""Which source lines of the train step launch torch's own kernels (aten elementwise / fill / copy / cat)?  Every such
launch is plumbing that leaked onto the hot path; the table tells where to fuse it away.  One eager step under a
TorchDispatchMode (backward on the calling thread so that the engine's own accumulation ops are seen too)."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from closed_form import make_batch
import styler_amd
from styler_amd import rt
from styler_amd.training import TrainState, train_step

VIEWS = {"view", "_unsafe_view", "reshape", "slice", "select", "transpose", "expand", "unsqueeze", "squeeze", "detach", "alias",
         "as_strided", "empty", "empty_like", "empty_strided", "new_empty", "permute", "unbind", "split", "split_with_sizes",
         "_local_scalar_dense", "t", "is_same_size", "sym_size", "sym_stride", "sym_numel", "stride", "size", "numel",
         "lift_fresh", "record_stream", "unflatten", "flatten", "_reshape_alias", "narrow", "is_pinned", "set_", "resize_"}


def is_view(name):
    """aten.<op>.<overload> -> True for metadata-only ops (exact op names: 'cat' must not match 't')."""
    parts = name.split(".")
    return len(parts) >= 2 and parts[1] in VIEWS


class Tracer(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if is_view(name):
            return out
        flat = [a for a in torch.utils._pytree.tree_leaves((args, kwargs)) if torch.is_tensor(a)]
        if not any(a.is_cuda for a in flat):
            return out
        site = "?"
        for fr in reversed(traceback.extract_stack()):
            if ("styler_amd" in fr.filename or fr.filename.endswith("bench.py")) and "find_torch_ops" not in fr.filename:
                site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                break
        if os.environ.get("FIND_SHAPES"):          # which tensors: shapes / dtypes of the tensor arguments
            site += "  " + " ".join(f"{tuple(a.shape)}:{str(a.dtype)[6:]}" for a in flat[:3])
        self.counts[(name, site)] += 1
        return out


dev = torch.device("cuda")
torch.manual_seed(0)
m = styler_amd.STYLER().to(dev).train()
rt.set_precision("bf16")
rt.strict_inputs = False
st = TrainState(m)
from styler_amd.training import add_pair_inputs
b = add_pair_inputs({k: v.to(dev) for k, v in make_batch(48, 20, 60, 2, 13, seed=1234).items()})
for _ in range(3):
    train_step(m, st, b)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
tr = Tracer()
with tr:
    train_step(m, st, b)
torch.cuda.synchronize()
for (name, site), n in sorted(tr.counts.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n:4d}  {name:34s} {site}")
print("total:", sum(tr.counts.values()))

#!/usr/bin/env python
"""Where the derived-layout refresh (runtime.Derived.refresh_all, one strided_copy_multi launch after every optimiser step) spends
its time: the descriptors of the training model grouped by copy pattern (plain cast / tap interleave / tiled transpose), each
group timed as its own launch (HIP events, median of 20)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import styler_amd
from styler_amd import ops, rt, _lib
from styler_amd.runtime import Derived
from styler_amd.training import TrainState, add_pair_inputs, train_step
from closed_form import make_batch

dev = torch.device("cuda")
model = styler_amd.STYLER().to(dev).train()
rt.set_precision("bf16"); rt.strict_inputs = False
bd = add_pair_inputs({k: v.to(dev) for k, v in make_batch(48, 20, 60, 2, 13, seed=1234).items()})
state = TrainState(model)
for _ in range(2):
    train_step(model, state, bd)
torch.cuda.synchronize()
specs = [r() for r in Derived._registry if r() is not None]
groups = {"plain": [], "taps": [], "tiled": [], "all": []}
for sp in specs:
    descs = []
    sp.fill(descs, 0)
    for d in descs:
        groups["tiled" if d.flags & 2 else "taps" if d.flags & 4 else "plain"].append(d)
        groups["all"].append(d)


def nblocks(d):
    dims = (d.d0, d.d1, d.d2)
    if d.flags & 2:
        return ((dims[2] + 63) // 64) * ((dims[0] * dims[1] + 63) // 64)
    if d.flags & 4:
        return ((dims[0] + 1) // 2) * ((dims[2] + 127) // 128)
    return (dims[0] * dims[1] * dims[2] + 1023) // 1024


for name, ds in groups.items():
    if not ds:
        continue
    start = 0
    elems = 0
    for d in ds:
        d.block_start = start
        start += nblocks(d)
        elems += d.d0 * d.d1 * d.d2
    arr = (_lib.CopyDesc * len(ds))(*ds)
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    ts = []
    for _ in range(25):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops._chk(ops.lib.styler_strided_copy_multi(tab.data_ptr(), len(ds), start, ops._stream()), "copy")
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    print(f"{name:6s} {len(ds):4d} descriptors {start:7d} blocks {elems / 1e6:7.2f} M elements  {us:7.1f} us  "
          f"{elems * 6 / us / 1e6:6.2f} TB/s (4 B read + 2 B written per element)")

#!/usr/bin/env python
"""Bytes of split-K partial tiles + parameter-gradient slots one training step writes into the wgrad arena (what the
step's multi-tensor reduce has to read back), at the bench shape, bf16 mode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import styler_amd
from styler_amd import rt, ops
from styler_amd.training import TrainState, forward_backward, add_pair_inputs
dev = torch.device("cuda")
torch.manual_seed(0)
m = styler_amd.STYLER().to(dev).train()
st = TrainState(m)
from closed_form import make_batch
b = make_batch(48, 20, 60, 2, 13, seed=1234)            # the bench's rank-0 batch
bd = add_pair_inputs({k: v.to(dev) for k, v in b.items()})
rt.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
for i in range(4):
    st._accum = 0
    forward_backward(m, st, bd)
    torch.cuda.synchronize()
    print(f"pass {i}: arena floats used {st.arena.used}  ({st.arena.used * 4 / 1e6:.1f} MB), demand {st.arena.total * 4 / 1e6:.1f} MB, "
          f"gradient {st.flat_g.numel() * 4 / 1e6:.1f} MB")

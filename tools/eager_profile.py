"""Where does an EAGER train step spend its time?  Wall time (device-synchronised) per phase + top host functions."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from closed_form import make_batch
import styler_amd
from styler_amd import rt, ops, hparams as hp
from styler_amd.training import TrainState, train_step, train_losses
dev = torch.device("cuda")
torch.manual_seed(0)
m = styler_amd.STYLER().to(dev).train()
rt.set_precision("bf16"); rt.strict_inputs = False
st = TrainState(m)
b = {k: v.to(dev) for k, v in make_batch(48, 20, 60, 2, 13, seed=1234).items()}
for _ in range(4):
    train_step(m, st, b)
torch.cuda.synchronize()

def phase(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{name:28s} host {1e3 * (t1 - t0):7.2f} ms   host+device {1e3 * (time.perf_counter() - t0):7.2f} ms")
    return r

for it in range(2):
    st.zero_grad(); st.drop_epoch.add_(1); st.zero_slab.begin(dev); ops.zero_slab = st.zero_slab
    losses = phase("forward + losses", lambda: train_losses(m, b))
    st.arena.begin(dev); ops.wgrad_arena = st.arena
    phase("backward", lambda: (losses[0] / hp.acc_steps).backward())
    phase("arena flush", lambda: st.arena.flush(dev))
    ops.wgrad_arena = None; ops.zero_slab = None
    phase("state.step", st.step)

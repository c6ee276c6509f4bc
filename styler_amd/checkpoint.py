"""Checkpoints in the reference's format (train.py:54,61-66,221-224):

    torch.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict()}, 'checkpoint_{step}.pth.tar')

with `model = nn.DataParallel(STYLER())` (every key carries the `module.` prefix, 328 tensors) and
`optimizer = torch.optim.Adam(model.parameters(), betas, eps, weight_decay)` (train.py:52-53: the parameter list
includes the frozen `position_enc` x2 / `pitch_bins` / `energy_bins`, which never get optimizer state).

`training.TrainState` keeps parameters, gradients and the two Adam moments in FLAT fp32 buffers; the functions here map
those buffers to and from torch.optim.Adam's per-parameter `{step, exp_avg, exp_avg_sq}` layout.  They are plain
torch tensor code (no kernels), so the round trip is covered on the CPU."""
import torch

from . import hparams as hp

PREFIX = "module."


def model_state_dict(model, prefix=PREFIX):
    """`nn.DataParallel(model).state_dict()`: the 328 reference keys with the `module.` prefix."""
    if hasattr(model, "module"):
        model = model.module
    return {prefix + k: v for k, v in model.state_dict().items()}


def load_model_state_dict(model, sd, strict=True):
    """Accepts the reference checkpoint's `module.`-prefixed keys as well as bare ones.  In-place copies: parameters that
    are views of a TrainState's flat buffer stay views."""
    if hasattr(model, "module"):
        model = model.module
    if sd and all(k.startswith(PREFIX) for k in sd):
        sd = {k[len(PREFIX):]: v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)


def _param_slots(model, params):
    """[(index in list(model.parameters()), flat offset, numel, shape)] of the trainable parameters, TrainState's
    packing (every view starts 16-byte aligned)."""
    index = {id(p): i for i, p in enumerate(model.parameters())}
    out, off = [], 0
    for p in params:
        out.append((index[id(p)], off, p.numel(), p.shape))
        off += (p.numel() + 3) & ~3
    return out


def adam_state_from_flat(model, params, flat_m, flat_v, adam_steps, lr):
    """torch.optim.Adam(model.parameters(), ...).state_dict() equivalent of the flat moments.  Parameters whose second
    moment is still all zero never received a gradient (torch skips `grad is None` parameters: no state entry) --
    `pitch_norm_linear`, which the forward never calls (modules.py:253 vs 335-350)."""
    n_all = len(list(model.parameters()))
    state = {}
    if adam_steps > 0:
        for idx, off, k, shape in _param_slots(model, params):
            v = flat_v[off:off + k]
            if not bool((v != 0).any()):
                continue
            state[idx] = {"step": torch.tensor(float(adam_steps)),
                          "exp_avg": flat_m[off:off + k].detach().clone().view(shape),
                          "exp_avg_sq": v.detach().clone().view(shape)}
    group = {"lr": float(lr), "betas": tuple(hp.betas), "eps": hp.eps, "weight_decay": hp.weight_decay,
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "decoupled_weight_decay": False, "params": list(range(n_all))}
    return {"state": state, "param_groups": [group]}


def flat_from_adam_state(sd, model, params, flat_m, flat_v):
    """Fill the flat moments from a torch.optim.Adam state dict (reference checkpoints included: torch 1.6 stores `step`
    as an int, current torch as a tensor).  Returns Adam's step count (0 for an empty state).  Parameters without a state
    entry get zero moments; every entry present must carry the same step (one optimizer, no frozen phases)."""
    groups = sd["param_groups"]
    ids = [i for g in groups for i in g["params"]]
    n_all = len(list(model.parameters()))
    if len(ids) != n_all:
        raise ValueError(f"optimizer state was saved for {len(ids)} parameters, the model has {n_all}")
    pos = {pid: k for k, pid in enumerate(ids)}            # saved id -> position in model.parameters()
    by_pos = {pos[pid]: st for pid, st in sd["state"].items()}
    flat_m.zero_()
    flat_v.zero_()
    steps = set()
    for idx, off, k, _ in _param_slots(model, params):
        st = by_pos.pop(idx, None)
        if st is None:
            continue
        if st["exp_avg"].numel() != k:
            raise ValueError(f"optimizer state of parameter {idx}: {st['exp_avg'].numel()} elements, expected {k}")
        flat_m[off:off + k].copy_(st["exp_avg"].reshape(-1))
        flat_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
        steps.add(int(st["step"].item() if torch.is_tensor(st["step"]) else st["step"]))
    if by_pos:
        raise ValueError(f"optimizer state for frozen / unknown parameters: {sorted(by_pos)}")
    if len(steps) > 1:
        raise ValueError(f"per-parameter Adam steps differ ({sorted(steps)}): not a single-optimizer checkpoint")
    return steps.pop() if steps else 0


def save_checkpoint(path, model, state):
    """train.py:221-224."""
    torch.save({"model": model_state_dict(model), "optimizer": state.state_dict()}, path)


def load_checkpoint(path, model, state=None, restore_step=None, map_location="cpu"):
    """train.py:61-66 (+ `ScheduledOptim(..., args.restore_step)`, train.py:54-55): model weights, Adam moments, and the
    Noam counter set to `restore_step` (default: Adam's own step count, which equals it for acc_steps == 1)."""
    ckpt = torch.load(path, map_location=map_location) if isinstance(path, str) else path
    load_model_state_dict(model, ckpt["model"])
    if state is not None:
        state.load_state_dict(ckpt["optimizer"], restore_step=restore_step)
    return ckpt

"""Noam learning-rate schedule behind the reference's `optimizer.ScheduledOptim` interface (optimizer.py:4-32;
hyper-parameters hparams.py:93-101).

    lr(n) = d_model^-0.5 * min(n^-0.5, n * warmup^-1.5)

with the step counter advanced BEFORE the rate is evaluated (the first update runs at n = restore_step + 1).  The fused
clip + Adam path (`training.TrainState`) evaluates the same `noam_lr`; this wrapper exists for callers that keep a
`torch.optim` optimizer (train.py:52-58)."""


def noam_lr(step, d_model=256, warmup=4000):
    """Learning rate of update number `step`; 0 at step 0 (np.power(0, -0.5) = inf loses the min in optimizer.py:22-25)."""
    if step <= 0:
        return 0.0
    return d_model ** -0.5 * min(step ** -0.5, step * warmup ** -1.5)


class ScheduledOptim:
    """Same constructor, public methods and public attributes (`n_warmup_steps`, `n_current_steps`, `init_lr`) as the
    reference wrapper."""

    def __init__(self, optimizer, d_model, n_warmup_steps, current_steps):
        self._optimizer, self._d_model = optimizer, d_model
        self.n_warmup_steps, self.n_current_steps = n_warmup_steps, current_steps
        self.init_lr = d_model ** -0.5

    def _get_lr_scale(self):
        return noam_lr(self.n_current_steps, self._d_model, self.n_warmup_steps) / self.init_lr

    def _update_learning_rate(self):
        self.n_current_steps += 1
        rate = noam_lr(self.n_current_steps, self._d_model, self.n_warmup_steps)
        for group in self._optimizer.param_groups:
            group["lr"] = rate
        return rate

    def step_and_update_lr(self):
        self._update_learning_rate()
        self._optimizer.step()

    def zero_grad(self):
        self._optimizer.zero_grad()


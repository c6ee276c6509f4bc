"""Training step of the path (reference train.py:99-186): forward (clean + noisy decode), clean losses,
noisy mel losses, DAT pass + DAT loss, backward (HIP kernels through the autograd tape), gradient all-reduce,
clip_grad_norm_(1.0) + Adam with the Noam schedule -- the last two fused over FLAT fp32 buffers."""
import torch

from . import hparams as hp
from . import autograd as AG
from . import ops
from .dist import BUCKET_BYTES, Bf16Reducer, allreduce_sum_, world_size
from .loss import DomainAdversarialTrainingLoss, STYLERLoss
from .optimizer import noam_lr
from .runtime import Derived, rt


_LAZY_DERIVED = bool(int(__import__("os").environ.get("STYLER_LAZY_DERIVED", "0")))   # A/B switch: per-entry refresh


class TrainState:
    """Flat fp32 parameter / gradient / Adam-moment buffers.  Every trainable nn.Parameter becomes a view of
    `flat_p`, its `.grad` a view of `flat_g` (so the backward kernels' atomics, the RCCL all-reduce, the global
    norm and the Adam update all run over one contiguous 117.9 MB buffer)."""

    def __init__(self, model, restore_step=0, broadcast=True):
        import torch.distributed as dist
        self.model = model.module if hasattr(model, "module") else model
        model = self.model
        params = [p for p in model.parameters() if p.requires_grad]
        align = lambda k: (k + 3) & ~3                  # every view starts 16-byte aligned (float4 / MFMA staging loads)
        n = sum(align(p.numel()) for p in params)
        dev = params[0].device
        self.flat_p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self.flat_g[off:off + k].view(p.shape)
                off += align(k)
        self.params = params
        self.n = n
        # flat offset where the decoder / mel_linear / PostNet parameters start (registration order of STYLER puts
        # style_modeling first): their gradients are final as soon as backward reaches the decoder INPUT, so their
        # all-reduce is launched from there and overlaps the rest of backward (style encoders, predictors, DAT pass)
        self.tail_start = self._tail_offset(model, params, align)
        self._tail_works = None
        # rt.ar_text_point: the text encoder's parameters open the flat buffer ([0, text_end), 23.3 MB); 0 = no such prefix
        self.text_end = self._prefix_end(model, params, align, "style_modeling.style_encoder.text_encoder.")
        self._text_works = None
        self.text_hook = None                          # set while GraphedTrainStep captures (the second cut)
        self.ar_events = None          # diagnostic: a list here makes step() / on_decoder_grads_ready time the collectives
        self._ar_t0 = None
        # optional bf16 transport of the gradient all-reduce (dist.Bf16Reducer; off by default)
        self.reducer = (Bf16Reducer(self.flat_g) if __import__("os").environ.get("STYLER_ALLREDUCE_BF16", "0") == "1"
                        else None)
        self.n_current_steps = restore_step            # optimizer.py:10: the Noam counter (ScheduledOptim)
        # torch.optim.Adam's own per-parameter `step` (bias correction): a separate counter in the reference -- a run
        # resumed with restore_step > 0 but without optimizer state starts Adam at 0 while the schedule continues
        self.adam_steps = 0
        self._accum = 0                                # micro-batches accumulated since the last update (acc_steps)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float64)
        self.arena = ops.WgradArena()                  # split-K partials of every weight gradient of one backward
        self.zero_slab = ops.ZeroSlab()
        self.split_hook = None                         # set while GraphedTrainStep captures its two graphs
        self._flush_stream, self._flush_side, self._flush_keep = None, None, None     # rt.early_flush
        self.overlap_allreduce = True                  # start the decoder-side all-reduce from inside backward
        # device step counter mixed into every dropout seed: host seeds are baked into a captured hipGraph, the
        # counter is what changes between replays (styler_set_dropout_counter)
        self.drop_epoch = torch.zeros(1, device=dev, dtype=torch.int64)
        ops.set_dropout_counter(self.drop_epoch)
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if world > 1:
            # one dropout stream per rank (the per-process call counter and the device step counter are identical on
            # every rank), and rank 0's initial weights everywhere (what DataParallel's replicate does every step)
            rt.seed = rt.base_seed * world + dist.get_rank()
            if broadcast:
                dist.broadcast(self.flat_p, 0)
                rt.weights_epoch += 1

    def close(self):
        """Unregister this state's device step counter from the library (the dropout kernels dereference the
        registered address: it must not outlive the tensor)."""
        if getattr(self, "drop_epoch", None) is not None and ops.dropout_counter_is(self.drop_epoch):
            ops.set_dropout_counter(None)
        self.drop_epoch = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- checkpoint (train.py:61-66, 221-224) -----------------------------------------------------------------------
    def state_dict(self):
        """`torch.optim.Adam(model.parameters(), ...).state_dict()` of the flat moments (checkpoint.py)."""
        from .checkpoint import adam_state_from_flat
        lr = noam_lr(max(self.n_current_steps, 1), hp.encoder_hidden, hp.n_warm_up_step)
        return adam_state_from_flat(self.model, self.params, self.flat_m, self.flat_v, self.adam_steps, lr)

    def load_state_dict(self, sd, restore_step=None):
        """Consume a torch.optim.Adam state dict (e.g. `checkpoint['optimizer']` of a reference run) and set the Noam
        counter as `ScheduledOptim(optimizer, d_model, n_warm_up_step, restore_step)` does (train.py:54-55)."""
        from .checkpoint import flat_from_adam_state
        self.adam_steps = flat_from_adam_state(sd, self.model, self.params, self.flat_m, self.flat_v)
        self.n_current_steps = self.adam_steps if restore_step is None else int(restore_step)
        self.flat_g.zero_()
        self._accum = 0

    @staticmethod
    def _tail_offset(model, params, align):
        ids = {id(p) for n_, p in model.named_parameters() if not n_.startswith("style_modeling.")}
        off = 0
        for p in params:
            if id(p) in ids:
                return off
            off += align(p.numel())
        return off

    @staticmethod
    def _prefix_end(model, params, align, prefix):
        """Flat offset behind the parameters whose names start with `prefix`, if exactly those open the buffer (else 0)."""
        names = {id(p): n_ for n_, p in model.named_parameters()}
        off, seen_other = 0, False
        end = 0
        for p in params:
            if names.get(id(p), "").startswith(prefix):
                if seen_other:
                    return 0
                end = off + align(p.numel())
            else:
                seen_other = True
            off += align(p.numel())
        return end

    def zero_grad(self, fill=True):
        """`fill=False`: bookkeeping only, the caller clears flat_g (forward_backward: one launch with the slab clear)."""
        if fill:
            self.flat_g.zero_()
        self._tail_works = None
        self._text_works = None
        if self.reducer is not None:                  # a skipped step must not leave ranges that finish() would cast up again
            self.reducer.pending = []
        self._accum = 0

    def _start_allreduce(self, lo, hi):
        """Begin the SUM all-reduce of flat_g[lo:hi] -> work handles (fp32 in place, or through the bf16 shadow)."""
        if self.reducer is not None:
            return self.reducer.start(lo, hi)
        return allreduce_sum_(self.flat_g[lo:hi])

    def early_flush_on_side(self):
        """rt.early_flush (one rank, no all-reduce to start): called from BucketEmbedAddFn.backward like on_decoder_grads_ready.
        The decoder-side weight-gradient work that waits for a flush -- the grouped launches of the decoder's deferred Linear
        gradients and the HBM-bound fold of all split-K partials so far (54 % of the gradient bytes) -- runs on a side stream
        next to the rest of backward (the AudioEncoder's MFMA-bound convolutions): one fork here, one join in front of the
        final flush.  Nothing behind this point reads or writes those gradients before the optimiser."""
        if self._flush_side is not None:
            return
        main = torch.cuda.current_stream()
        if self._flush_stream is None:
            self._flush_stream = torch.cuda.Stream(device=self.flat_g.device)
        side = self._flush_stream
        side.wait_stream(main)
        keep = list(self.arena.group_keep)              # operands of the grouped launches: alive until the join
        with torch.cuda.stream(side):
            self.arena.flush(self.flat_g.device)
        self._flush_side, self._flush_keep = side, keep

    def join_early_flush(self):
        if self._flush_side is not None:
            torch.cuda.current_stream().wait_stream(self._flush_side)
            self._flush_side, self._flush_keep = None, None

    def on_decoder_grads_ready(self):
        """Called from BucketEmbedAddFn.backward (both decode branches fully back-propagated): start the all-reduce
        of the decoder + mel_linear + PostNet gradient range (55 % of the bytes) while backward continues."""
        if self._tail_works is None:
            self.arena.flush(self.flat_g.device)      # fold the decoder-side split-K partials before they are reduced
            self._tail_works = self._start_allreduce(self.tail_start, self.n)
            if self.ar_events is not None:            # diagnostic (bench.py, N > 1): when the overlapped range was launched
                self._ar_t0 = torch.cuda.Event(enable_timing=True)
                self._ar_t0.record()

    def on_text_grads_ready(self):
        """rt.ar_text_point -- called from EmbedPosFn.backward (the text encoder fully back-propagated; the AudioEncoder's backward
        is still to come): start the all-reduce of the first range of the flat gradient, [0, text_end)."""
        if self._text_works is None and self.text_end > 0:
            self.arena.flush(self.flat_g.device)      # fold the split-K partials written so far (the text encoder's among them)
            self._text_works = self._start_allreduce(0, self.text_end)

    def lr(self):
        """optimizer.py:21-32: the counter is incremented BEFORE the rate is computed."""
        self.n_current_steps += 1
        return noam_lr(self.n_current_steps, hp.encoder_hidden, hp.n_warm_up_step)

    def step(self):
        """nn.utils.clip_grad_norm_(params, 1.0) + ScheduledOptim.step_and_update_lr() (train.py:181-185)."""
        ev = None
        if self.ar_events is not None:                       # diagnostic: main-stream time spent blocked in the collective
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        overlapped = self._tail_works is not None
        lo = self.text_end if self._text_works is not None else 0   # (the text encoder's range may be in flight as well)
        if overlapped:                                       # tail range already in flight: reduce only the head
            works = self._tail_works + (self._text_works or []) + self._start_allreduce(lo, self.tail_start)
        else:
            works = (self._text_works or []) + self._start_allreduce(lo, self.n)
        for w in works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self.ar_events.append((self._ar_t0 if overlapped else None, ev[0], ev[1]))
            self._ar_t0 = None
        if self.reducer is not None:
            self.reducer.finish()
        self._tail_works = None
        self._text_works = None
        self._accum = 0
        lr = self.lr()
        self.adam_steps += 1
        self.sumsq.zero_()
        ops.sumsq(self.flat_g, self.sumsq)
        ops.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.sumsq, hp.grad_clip_thresh, lr,
                      hp.betas[0], hp.betas[1], hp.eps, self.adam_steps, grad_scale=1.0 / world_size())
        rt.weights_epoch += 1       # the flat update bypasses torch's version counters: invalidate derived layouts
        if not _LAZY_DERIVED:
            Derived.refresh_all()   # ... and rebuild every declared one (bf16 shadows, kernel layouts) with ONE launch
        return lr

    def grad_norm(self):
        """Global L2 norm of what the flat gradient buffer holds (after `step()` on several ranks: the rank SUM)."""
        self.sumsq.zero_()
        ops.sumsq(self.flat_g, self.sumsq)
        return float(self.sumsq.sqrt().item())

    def allreduce_timing(self):
        """Summary of the events `ar_events` collected (set it to [] before a few diagnostic steps): per step, `exposed_ms` = the
        main stream's time inside step()'s collective waits (the head range's whole all-reduce + what was left of the overlapped
        tail range), `overlap_window_ms` = from the tail range's launch (backward has left both decode branches) to step() --
        the time the tail's all-reduce had to hide in.  Medians; the caller synchronises first."""
        ev = self.ar_events or []
        if not ev:
            return None
        med = lambda v: sorted(v)[len(v) // 2]
        out = {"steps": len(ev), "exposed_ms": round(med([a.elapsed_time(b) for _, a, b in ev]), 3)}
        win = [t0.elapsed_time(a) for t0, a, _ in ev if t0 is not None]
        if win:
            out["overlap_window_ms"] = round(med(win), 3)
        es = 2 if self.reducer is not None else 4
        text = self.text_end if rt.ar_text_point else 0
        out["exposed_bytes"] = (self.tail_start - text) * es
        out["overlapped_bytes"] = (self.n - self.tail_start + text) * es
        return out

    def allreduce_info(self):
        """What one step exchanges (bench.py prints it, so a silent fallback of the overlap is visible)."""
        tail = self.n - self.tail_start
        text = self.text_end if rt.ar_text_point else 0
        es = 2 if self.reducer is not None else 4
        nb = lambda k: (k * es + BUCKET_BYTES - 1) // BUCKET_BYTES
        return {"allreduce_bytes": self.n * es, "allreduce_dtype": "bf16" if es == 2 else "fp32",
                "allreduce_bucket_bytes": BUCKET_BYTES,
                "allreduce_buckets": nb(tail) + nb(self.tail_start - text) + (nb(text) if text else 0),
                "allreduce_overlapped_bytes": (tail + text) * es, "allreduce_launch_points": 3 if text else 2,
                "allreduce_world": world_size()}


_seed_cache = {}


def _seed_grad(value, device):
    key = (float(value), str(device))
    if key not in _seed_cache:
        _seed_cache[key] = torch.full((), float(value), device=device, dtype=torch.float32)
    return _seed_cache[key]


PAIR_KEYS = {"pair_mel": ("mel_target", "mel_aug"), "pair_f0n": ("f0_norm", "f0_norm_aug"),
             "pair_ein": ("energy_input", "energy_input_aug"), "pair_mela": ("mel_aug", "mel_aug"),
             "pair_mel_len": ("mel_len", "mel_len"), "pair_src_len": ("src_len", "src_len")}


def add_pair_inputs(batch):
    """The stacked [2B, ...] inputs of the AudioEncoder's combined main + DAT pass (rt.pair_audio; train.py:135-136 and
    149-150 feed the same module with (mel_target, f0_norm, energy_input, mel_aug) and with (mel_aug, f0_norm_aug,
    energy_input_aug, mel_aug)) as part of the batch: a layout the FEED produces once per batch (data.BatchFeeder collates
    into it), not six concatenation kernels inside every step.  Adds the keys in place; returns the batch."""
    for k, (a, b) in PAIR_KEYS.items():
        if k not in batch:
            batch[k] = torch.cat([batch[a], batch[b]])
    return batch


def train_losses(model, batch, loss_fn=None, dat_fn=None):
    """The ten scalars of one step (total first), train.py:135-160.  `batch` holds CUDA tensors."""
    loss_fn = loss_fn or STYLERLoss()
    dat_fn = dat_fn or DomainAdversarialTrainingLoss()
    B = batch["text"].shape[0]
    S, T = batch["text"].shape[1], batch["mel_target"].shape[1]
    dev = batch["text"].device
    se = model.style_modeling.style_encoder
    if rt.pair_audio:
        se.dat_inputs = (batch["mel_aug"], batch["f0_norm_aug"], batch["energy_input_aug"],
                         batch if "pair_mel" in batch else None)
    rt.pair_lens = (batch["mel_len"], batch["pair_mel_len"]) if "pair_mel_len" in batch else None
    try:
        out = model(batch["text"], batch["mel_target"], batch["mel_aug"], batch["f0_norm"], batch["energy_input"],
                    batch["src_len"], batch["mel_len"], batch["D"], batch["f0"], batch["energy"], S, T,
                    speaker_embed=batch["speaker_embed"])
    finally:
        rt.pair_lens = None
    (mel, mel_n), (post, post_n), log_d, p_pred, e_pred, src_mask, mel_mask, _, aug = out
    # labels of train.py:139,152 (zeros for the clean pass, ones for the DAT pass): passed as python ints, the NLL kernel
    # needs no label tensor then; the masks are consumed as lengths (loss.py docstring), so no ~mask launches either
    zeros, ones = 0, 1
    sm = model.style_modeling
    if (rt.fused_loss and type(loss_fn) is STYLERLoss and type(dat_fn) is DomainAdversarialTrainingLoss
            and torch.is_grad_enabled() and getattr(sm, "dat_posteriors", None) is not None and mel.requires_grad):
        # Round 6: the whole loss head as two tape nodes -- the seven masked-error means of STYLERLoss.forward + cal_mel_loss
        # (loss.py:16-50) in ONE launch each way, and the two classifier NLL3 terms + the weighted total of train.py:156-160 in
        # ONE launch each way (autograd.LossTailFn): 4 launches on the serial chain between forward and backward instead of 10.
        # Same kernels / same arithmetic per term as the loss modules (STYLER_FUSED_LOSS=0: the modules).
        from . import loss as L
        lens, slens = batch["mel_len"], batch["src_len"]
        mel_l, post_l, d_l, p_l, e_l, mel_nl, post_nl = L._masked_means([
            (mel, batch["mel_target"], 0, lens), (post, batch["mel_target"], 0, lens), (log_d, batch["log_D"], 1, slens),
            (p_pred, batch["f0"], 1, lens), (e_pred, batch["energy"], 1, lens),
            (mel_n, batch["mel_aug"], 0, lens), (post_n, batch["mel_aug"], 0, lens)])
        dat_post, sm.dat_posteriors = sm.dat_posteriors, None
        weights = (1.0,) * 7 + (float(hp.dat_weight),) * 2
        total, cls, cls_dat = AG.LossTailFn.apply(weights, zeros, ones, mel_l, post_l, mel_nl, post_nl, d_l, p_l, e_l,
                                                  *aug, *dat_post)
        return total, mel_l, post_l, mel_nl, post_nl, d_l, p_l, e_l, cls, cls_dat
    mel_l, post_l, d_l, p_l, e_l, cls = loss_fn(log_d, batch["log_D"], p_pred, batch["f0"], e_pred, batch["energy"],
                                                mel, post, batch["mel_target"], None, None,
                                                batch["src_len"], batch["mel_len"], aug, zeros)
    mel_nl, post_nl = loss_fn.cal_mel_loss(mel_n, post_n, batch["mel_aug"], None, batch["mel_len"])
    sm = model.style_modeling
    if getattr(sm, "dat_posteriors", None) is not None:   # the forward ran the classifiers on both passes' encodings at once
        dat_post, sm.dat_posteriors = sm.dat_posteriors, None
    else:
        if se.dat_encodings is not None:            # the forward ran the DAT pass in the same AudioEncoder batch
            (d, p, e), se.dat_encodings = se.dat_encodings, None
        else:
            enc_cat = se.encoder_input_cat(batch["mel_aug"], batch["f0_norm_aug"], batch["energy_input_aug"],
                                           batch["mel_aug"])
            d, p, e, _ = se.audio_encoder(enc_cat, batch["mel_len"], batch["src_len"], mask=None, max_seq_len=S)
        dat_post = (sm.augmentation_classifier_d(d), sm.augmentation_classifier_p(p), sm.augmentation_classifier_e(e))
    cls_dat = dat_fn(dat_post, ones)
    terms = (mel_l, post_l, mel_nl, post_nl, d_l, p_l, e_l, cls, cls_dat)
    weights = (1.0,) * 7 + (float(hp.dat_weight),) * 2
    if torch.is_grad_enabled() and any(t.requires_grad for t in terms):
        total = AG.WeightedSumFn.apply(weights, *terms)             # train.py:156-160 in one launch (and one in backward)
    else:
        total = ops.weighted_sum([t.reshape(1) for t in terms], weights).view(())
    return total, mel_l, post_l, mel_nl, post_nl, d_l, p_l, e_l, cls, cls_dat


def forward_backward(model, state, batch, loss_fn=None, dat_fn=None):
    """Everything of one step that runs on the device without host decisions: forward, the ten losses, backward into
    the flat gradient (train.py:135-176).  Capturable in a hipGraph.  The gradient buffer is cleared at the start of an
    accumulation window only (the reference zeroes after each optimiser update, train.py:185; with acc_steps > 1 the
    micro-batches in between add up)."""
    clear_g = state._accum == 0
    if clear_g:
        state.zero_grad(fill=False)
    state._accum += 1
    slab = state.zero_slab.begin(state.flat_g.device, fill=False)   # the norm kernels' statistics workspaces: one clear per step
    # gradient clear (accumulation windows: the first micro-batch only), slab clear and the dropout step counter: ONE launch
    ops.step_begin(state.flat_g if clear_g else None, slab, state.drop_epoch)
    ops.zero_slab = state.zero_slab
    ops.x3_cache = {} if (rt.prec == ops.PREC_BF16X3 and rt.x3_cache) else None   # (bf16x3: splits live until the step ends)
    try:
        losses = train_losses(model, batch, loss_fn, dat_fn)
        # The decoder-side all-reduce may only start from the LAST micro-batch of an accumulation window: an earlier one
        # would reduce a partial sum while the next micro-batch is still adding to it (and those additions would never be
        # reduced) -- round-2 advisor finding.  `state._accum` already counts this micro-batch.
        last_micro = state._accum % hp.acc_steps == 0
        rt.grad_ready_hook = state.split_hook or (state.on_decoder_grads_ready
                                                  if (state.overlap_allreduce and last_micro) else None)
        if rt.grad_ready_hook is None and rt.early_flush:
            rt.grad_ready_hook = state.early_flush_on_side
        rt.text_ready_hook = state.text_hook or (state.on_text_grads_ready if (state.overlap_allreduce and last_micro and
                                                                                 rt.ar_text_point and world_size() > 1) else None)
        state.arena.begin(state.flat_g.device)
        ops.wgrad_arena = state.arena
        # loss / acc_steps (train.py:175) as the seed gradient of backward: no division kernel, no ones_like
        losses[0].backward(gradient=_seed_grad(1.0 / hp.acc_steps, losses[0].device))
        state.join_early_flush()
        state.arena.flush(state.flat_g.device)         # one launch folds all split-K partials into flat_g
    finally:
        rt.grad_ready_hook = None
        rt.text_ready_hook = None
        ops.wgrad_arena = None
        ops.zero_slab = None
        ops.x3_cache = None
        ops.loss_side_stream = None
        state.join_early_flush()                        # (error paths: never leave a forked stream behind)
    return losses


def train_step(model, state, batch, loss_fn=None, dat_fn=None):
    """One iteration of the reference loop (train.py:135-186).  Returns the 10 loss scalars (device tensors) and the lr of
    the update -- None when the `acc_steps` gate (train.py:177-178) skipped it: the gradient keeps accumulating."""
    losses = forward_backward(model, state, batch, loss_fn, dat_fn)
    if state._accum % hp.acc_steps != 0:
        return losses, None
    lr = state.step()
    return losses, lr


class GraphedTrainStep:
    """The training step with forward + losses + backward replayed from hipGraphs (about 900 kernel launches per step;
    launched eagerly the host needs as long to enqueue them as the GPU needs to run them).  Outside the graphs stay the
    steps that take host decisions: the RCCL all-reduce of the flat gradient, the Noam learning rate and the fused
    clip + Adam launch (`TrainState.step`).

    `split=True` (default when torch.distributed runs with more than one rank) captures TWO graphs, cut where backward has
    finished both decode branches (BucketEmbedAddFn.backward): the decoder + mel_linear + PostNet gradient range (55 % of
    the bytes) is final there, so its bucketed all-reduce is launched between the two replays and runs on RCCL's stream
    while the second graph back-propagates the style encoders and predictors.  `split=False`: one graph, the whole buffer
    is reduced after it.

    The graphs are captured for the shapes of `batch`; `__call__(batch)` copies a new batch of the same shapes into the
    static input tensors (callers bucket their batches by padded shape and keep one instance per bucket).  The
    reference's host-side range assertion on p_norm / e_input (utils.py:423) is not evaluated inside the graphs."""

    def __init__(self, model, state, batch, warmup=3, loss_fn=None, dat_fn=None, split=None):
        import torch.distributed as dist
        self.model, self.state = model, state
        self.static = {k: v.clone() for k, v in batch.items()}
        if rt.pair_audio:
            add_pair_inputs(self.static)                      # the stacked AudioEncoder inputs: static buffers of the graph
        if split is None:
            split = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if hp.acc_steps != 1:
            raise ValueError("GraphedTrainStep replays a whole optimisation step: acc_steps must be 1 "
                             "(use train_step for gradient accumulation)")
        # Everything a captured graph addresses by raw pointer must live exactly as long as the graph.  The split-K
        # workspace arena (its buffer, its reduce / group descriptor tables) and the zero slab are per-pass scratch that
        # the shared TrainState re-sizes when a later, larger batch shape asks for more -- so every graphed step OWNS its
        # own pair: a second bucket's warm-up can neither free nor re-use memory this graph writes to on replay
        # (round-2 advisor finding; tests/test_90_equivalences.py::test_graph_cache_two_buckets_small_then_large).
        self.arena, self.zero_slab = ops.WgradArena(), ops.ZeroSlab()
        self.arena.owned_by_graph = True
        shared = state.arena, state.zero_slab, state.overlap_allreduce
        state.arena, state.zero_slab = self.arena, self.zero_slab
        state.overlap_allreduce = False                       # inside a graph the cut is the split hook, not the eager hook
        strict, rt.strict_inputs = rt.strict_inputs, False    # the [0, 1] input assertion is a host sync (utils.py:423)
        # The host-side dropout call counter advances during the capture pass and its values are baked into the graph as
        # seeds: left advanced, the dropout streams of every later eager or captured step would depend on how many shapes
        # had been captured before (round-3 advisor finding).  Every graph is captured from the SAME counter value (the
        # device step counter is what makes replays differ) and the counter is restored afterwards.
        calls0 = rt.dropout_calls
        try:
            self._warmup(model, state, warmup, loss_fn, dat_fn, split)
            rt.dropout_calls = calls0
            self.graphs = None
            if split:
                try:
                    self._capture_split(model, state, loss_fn, dat_fn)
                except Exception as e:                  # keep training alive: fall back to the single graph
                    import warnings
                    warnings.warn(f"split hipGraph capture failed ({type(e).__name__}: {e}); using one graph")
                    self.graphs = None
                    torch.cuda.synchronize()
            if self.graphs is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g), torch.enable_grad():
                    self.losses = forward_backward(model, state, self.static, loss_fn, dat_fn)
                self.graphs = (g,)
            # a capture does not execute: its step-counter increment and BatchNorm momentum updates are part of the
            # graph, nothing to undo
            state._accum = 0
            self.arena.frozen = self.zero_slab.frozen = True    # the graphs hold their addresses: no re-sizing from here on
        finally:
            rt.strict_inputs = strict
            rt.dropout_calls = calls0
            state.arena, state.zero_slab, state.overlap_allreduce = shared

    def _warmup(self, model, state, warmup, loss_fn, dat_fn, split):
        """Eager passes before the capture: they size the wgrad arena / zero slab and build the descriptor tables.  Side
        effects are undone -- constructing an instance (one per padded batch shape) must not train: no optimiser step, no
        all-reduce (ranks may construct instances at different times), and the state the passes do touch (BatchNorm
        running statistics / `num_batches_tracked`, the dropout step counter, the gradient buffer) is restored."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        if split:                                       # same flush pattern as the split capture: its descriptor tables
            state.split_hook = lambda: state.arena.flush(state.flat_g.device)   # must exist before (no H2D in a capture)
            if rt.ar_text_point and state.text_end > 0:
                state.text_hook = lambda: state.arena.flush(state.flat_g.device)
        keep = [(b, b.detach().clone()) for b in model.buffers()]
        epoch, calls, accum = state.drop_epoch.clone(), rt.dropout_calls, state._accum
        grads = state.flat_g.clone() if accum else None
        try:
            with torch.cuda.stream(side), torch.enable_grad():
                # at least `warmup` passes, and then until a pass neither re-sized the arena / the zero slab nor built a new
                # descriptor table: the capture may not upload one (the first pass measures, the second runs with the
                # buffers that measurement sized, grouped launches and slot arrays settle one or two passes later)
                def scratch_state():
                    a, z = state.arena, state.zero_slab
                    return (a.buf.data_ptr() if a.buf is not None else 0, a.buf.numel() if a.buf is not None else 0,
                            z.buf.data_ptr() if z.buf is not None else 0, len(a._cache), len(a._gcache))
                done, prev = 0, None
                while done < warmup or (scratch_state() != prev and done < warmup + 6):
                    prev = scratch_state()
                    state._accum = 0
                    forward_backward(model, state, self.static, loss_fn, dat_fn)
                    done += 1
                if scratch_state() != prev:          # still moving at the cap: the capture would bake a sizing pass in
                    import warnings
                    warnings.warn(f"GraphedTrainStep: the wgrad arena / zero slab / descriptor tables did not settle in {done} "
                                  f"warm-up passes ({prev} -> {scratch_state()}); the captured step may keep a sizing pass's "
                                  f"atomic fallbacks (not bit-reproducible)", RuntimeWarning)
        finally:
            state.split_hook = None
            state.text_hook = None
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for b, v in keep:
                b.copy_(v)
            state.drop_epoch.copy_(epoch)
            if grads is not None:
                state.flat_g.copy_(grads)
            else:
                state.flat_g.zero_()
        rt.dropout_calls, state._accum = calls, accum
        torch.cuda.synchronize()

    def _capture_split(self, model, state, loss_fn, dat_fn):
        """Two graphs sharing one memory pool, cut inside backward by the decoder-gradients-ready hook.  Backward must run
        on the capturing thread for that (a stream capture is ended by the thread that began it)."""
        import gc
        three = bool(rt.ar_text_point and state.text_end > 0)    # a second cut where the text encoder's gradients are final
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        gc_ = torch.cuda.CUDAGraph() if three else None
        cut = {"done": False, "text": False}

        def split_here():
            if cut["done"]:
                return
            state.arena.flush(state.flat_g.device)      # fold the decoder-side split-K partials: that range is final now
            ga.capture_end()                            # (the flush joined the loss-only side stream: no forked stream is left)
            gb.capture_begin(pool=ga.pool())
            if ops.loss_side_stream is not None:        # ... and forks it again: the rest of its backward chain belongs to graph B
                ops.loss_side_stream.wait_stream(torch.cuda.current_stream())
            cut["done"] = True

        def split_text():
            if cut["text"] or not cut["done"]:
                return
            state.arena.flush(state.flat_g.device)      # (the text encoder's deferred weight gradients and partial tiles)
            gb.capture_end()
            gc_.capture_begin(pool=ga.pool())
            if ops.loss_side_stream is not None:
                ops.loss_side_stream.wait_stream(torch.cuda.current_stream())
            cut["text"] = True

        gc.collect()
        torch.cuda.empty_cache()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        mt = torch.autograd.is_multithreading_enabled()
        torch.autograd.set_multithreading_enabled(False)
        state.split_hook = split_here
        state.text_hook = split_text if three else None
        try:
            with torch.cuda.stream(stream), torch.enable_grad():
                ga.capture_begin()
                try:
                    self.losses = forward_backward(model, state, self.static, loss_fn, dat_fn)
                finally:
                    (gc_ if cut["text"] else gb if cut["done"] else ga).capture_end()
        finally:
            state.split_hook = None
            state.text_hook = None
            torch.autograd.set_multithreading_enabled(mt)
        torch.cuda.current_stream().wait_stream(stream)
        if not cut["done"]:
            raise RuntimeError("backward never reached the decoder-gradients-ready hook")
        if three and not cut["text"]:
            raise RuntimeError("backward never reached the text-encoder-gradients-ready hook")
        self.graphs = (ga, gb, gc_) if three else (ga, gb)

    def __call__(self, batch=None):
        if batch is not None:
            for k, v in batch.items():
                if k not in self.static:              # e.g. pair_* keys of a BatchFeeder(pairs=True) while rt.pair_audio is off
                    continue
                if self.static[k].shape != v.shape:
                    raise ValueError(f"GraphedTrainStep was captured for {k} of shape {tuple(self.static[k].shape)}, "
                                     f"got {tuple(v.shape)}")
                self.static[k].copy_(v, non_blocking=True)
            if rt.pair_audio and "pair_mel" not in batch:     # a feed that does not collate the stacked layout itself
                B = batch["text"].shape[0]
                for k, (a, b) in PAIR_KEYS.items():
                    self.static[k][:B].copy_(batch[a], non_blocking=True)
                    self.static[k][B:].copy_(batch[b], non_blocking=True)
        st = self.state
        st._accum = 1                                   # the captured pass starts with its own zero_grad
        self.graphs[0].replay()
        if len(self.graphs) >= 2:
            st._tail_works = st._start_allreduce(st.tail_start, st.n)     # overlaps the second graph
            self.graphs[1].replay()
        if len(self.graphs) == 3:                                         # rt.ar_text_point: the text encoder's range overlaps the third
            st._text_works = st._start_allreduce(0, st.text_end)
            self.graphs[2].replay()
        lr = st.step()
        return self.losses, lr


class GraphedStepCache:
    """One `GraphedTrainStep` per padded batch shape, least-recently-used eviction.

    The graphed step is captured for fixed tensor shapes, while the reference's feed (dataset.py:188-207) pads every
    sub-batch to its own maxima: fed as is, nearly every step would be a capture (seconds) instead of a replay.  With the
    feeder's `bucket=(s_step, t_step)` option the rectangles fall on a small grid and this cache replays one graph per grid
    cell.  Constructing an instance has no side effects on the training state (GraphedTrainStep._warmup), so a miss costs
    time only.  With several ranks every rank owns its own cache; nothing collective happens at construction.

    What padding up changes: the valid positions, masks and losses are identical; the statistics that the reference takes
    over the padded rectangle (GroupNorm over padded T, PostNet BatchNorm over padded rows, the unpacked BiLSTMs, the
    classifier's time mean -- SURVEY 8c trap 1) see the extra zero rows exactly as they see those of a longer co-batched
    utterance: the result equals the reference run on the same padded rectangle (tests/test_14_train_step.py)."""

    def __init__(self, model, state, max_graphs=24, sync_misses=False, **kw):
        """`sync_misses` (several ranks): before every step the ranks exchange their batch shapes (one 3-int all-gather) and
        every rank captures, in the same step, the graph of EVERY shape some rank is about to miss -- its own with its batch,
        a peer's with a synthetic batch of that shape.  Without it each rank stalls its peers (they wait in the step's
        all-reduce) once per shape IT sees for the first time; with it a shape costs one capture stall for the whole job."""
        from collections import OrderedDict
        self.model, self.state, self.max_graphs, self.kw = model, state, max_graphs, kw
        self.steps = OrderedDict()
        self.hits = self.misses = self.prefetched = 0
        self.sync_misses = sync_misses and world_size() > 1

    @staticmethod
    def key(batch):
        return tuple(batch["text"].shape) + (batch["mel_target"].shape[1],)

    def _capture(self, k, batch, keep=()):
        """`keep`: shapes of the current exchange (sync_misses) -- never evicted to make room (round-3 advisor finding: at
        capacity, capturing a peer's shape could evict the very graph this rank was about to replay, and with per-rank LRU
        orders every rank then re-captured on most steps)."""
        while len(self.steps) >= self.max_graphs:
            victim = next((key for key in self.steps if key != k and key not in keep), None)
            if victim is None:
                raise RuntimeError(f"GraphedStepCache: max_graphs = {self.max_graphs} is smaller than the {len(keep) + 1} "
                                   "shapes one step of this job needs")
            del self.steps[victim]                                   # frees the evicted graph's memory pool and arena
        step = self.steps[k] = GraphedTrainStep(self.model, self.state, batch, **self.kw)
        return step

    def prefetch(self, keys):
        """Capture the graphs of the given (B, S, T) shapes up front (e.g. the feeder's bucket grid) on synthetic batches:
        constructing a graphed step has no side effects on the training state, so nothing trains; with several ranks every
        rank calls this with the same list and the capture stalls happen once, together, before the loop."""
        dev = self.state.flat_g.device
        for k in keys:
            k = tuple(int(x) for x in k)
            if k not in self.steps:
                self._capture(k, synthetic_batch(*k, device=dev))
                self.prefetched += 1

    def _exchange(self, k):
        """Shapes the job is about to run this step that at least one rank has no graph for (all ranks get the same list)."""
        import torch.distributed as dist
        dev = self.state.flat_g.device
        mine = torch.tensor(list(k) + [0 if k in self.steps else 1], device=dev, dtype=torch.int64)
        allk = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allk, mine)
        rows = torch.stack(allk).cpu().tolist()
        return sorted({tuple(r[:3]) for r in rows if r[3]})

    def __call__(self, batch):
        k = self.key(batch)
        if self.sync_misses:
            if k in self.steps:
                self.steps.move_to_end(k)                            # this rank's own graph first: most recently used
            missing = self._exchange(k)
            keep = set(missing) | {k}
            for other in missing:
                if other != k and other not in self.steps:
                    self._capture(other, synthetic_batch(*other, device=self.state.flat_g.device), keep=keep)
                    self.prefetched += 1
        step = self.steps.get(k)
        if step is None:
            self.misses += 1
            step = self._capture(k, batch)
        else:
            self.hits += 1
            self.steps.move_to_end(k)
        return step(batch)


def synthetic_batch(B, S, T, device="cpu", seed=0):
    """A batch of the collate's keys, dtypes and padded shapes (dataset.py:188-207; tests/closed_form.make_batch) with every
    item at full length: what GraphedStepCache captures a shape with when no real batch of that shape is at hand (the
    capture only needs shapes; its warm-up passes restore everything they touch)."""
    g = torch.Generator().manual_seed(seed)
    D = torch.full((B, S), T // S, dtype=torch.int64)
    D[:, -1] += T - int(D[0].sum())
    ru = lambda *s: torch.rand(*s, generator=g)
    spk = torch.randn(B, 512, generator=g)
    b = dict(text=torch.randint(1, 152, (B, S), generator=g), mel_target=torch.randn(B, T, 80, generator=g),
             mel_aug=torch.randn(B, T, 80, generator=g), D=D, log_D=torch.log(D.float() + hp.log_offset),
             f0=80.0 + 300.0 * ru(B, T), f0_norm=ru(B, T), f0_norm_aug=ru(B, T), energy=100.0 * ru(B, T),
             energy_input=ru(B, T), energy_input_aug=ru(B, T), speaker_embed=spk / spk.norm(dim=1, keepdim=True),
             src_len=torch.full((B,), S, dtype=torch.int64), mel_len=torch.full((B,), T, dtype=torch.int64))
    return {k: v.to(device) for k, v in b.items()}

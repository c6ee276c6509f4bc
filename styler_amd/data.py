"""Data feed next to the hot path: the reference's `batch_size^2` sorted collate
(dataset.py:116-207, utils.py:296-329) as host-side numpy, plus pinned-memory asynchronous H2D staging so the
15 per-step `.to(device)` copies of train.py:107-130 become one non-blocking batch transfer."""
import math

import numpy as np
import torch

from . import hparams as hp

_KEYS_1D = ("text", "D", "f0", "f0_norm", "f0_norm_aug", "energy", "energy_input", "energy_input_aug")
_KEYS_2D = ("mel_target", "mel_aug")


def pad_1D(inputs, PAD=0):
    """utils.py:296-307."""
    max_len = max(len(x) for x in inputs)
    return np.stack([np.pad(x, (0, max_len - x.shape[0]), mode="constant", constant_values=PAD) for x in inputs])


def pad_2D(inputs, maxlen=None):
    """utils.py:310-329."""
    max_len = maxlen if maxlen else max(np.shape(x)[0] for x in inputs)
    out = []
    for x in inputs:
        if np.shape(x)[0] > max_len:
            raise ValueError("not max_len")
        out.append(np.pad(x, ((0, max_len - np.shape(x)[0]), (0, 0)), mode="constant", constant_values=0))
    return np.stack(out)


def reprocess(batch, cut_list):
    """Dataset.reprocess, dataset.py:131-186: gather the items of one sub-batch, pad with 0, add log_D and lengths."""
    items = [batch[i] for i in cut_list]
    out = {"id": [it["id"] for it in items]}
    for k in _KEYS_1D:
        out[k] = pad_1D([it[k] for it in items])
    for k in _KEYS_2D:
        out[k] = pad_2D([it[k] for it in items])
    out["log_D"] = np.log(out["D"] + hp.log_offset)
    out["speaker_embed"] = np.concatenate([it["speaker_embed"] for it in items], axis=0)
    out["src_len"] = np.array([it["text"].shape[0] for it in items], dtype=np.float64)
    out["mel_len"] = np.array([it["mel_target"].shape[0] for it in items], dtype=np.float64)
    return out


def collate_fn(batch, sort=True):
    """Dataset.collate_fn, dataset.py:188-207: `batch` holds batch_size^2 items; sort by text length (descending)
    and cut into batch_size sub-batches of batch_size items."""
    len_arr = np.array([d["text"].shape[0] for d in batch])
    index_arr = np.argsort(-len_arr)
    real = int(math.sqrt(len(batch)))
    cuts = [index_arr[i * real:(i + 1) * real] if sort else np.arange(i * real, (i + 1) * real) for i in range(real)]
    return [reprocess(batch, c) for c in cuts]


_DTYPES = {"text": torch.long, "D": torch.long, "src_len": torch.long, "mel_len": torch.long,
           "pair_src_len": torch.long, "pair_mel_len": torch.long}


def bucket_up(n, step, cap=hp.max_seq_len + 1):
    """Smallest multiple of `step` >= n, capped at the position table (train mode forbids L > 1001, Models.py:124-125)."""
    if step <= 1:
        return n
    return min(-(-n // step) * step, max(cap, n))


def pad_to_rectangle(sub_batch, S, T):
    """Zero-pad a collated sub-batch up to max_src_len = S, max_mel_len = T (the collate's own padding value, utils.py:
    296-329); lengths are untouched, so every mask / loss sees the same valid positions."""
    out = dict(sub_batch)
    for k in _KEYS_1D + ("log_D",):
        v = np.asarray(sub_batch[k])
        want = S if k in ("text", "D", "log_D") else T
        if v.shape[1] < want:
            out[k] = np.pad(v, ((0, 0), (0, want - v.shape[1])), mode="constant")
    for k in _KEYS_2D:
        v = np.asarray(sub_batch[k])
        if v.shape[1] < T:
            out[k] = np.pad(v, ((0, 0), (0, T - v.shape[1]), (0, 0)), mode="constant")
    # the collate computes log_D over the PADDED D (dataset.py:167: np.log(Ds + hparams.log_offset)), so padded positions hold
    # log(log_offset), not 0: recompute it from the padded D (round-2 advisor finding; the duration loss is length-masked,
    # so this only keeps bucketed and exact batches identical cell for cell)
    if "D" in out and "log_D" in out:
        out["log_D"] = np.log(np.asarray(out["D"]) + hp.log_offset).astype(np.asarray(sub_batch["log_D"]).dtype)
    return out


_PAIRS = (("pair_mel", "mel_target", "mel_aug"), ("pair_f0n", "f0_norm", "f0_norm_aug"),
          ("pair_ein", "energy_input", "energy_input_aug"), ("pair_mela", "mel_aug", "mel_aug"),
          ("pair_mel_len", "mel_len", "mel_len"), ("pair_src_len", "src_len", "src_len"))


def to_device(sub_batch, device, pinned=True, bucket=None, pairs=False):
    """train.py:107-132 in one shot: numpy -> pinned host tensors -> non-blocking H2D on the current stream.
    Returns (tensors dict, max_src_len, max_mel_len).  `bucket` = (s_step, t_step): pad the rectangle up to multiples of
    the steps, so that a few hipGraphs (training.GraphedStepCache) cover every batch of an epoch; the returned maxima are
    the PADDED extents (what `STYLER.forward` must be given as max_src_len / max_mel_len)."""
    S, T = int(np.max(sub_batch["src_len"])), int(np.max(sub_batch["mel_len"]))
    if bucket is not None:
        S, T = bucket_up(S, bucket[0]), bucket_up(T, bucket[1])
        sub_batch = pad_to_rectangle(sub_batch, S, T)
    if pairs:                                      # the stacked [2B, ...] AudioEncoder inputs (training.add_pair_inputs) are a
        sub_batch = dict(sub_batch)                # host-side collate layout: no concatenation kernels inside the step
        for k, a, b in _PAIRS:
            sub_batch[k] = np.concatenate([sub_batch[a], sub_batch[b]], axis=0)
    out = {}
    for k, v in sub_batch.items():
        if k == "id":
            continue
        t = torch.from_numpy(np.ascontiguousarray(v)).to(_DTYPES.get(k, torch.float32))
        if pinned and torch.cuda.is_available():
            t = t.pin_memory()
        out[k] = t.to(device, non_blocking=True)
    return out, S, T


# ---- feature store + prefetching feeder --------------------------------------------------------------------------------
_STORE_FILES = (                                   # key, sub-directory, file tag (dataset.py:90-118)
    ("mel_target", "mel_clean", "mel"), ("mel_aug", "mel_aug", "mel"), ("D", "alignment", "ali"),
    ("f0", "f0", "f0"), ("f0_norm", "f0_norm", "f0"), ("f0_norm_aug", "f0_norm_aug", "f0"),
    ("energy", "energy", "energy"), ("energy_input", "energy_0to1", "energy"),
    ("energy_input_aug", "energy_0to1_aug", "energy"),
)


def process_meta(meta_path):
    """utils.py:87-95: `basename|text` lines."""
    names, texts = [], []
    with open(meta_path, "r", encoding="utf-8") as f:
        for line in f.readlines():
            n, t = line.strip("\n").split("|")
            names.append(n)
            texts.append(t)
    return names, texts


class FeatureStore:
    """`dataset.Dataset` (dataset.py:74-129) over the preprocessed `.npy` feature store: same directory / file naming
    scheme, same item dict.  The text front end (`text.text_to_sequence`) is outside this path: pass it (or any
    `str -> int array` callable) as `tokenizer`."""

    def __init__(self, root, tokenizer, filename="train.txt", dataset="VCTK", sort=True):
        import os
        self.root, self.dataset, self.tokenizer, self.sort = root, dataset, tokenizer, sort
        self.basename, self.text = process_meta(os.path.join(root, filename))

    def __len__(self):
        return len(self.text)

    def _load(self, sub, tag, name):
        import os
        return np.load(os.path.join(self.root, sub, "{}-{}-{}.npy".format(self.dataset, tag, name)))

    def __getitem__(self, idx):
        basename = self.basename[idx]
        item = {"id": basename, "text": np.array(self.tokenizer(self.text[idx])),
                "speaker_embed": self._load("spker_embed", "spker_embed", str(basename.split("_")[0]))}
        for key, sub, tag in _STORE_FILES:
            item[key] = self._load(sub, tag, basename)
        return item

    def collate_fn(self, batch):
        return collate_fn(batch, self.sort)


class BatchFeeder:
    """The training loop's data side (train.py:28-30, 99-132) for one rank: `batch_size^2` items per group (shuffled
    order, last partial group dropped, as DataLoader(shuffle=True, drop_last=True)), groups dealt round-robin to the
    ranks, each group collated into `batch_size` sorted sub-batches; a background thread reads and collates `depth`
    sub-batches ahead and stages them through pinned memory on a copy stream, so the consumer never waits on `np.load`,
    padding or the H2D copies.  Iterating yields `(tensors, max_src_len, max_mel_len)` like `to_device`."""

    def __init__(self, store, device, batch_size=None, rank=0, world=1, shuffle=True, seed=0, depth=4, bucket=None,
                 pairs=None):
        self.store, self.device = store, torch.device(device)
        self.batch_size = hp.batch_size if batch_size is None else batch_size
        self.rank, self.world, self.shuffle, self.seed, self.depth = rank, world, shuffle, seed, depth
        # (s_step, t_step): pad every sub-batch up to multiples of these (None: the reference's exact maxima).  The real feed
        # yields a different (S, T) for nearly every sub-batch; a graphed step is captured per shape, so without buckets it
        # would be captured once and never replayed (training.GraphedStepCache)
        self.bucket = bucket
        if pairs is None:                          # default: whatever the training step consumes (rt.pair_audio) -- a feed
            from .runtime import rt                # without the stacked layout costs the graphed step 12 extra copies
            pairs = bool(rt.pair_audio)
        self.pairs = pairs                         # also collate the stacked AudioEncoder inputs (to_device)
        self.epoch = 0

    def groups(self):
        """Item indices of this rank's groups for the current epoch (every rank derives the same permutation)."""
        n, g = len(self.store), self.batch_size ** 2
        order = np.random.RandomState(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        full = [order[i * g:(i + 1) * g] for i in range(n // g)]
        full = full[:len(full) // self.world * self.world]     # every rank runs the same number of steps per epoch (each
        return full[self.rank::self.world]                     # step ends in a collective): the remainder is dropped

    def __len__(self):
        return len(self.groups()) * self.batch_size

    def _produce(self, q, stop):
        use_cuda = self.device.type == "cuda"
        copy_stream = torch.cuda.Stream(device=self.device) if use_cuda else None
        try:
            for group in self.groups():
                for sub in self.store.collate_fn([self.store[int(i)] for i in group]):
                    if stop.is_set():
                        return
                    if use_cuda:
                        with torch.cuda.stream(copy_stream):
                            out = to_device(sub, self.device, pinned=True, bucket=self.bucket, pairs=self.pairs)
                            ready = torch.cuda.Event()
                            ready.record(copy_stream)
                    else:
                        out, ready = to_device(sub, self.device, pinned=False, bucket=self.bucket, pairs=self.pairs), None
                    q.put((out, ready))
            q.put(None)
        except BaseException as e:                     # surface reader errors in the consumer
            q.put(e)

    def __iter__(self):
        import queue
        import threading
        q, stop = queue.Queue(maxsize=self.depth), threading.Event()
        worker = threading.Thread(target=self._produce, args=(q, stop), daemon=True)
        worker.start()
        try:
            while True:
                got = q.get()
                if got is None:
                    break
                if isinstance(got, BaseException):
                    raise got
                (tensors, s, t), ready = got
                if ready is not None:
                    torch.cuda.current_stream(self.device).wait_event(ready)
                    for v in tensors.values():         # allocated on the copy stream, consumed on this one
                        v.record_stream(torch.cuda.current_stream(self.device))
                yield tensors, s, t
        finally:
            stop.set()
            while worker.is_alive():                   # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    worker.join(timeout=0.05)
            self.epoch += 1

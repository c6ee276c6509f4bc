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


_DTYPES = {"text": torch.long, "D": torch.long, "src_len": torch.long, "mel_len": torch.long}


def to_device(sub_batch, device, pinned=True):
    """train.py:107-132 in one shot: numpy -> pinned host tensors -> non-blocking H2D on the current stream.
    Returns (tensors dict, max_src_len, max_mel_len)."""
    out = {}
    for k, v in sub_batch.items():
        if k == "id":
            continue
        t = torch.from_numpy(np.ascontiguousarray(v)).to(_DTYPES.get(k, torch.float32))
        if pinned and torch.cuda.is_available():
            t = t.pin_memory()
        out[k] = t.to(device, non_blocking=True)
    return out, int(np.max(sub_batch["src_len"])), int(np.max(sub_batch["mel_len"]))

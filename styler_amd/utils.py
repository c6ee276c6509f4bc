"""Tensor helpers on the path (reference utils.py:223-232, 351-384, 417-429), HIP-backed."""
import torch

from . import ops


def get_mask_from_lengths(lengths, max_len=None):
    """utils.py:223-232: True = padding.  `max_len=None` needs max(lengths) on the host, as the
    reference's `.item()` does."""
    if max_len is None:
        max_len = int(torch.max(lengths).item())
    return ops.length_mask(lengths.contiguous(), int(max_len))


def get_scale(src, tgt):
    """utils.py:351-352."""
    return [src // tgt + (1 if x < src % tgt else 0) for x in range(tgt)]


def mel_calibrator(mel, mel_len, seq_len):
    """utils.py:355-384."""
    S = int(seq_len.max().item())
    return ops.mel_calibrate(mel, mel_len.contiguous(), seq_len.contiguous(), S)

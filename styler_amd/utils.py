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


def get_vocoder(checkpoint=None, device="cuda"):
    """utils.py:235-273, the `hp.vocoder == "HiFi-GAN"` branch: Generator(hifigan/config.json), optionally
    `checkpoint["generator"]` (the released generator_*.pth.tar files, absent here: random init), eval,
    remove_weight_norm, to(device)."""
    from . import hifigan
    vocoder = hifigan.Generator(hifigan.config_v1())
    if checkpoint is not None:
        ckpt = torch.load(checkpoint, map_location="cpu") if isinstance(checkpoint, str) else checkpoint
        vocoder.load_state_dict(ckpt["generator"] if "generator" in ckpt else ckpt)
    vocoder.eval()
    vocoder.remove_weight_norm()
    return vocoder.to(device)


def vocoder_infer(mel, vocoder, path=None):
    """utils.py:276-293 (HiFi-GAN branch): mel [B, 80, T] (or [80, T]) -> int16 waveform, written to `path` as a
    22.05 kHz wav when given."""
    from . import hparams as hp
    with torch.no_grad():
        wav = vocoder(mel).squeeze(1)
    wav = (wav.squeeze().cpu().numpy() * hp.max_wav_value).astype("int16")
    if path is not None:
        from scipy.io import wavfile
        wavfile.write(path, hp.sampling_rate, wav)
    return wav

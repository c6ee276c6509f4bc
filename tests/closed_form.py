"""Closed-form (seed-free, platform-exact) weights for parity fixtures.

Every tensor of a reference-format STYLER state dict is overwritten with a deterministic
function of (crc32(key), flat index) -- a splitmix64 integer hash mapped to [-1, 1) in
float64 -- so the golden generator (tests/golden/make_golden.py, which imports the
reference in the build container) and the tests on the GPU box regenerate bit-identical
weights and the 120 MB of weights never need committing.
"""
import zlib

import numpy as np
import torch

_KEEP = ("position_enc", "pitch_bins", "energy_bins", "num_batches_tracked")


def hash_uniform(seed: int, n: int) -> np.ndarray:
    """n doubles in [-1, 1), exact integer arithmetic (splitmix64 finaliser)."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 52) - 1.0


def closed_form_tensor(key: str, ref: torch.Tensor) -> torch.Tensor:
    if any(k in key for k in _KEEP):
        return ref.clone()
    u = hash_uniform(zlib.crc32(key.encode()), ref.numel()).reshape(tuple(ref.shape))
    if key == "style_modeling.duration_predictor.linear_layer.bias":
        v = np.full(tuple(ref.shape), 1.3)             # free-running durations ~ e^1.3 - 1 > 0
    elif key.endswith(".weight_g"):                      # weight-norm gains of the vocoder (hifigan/models.py)
        v = 0.5 + 0.25 * u
    elif key.endswith("running_var"):
        v = 0.6 + 0.4 * (u + 1.0)                      # [0.6, 1.4)
    elif key.endswith("running_mean"):
        v = 0.1 * u
    elif ref.dim() == 1 and key.endswith(".weight"):     # LayerNorm / GroupNorm / BatchNorm gain
        v = 1.0 + 0.1 * u
    elif ref.dim() == 1:                                 # biases
        v = 0.1 * u
    elif "embedding" in key or "src_word_emb" in key:
        v = 0.5 * u
    else:                                                # Linear / Conv1d / LSTM matrices
        fan_in = int(np.prod(ref.shape[1:]))
        v = np.sqrt(3.0 / fan_in) * u
    return torch.from_numpy(np.ascontiguousarray(v)).to(ref.dtype)


def closed_form_state_dict(ref_sd):
    return {k: closed_form_tensor(k, v) for k, v in ref_sd.items()}


def make_batch(B, s_lo, s_hi, d_lo, d_hi, seed=1234, fix_src=None, fix_mel=None):
    """Seeded synthetic VCTK-shape batch, zero beyond lengths."""
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi, shape: torch.randint(lo, hi + 1, shape, generator=g)
    src_len = ri(s_lo, s_hi, (B,)) if fix_src is None else torch.full((B,), fix_src)
    S = int(src_len.max())
    src_valid = torch.arange(S)[None] < src_len[:, None]
    D = ri(d_lo, d_hi, (B, S)) * src_valid
    if fix_mel is not None:
        for b in range(B):
            last = int(src_len[b]) - 1
            D[b, last] += fix_mel - int(D[b].sum())
            assert D[b, last] >= 0
    mel_len = D.sum(1)
    T = int(mel_len.max())
    mel_valid = (torch.arange(T)[None] < mel_len[:, None])
    mv = mel_valid.float()
    rn = lambda *s: torch.randn(*s, generator=g)
    ru = lambda *s: torch.rand(*s, generator=g)

    def unit(shape):
        x = ru(*shape)
        return x * (ru(*shape) >= 0.3).float() * mv

    spk = rn(B, 512)
    batch = dict(
        text=ri(1, 151, (B, S)) * src_valid,
        mel_target=rn(B, T, 80) * mv[..., None],
        mel_aug=rn(B, T, 80) * mv[..., None],
        D=D, log_D=torch.log(D.float() + 1.0),
        f0=(80.0 + 300.0 * ru(B, T)) * mv,
        f0_norm=unit((B, T)), f0_norm_aug=unit((B, T)),
        energy=100.0 * ru(B, T) * mv,
        energy_input=unit((B, T)), energy_input_aug=unit((B, T)),
        speaker_embed=spk / spk.norm(dim=1, keepdim=True),
        src_len=src_len, mel_len=mel_len)
    return batch

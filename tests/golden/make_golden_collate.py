"""Golden for the collate (dataset.py:116-207): imports the reference Dataset in the build container with its wav /
alignment dependencies stubbed (they are only used by wav helpers), feeds a seeded synthetic item list, stores the
padded sub-batches.  Run: python tests/golden/make_golden_collate.py"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def synthetic_items(seed=5, n=16):
    rng = np.random.RandomState(seed)
    items = []
    for i in range(n):
        s = int(rng.randint(3, 12))
        D = rng.randint(1, 5, size=s)
        t = int(D.sum())
        items.append({"id": f"p{i:03d}_x", "text": rng.randint(1, 150, size=s),
                      "mel_target": rng.randn(t, 80).astype(np.float32), "mel_aug": rng.randn(t, 80).astype(np.float32),
                      "D": D, "f0": rng.rand(t) * 300, "f0_norm": rng.rand(t), "f0_norm_aug": rng.rand(t),
                      "energy": rng.rand(t) * 100, "energy_input": rng.rand(t), "energy_input_aug": rng.rand(t),
                      "speaker_embed": rng.randn(1, 512).astype(np.float32)})
    return items


def main():
    for name in ("unidecode", "inflect", "tgt", "pyworld", "pysptk", "librosa", "librosa.util", "librosa.filters"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["unidecode"].unidecode = lambda s: s
    sys.modules["inflect"].engine = type("E", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["pysptk"].sptk = types.ModuleType("sptk")
    lib = sys.modules["librosa"]
    lib.util, lib.filters = sys.modules["librosa.util"], sys.modules["librosa.filters"]
    lib.util.pad_center = lambda w, n: w
    lib.util.tiny = lambda x: 1e-30
    from oracle import styler_oracle as O
    lib.filters.mel = lambda sr, n_fft, n_mels, fmin, fmax: O.mel_filterbank(sr, n_fft, n_mels, fmin, fmax).numpy()
    sys.path.insert(0, "/root/reference")
    os.chdir("/tmp")
    import dataset as ref_dataset
    ds = ref_dataset.Dataset.__new__(ref_dataset.Dataset)
    ds.sort = True
    out = ds.collate_fn(synthetic_items())
    save = {}
    for k, v in enumerate(out):
        for kk, vv in v.items():
            save[f"b{k}:{kk}"] = np.array(vv)
    np.savez_compressed(os.path.join(HERE, "collate.npz"), **save)
    print(len(out), list(out[0].keys()), out[0]["src_len"])


if __name__ == "__main__":
    main()

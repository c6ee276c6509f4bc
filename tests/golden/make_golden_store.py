"""Golden for the `.npy` feature-store reader (dataset.py:74-129): writes a seeded synthetic store under /tmp in the
reference's directory layout, reads it with the REFERENCE `dataset.Dataset` (wav / alignment dependencies stubbed as in
make_golden_collate.py; `text_to_sequence` replaced by the same whitespace-int tokenizer the test passes to the build's
reader, the text front end being out of scope) and stores what it returned.  Run: python tests/golden/make_golden_store.py"""
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def tokenizer(text):
    return [int(t) for t in text.split()]


def write_store(root, seed=11, n=20, dataset="VCTK"):
    """Synthetic store: `n` utterances of 3 speakers, file names as dataset.py:90-118."""
    from make_golden_collate import synthetic_items
    items = synthetic_items(seed=seed, n=n)
    subs = {"mel_target": ("mel_clean", "mel"), "mel_aug": ("mel_aug", "mel"), "D": ("alignment", "ali"),
            "f0": ("f0", "f0"), "f0_norm": ("f0_norm", "f0"), "f0_norm_aug": ("f0_norm_aug", "f0"),
            "energy": ("energy", "energy"), "energy_input": ("energy_0to1", "energy"),
            "energy_input_aug": ("energy_0to1_aug", "energy")}
    for sub, _ in list(subs.values()) + [("spker_embed", None)]:
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    lines = []
    for i, it in enumerate(items):
        spk = "p%03d" % (225 + i % 3)
        base = "%s_%03d" % (spk, i)
        for key, (sub, tag) in subs.items():
            np.save(os.path.join(root, sub, "{}-{}-{}.npy".format(dataset, tag, base)), it[key])
        np.save(os.path.join(root, "spker_embed", "{}-spker_embed-{}.npy".format(dataset, spk)),
                items[i % 3]["speaker_embed"])
        lines.append(base + "|" + " ".join(str(int(t)) for t in it["text"]))
    with open(os.path.join(root, "train.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")


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
    root = "/tmp/styler_store_golden"
    shutil.rmtree(root, ignore_errors=True)
    write_store(root)
    sys.path.insert(0, "/root/reference")
    os.chdir("/tmp")
    import hparams as ref_hp
    import dataset as ref_dataset
    ref_hp.preprocessed_path, ref_hp.dataset = root, "VCTK"
    ref_dataset.text_to_sequence = lambda text, cleaners: tokenizer(text)
    ds = ref_dataset.Dataset("train.txt")
    save = {"n": np.array(len(ds))}
    for idx in (0, 7, 19):
        for k, v in ds[idx].items():
            save[f"item{idx}_{k}"] = np.array(v)
    subs = ds.collate_fn([ds[i] for i in range(16)])
    save["n_sub"] = np.array(len(subs))
    for j in (0, 3):
        for k, v in subs[j].items():
            save[f"sub{j}_{k}"] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "store.npz"), **save)
    print(len(save), "arrays", os.path.getsize(os.path.join(HERE, "store.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()

"""Golden for `StyleModeling.predict_inference` (modules.py:285-309, the synthesize.py inspection / control entry point):
imports the reference model (text-only dependencies stubbed as in make_golden.py), closed-form weights, seeded
encodings; stores inputs + the reference's 9-tuple for two control settings.
Run in the build container: python tests/golden/make_golden_inference.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

NAMES = ("text", "pitch_emb", "speaker", "energy_emb", "noise", "log_d", "p_pred", "e_pred", "mel_mask")
CASES = {"a": dict(speaker_normalized=True, d_control=1.0, p_control=1.0, e_control=1.0),
         "b": dict(speaker_normalized=False, d_control=1.3, p_control=1.2, e_control=0.8)}


def inputs(B=2, S=9):
    from closed_form import hash_uniform
    enc = {k: torch.from_numpy(hash_uniform(500 + i, B * S * 256).reshape(B, S, 256)).float()
           for i, k in enumerate(("text", "pitch", "energy", "duration", "speaker", "noise"))}
    src_len = torch.tensor([S, S - 3])
    src_mask = torch.arange(S)[None] >= src_len[:, None]
    for k in enc:                                           # padded phonemes carry zeros, as after the masked encoder
        enc[k] = enc[k] * (~src_mask)[..., None]
    return enc, src_mask


def main():
    from make_golden import _import_reference
    from closed_form import closed_form_state_dict
    styler, modules, loss, optimizer, utils = _import_reference()
    torch.manual_seed(0)
    model = styler.STYLER().eval()
    model.load_state_dict(closed_form_state_dict(model.state_dict()))
    sm = model.style_modeling
    enc, src_mask = inputs()
    save = {"src_mask": src_mask.numpy()}
    save.update({"in_" + k: v.numpy() for k, v in enc.items()})
    with torch.no_grad():
        for tag, kw in CASES.items():
            out = sm.predict_inference(enc["text"], enc["pitch"], enc["energy"], enc["duration"], enc["speaker"],
                                       enc["noise"], src_mask, None, **kw)
            for n, v in zip(NAMES, out):
                save[f"{tag}_{n}"] = v.numpy()
            print(tag, {n: tuple(v.shape) for n, v in zip(NAMES, out)})
    np.savez_compressed(os.path.join(HERE, "predict_inference.npz"), **save)
    print(os.path.getsize(os.path.join(HERE, "predict_inference.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()

"""Golden vectors for the HiFi-GAN generator from the REFERENCE itself.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_hifigan.py

Imports `hifigan.Generator` (hifigan/models.py:112-173) with the repository's hifigan/config.json values, overwrites
every weight (weight_g / weight_v / bias, i.e. the checkpoint format before remove_weight_norm) with the closed-form
generator of tests/closed_form.py and stores inputs + reference outputs only: the waveform, the pre-tanh signal and the
input of every upsampling layer (captured with forward hooks), so a mismatch can be located per stage.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"


def main():
    sys.path.insert(0, REF)
    import hifigan
    from closed_form import closed_form_state_dict, hash_uniform

    with open(os.path.join(REF, "hifigan", "config.json")) as f:
        h = hifigan.AttrDict(json.load(f))
    torch.manual_seed(0)
    gen = hifigan.Generator(h).eval()
    gen.load_state_dict(closed_form_state_dict(gen.state_dict()))

    B, T = 2, 7                                   # 7*8 = 56 frames after the first stage: ragged dilation phases
    mel = torch.from_numpy(hash_uniform(20260927, B * 80 * T).reshape(B, 80, T) * 3.0 - 4.0).float()

    taps = {}
    for i, up in enumerate(gen.ups):
        up.register_forward_pre_hook(lambda m, inp, i=i: taps.__setitem__(f"ups_in_{i}", inp[0].detach().clone()))
    gen.conv_post.register_forward_hook(lambda m, inp, out: taps.__setitem__("pre_tanh", out.detach().clone()))
    with torch.no_grad():
        wav = gen(mel)
    out = {"mel": mel.numpy(), "wav": wav.numpy()}
    out.update({k: v.numpy() for k, v in taps.items()})
    path = os.path.join(HERE, "hifigan.npz")
    np.savez_compressed(path, **out)
    print({k: (v.shape, float(np.abs(v).max())) for k, v in out.items()}, os.path.getsize(path) // 1024, "KiB")
    shapes = {k: list(v.shape) for k, v in gen.state_dict().items()}
    with open(os.path.join(HERE, "hifigan_state_dict_shapes.json"), "w") as f:
        json.dump(shapes, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_addoption(parser):
    parser.addoption("--shuffle", type=int, default=None, metavar="SEED",
                     help="run the collected tests in a seeded random order (order-dependence screen; the default order is "
                          "the file order: oracle / golden parity, then the train step + optimiser + two-rank tests, "
                          "then the stand-alone backward kernels, HIP-vs-HIP equivalences last)")


def pytest_collection_modifyitems(config, items):
    seed = config.getoption("--shuffle")
    if seed is not None:
        import random
        random.Random(seed).shuffle(items)


@pytest.fixture(autouse=True)
def _seed_global_rngs(request):
    """Every test starts from global torch / numpy / python RNG states derived from ITS node id: `nn.*` constructors,
    `torch.randn` without a generator and numpy's legacy global draw the same numbers in every process and in every
    collection order (the suite used to test different weights in every process).  The seed is
    printed on failure (`-rA` / the assertion section) through the `record_property` channel."""
    import random
    import zlib
    import numpy as np
    import torch
    seed = zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    request.node.user_properties.append(("rng_seed", seed))
    yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    return load


def _reference_state_dict():
    """Reference-format state dict with closed-form weights (tests/closed_form.py), built
    from the committed key/shape table of the reference's `STYLER().state_dict()`
    (tests/golden/state_dict_shapes.json); the three formula-defined buffers come from
    their defining formulas (Models.py:11-30, modules.py:278-281)."""
    import json
    import numpy as np
    import torch
    from closed_form import closed_form_tensor
    from oracle import styler_oracle as O
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_shapes.json")) as f:
        table = json.load(f)
    sd = {}
    for k, (shape, dtype) in table.items():
        if "position_enc" in k:
            sd[k] = O.sinusoid_table(1001, 256)[None]
        elif k.endswith("pitch_bins"):
            sd[k] = torch.exp(torch.linspace(np.log(71.0), np.log(797.9), 255))
        elif k.endswith("energy_bins"):
            sd[k] = torch.linspace(0.1, 525.43, 255)
        else:
            sd[k] = closed_form_tensor(k, torch.zeros(shape, dtype=getattr(torch, dtype)))
    return sd


@pytest.fixture(scope="session")
def ref_state_dict():
    return _reference_state_dict()


def _hifigan_state_dict():
    """Weight-normed generator checkpoint (weight_g / weight_v / bias) with closed-form weights, from the committed
    key/shape table of the reference's `hifigan.Generator(config.json).state_dict()`."""
    import json
    import torch
    from closed_form import closed_form_tensor
    with open(os.path.join(ROOT, "tests", "golden", "hifigan_state_dict_shapes.json")) as f:
        table = json.load(f)
    return {k: closed_form_tensor(k, torch.zeros(shape)) for k, shape in table.items()}


@pytest.fixture(scope="session")
def hifigan_state_dict():
    return _hifigan_state_dict()

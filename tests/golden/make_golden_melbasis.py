"""Second, independent derivation of the 80-band Slaney mel filterbank of audio/stft.py:141-143 (librosa_mel_fn(22050, 1024, 80,
0, 8000)): `transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` (HF transformers, an implementation
written against librosa's behaviour and independent of oracle/styler_oracle.py).  librosa itself is absent from the
reference tree and from this image, so this is a cross-check, not a pin against librosa 0.7.2.

Writes tests/golden/mel_filterbank_hf.npz: the full [80, 513] float64 matrix (330 KB) + the version that produced it."""
import os

import numpy as np
import transformers
from transformers.audio_utils import mel_filter_bank

fb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0,
                     sampling_rate=22050, norm="slaney", mel_scale="slaney").T.astype(np.float64)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "mel_filterbank_hf.npz"), mel_basis=fb,
                    transformers_version=np.array(transformers.__version__))
print(fb.shape, transformers.__version__)

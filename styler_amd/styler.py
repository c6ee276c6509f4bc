"""`styler.STYLER` drop-in (reference styler.py:13-58): same constructor, forward signature, 9-tuple
return, `.decode`, attribute tree and state-dict layout; all arithmetic in libstyler_hip.so."""
import torch
import torch.nn as nn

from . import hparams as hp
from . import ops
from .modules import StyleModeling
from .runtime import rt
from .transformer import Decoder, PostNet, _HipModule
from .utils import get_mask_from_lengths


class STYLER(_HipModule):
    """STYLER.  Extra knobs (not in the reference): `clean_only` skips the second (noisy) decode of
    styler.py:55 and returns the clean outputs in both slots (BASELINE config 2)."""

    def __init__(self, use_postnet=True):
        super().__init__()
        self.style_modeling = StyleModeling()
        self.decoder = Decoder()
        self.mel_linear = nn.Linear(hp.decoder_hidden, hp.n_mel_channels)
        self.use_postnet = use_postnet
        if self.use_postnet:
            self.postnet = PostNet()
        self.clean_only = False

    def _lens_from_mask(self, mel_mask):
        return (~mel_mask).sum(dim=1).to(torch.int64)

    def decode(self, style_modeling_output, mel_mask, mel_len=None):
        """styler.py:29-37.  `mel_mask` True = padding; pass `mel_len` to skip deriving it."""
        lens = mel_len if mel_len is not None else self._lens_from_mask(mel_mask)
        decoder_output = self.decoder(style_modeling_output, lens)
        mel_output = self._gemm("mel_linear", decoder_output, self.mel_linear)
        if self.use_postnet:
            mel_output_postnet = self.postnet(mel_output, add_residual=mel_output)
        else:
            mel_output_postnet = mel_output
        return mel_output, mel_output_postnet

    def forward(self, src_seq, mel_target, mel_aug, p_norm, e_input, src_len, mel_len, d_target=None,
                p_target=None, e_target=None, max_src_len=None, max_mel_len=None, speaker_embed=None,
                d_control=1.0, p_control=1.0, e_control=1.0):
        if not src_seq.is_cuda:
            raise RuntimeError("styler_amd.STYLER runs on the MI355X HIP path only (no CPU fallback)")
        src_len = src_len.contiguous()
        mel_len = mel_len.contiguous()
        src_mask = get_mask_from_lengths(src_len, max_src_len if max_src_len is not None else src_seq.shape[1])
        mel_mask = get_mask_from_lengths(mel_len, max_mel_len if max_mel_len is not None else None)
        max_mel_len = None if max_mel_len is None else int(max_mel_len)

        (style_modeling_output, noise_encoding, d_prediction, p_prediction, e_prediction, new_len, new_mask,
         aug) = self.style_modeling(
            src_seq.contiguous(), speaker_embed.contiguous(), mel_target, mel_aug, p_norm, e_input, src_len,
            mel_len, src_mask, mel_mask, d_target, p_target, e_target, max_mel_len, d_control, p_control,
            e_control)
        if d_target is None:
            mel_len, mel_mask = new_len, new_mask

        mel_output, mel_output_postnet = self.decode(style_modeling_output, mel_mask, mel_len)
        if self.clean_only:
            mel_output_noisy, mel_output_postnet_noisy = mel_output, mel_output_postnet
        else:
            # styler.py:55: decode(style_modeling_output.detach() + noise_encoding); the sum was produced by
            # the bucketise/embed/add kernel in the same pass
            mel_output_noisy, mel_output_postnet_noisy = self.decode(self.style_modeling._out_noisy, mel_mask,
                                                                     mel_len)
        return ((mel_output, mel_output_noisy), (mel_output_postnet, mel_output_postnet_noisy), d_prediction,
                p_prediction, e_prediction, src_mask, mel_mask, mel_len, aug)

"""`styler.STYLER` drop-in (reference styler.py:13-58): same constructor, forward signature, 9-tuple
return, `.decode`, attribute tree and state-dict layout; all arithmetic in libstyler_hip.so."""
import torch
import torch.nn as nn

from . import autograd as AG
from . import hparams as hp
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

    @property
    def module(self):
        """`nn.DataParallel(STYLER()).module` of the reference's callers (train.py:33,38-43,149-153; synthesize.py:62-63,116-128,
        202,313; evaluate.py:20,97-101).  Data parallelism here is one process per GPU (styler_amd.dist), so the model is its
        own replica: `model.module` is the model.  (Wrapping in a real `nn.DataParallel` works as well -- with the one visible
        device of a rank it forwards straight to the module.)"""
        return self

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        """Also accepts the keys of a reference checkpoint, which carry the DataParallel wrapper's `module.` prefix
        (train.py:54,222; synthesize.py:63: `model.load_state_dict(checkpoint['model'])`)."""
        if state_dict and all(k.startswith("module.") for k in state_dict):
            state_dict = type(state_dict)((k[len("module."):], v) for k, v in state_dict.items())
        return super().load_state_dict(state_dict, strict=strict, **kwargs)

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

    def decode_pair(self, out_clean, out_noisy, mel_mask, mel_len=None):
        """`decode(out_clean)` and `decode(out_noisy)` (styler.py:52,55) with the Decoder and mel_linear run once on the
        stacked batch; PostNet stays per branch (its BatchNorm statistics are per call, Layers.py:126)."""
        lens = mel_len if mel_len is not None else self._lens_from_mask(mel_mask)
        B = out_clean.shape[0]
        mel2 = self._gemm("mel_linear", self.decoder.forward_pair(out_clean, out_noisy, lens), self.mel_linear)
        tape = self.training and torch.is_grad_enabled() and mel2.requires_grad
        split = (lambda t: AG.SplitBatchFn.apply(t)) if tape else (lambda t: (t[:B], t[B:]))
        if self.use_postnet and rt.pair_postnet:
            # both branches through the PostNet as one batch of 2B items, BatchNorm statistics per branch (segs = 2)
            post2 = self.postnet(mel2, add_residual=mel2, segs=2)
            return list(zip(split(mel2), split(post2)))
        outs = []
        for mel in split(mel2):
            outs.append((mel, self.postnet(mel, add_residual=mel) if self.use_postnet else mel))
        return outs

    def forward(self, src_seq, mel_target, mel_aug, p_norm, e_input, src_len, mel_len, d_target=None,
                p_target=None, e_target=None, max_src_len=None, max_mel_len=None, speaker_embed=None,
                d_control=1.0, p_control=1.0, e_control=1.0):
        if not src_seq.is_cuda:
            raise RuntimeError("styler_amd.STYLER runs on the MI355X HIP path only (no CPU fallback)")
        src_len = src_len.contiguous()
        mel_len = mel_len.contiguous()
        if max_mel_len is not None:                    # both extents known on the host: the two masks in one launch
            from . import ops
            src_mask, mel_mask = ops.length_mask2(src_len, int(max_src_len) if max_src_len is not None else src_seq.shape[1],
                                                  mel_len, int(max_mel_len))
        else:
            src_mask = get_mask_from_lengths(src_len, max_src_len if max_src_len is not None else src_seq.shape[1])
            mel_mask = get_mask_from_lengths(mel_len, None)
        max_mel_len = None if max_mel_len is None else int(max_mel_len)

        (style_modeling_output, noise_encoding, d_prediction, p_prediction, e_prediction, new_len, new_mask,
         aug) = self.style_modeling(
            src_seq.contiguous(), speaker_embed.contiguous(), mel_target, mel_aug, p_norm, e_input, src_len,
            mel_len, src_mask, mel_mask, d_target, p_target, e_target, max_mel_len, d_control, p_control,
            e_control)
        if d_target is None:
            mel_len, mel_mask = new_len, new_mask

        noisy_in = self.style_modeling._out_noisy
        if self.clean_only:
            mel_output, mel_output_postnet = self.decode(style_modeling_output, mel_mask, mel_len)
            mel_output_noisy, mel_output_postnet_noisy = mel_output, mel_output_postnet
        elif (rt.pair_decodes and rt.pack_decoder and 2 * src_seq.shape[0] <= 4096
              and noisy_in.shape == style_modeling_output.shape):
            # styler.py:52,55: decode(x) and decode(x.detach() + noise_encoding) as one stacked batch; the second input
            # was produced by the bucketise/embed/add kernel in the same pass
            (mel_output, mel_output_postnet), (mel_output_noisy, mel_output_postnet_noisy) = self.decode_pair(
                style_modeling_output, noisy_in, mel_mask, mel_len)
        else:
            mel_output, mel_output_postnet = self.decode(style_modeling_output, mel_mask, mel_len)
            mel_output_noisy, mel_output_postnet_noisy = self.decode(noisy_in, mel_mask, mel_len)
        side = getattr(self.style_modeling, "_pred_side", None)
        if side is not None:                          # rt.pred_stream: the predictors ran next to the decode
            torch.cuda.current_stream().wait_stream(side)
            self.style_modeling._pred_side = None
        return ((mel_output, mel_output_noisy), (mel_output_postnet, mel_output_postnet_noisy), d_prediction,
                p_prediction, e_prediction, src_mask, mel_mask, mel_len, aug)

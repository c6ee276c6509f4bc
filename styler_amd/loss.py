"""`loss.STYLERLoss` / `DomainAdversarialTrainingLoss` drop-ins (reference loss.py:7-68).

masked_select + MSELoss / L1Loss become one masked-reduction kernel per term (no compacted copies):
sum over valid positions / count, accumulated in fp64.  Masks arrive as the reference passes them
(True = valid, i.e. the caller's `~src_mask`, `~mel_mask`); the kernels take lengths, so `src_len` /
`mel_len` (already part of the reference signature) are what is actually consumed."""
import torch
import torch.nn as nn

from . import autograd as AG
from . import ops


def _masked_mean(a, b, kind, lens):
    if torch.is_grad_enabled() and a.requires_grad:
        return AG.MaskedErrFn.apply(a, b.detach(), kind, lens)
    return ops.masked_err_mean(a.contiguous(), b.contiguous(), kind, lens)[0].view(())


def _masked_means(terms):
    """[(a, b, kind, lens), ...] -> tuple of scalar means: ONE launch for all terms of a loss call (and one in backward)."""
    if torch.is_grad_enabled() and all(a.requires_grad for a, _, _, _ in terms):
        flat = []
        for a, b, _, _ in terms:
            flat += [a, b.detach()]
        return AG.MaskedErrMultiFn.apply(tuple(k for _, _, k, _ in terms), tuple(l for _, _, _, l in terms), *flat)
    if any(a.requires_grad for a, _, _, _ in terms) and torch.is_grad_enabled():
        return tuple(_masked_mean(a, b, k, l) for a, b, k, l in terms)
    means, _ = ops.masked_err_mean_multi([(a.contiguous(), b.contiguous(), k, l) for a, b, k, l in terms])
    return tuple(m.view(()) for m in means)


def _nll3(posteriors, label):
    """3 x NLLLoss(mean) on [B, 2] log-probabilities, summed (loss.py:46-48): one kernel.  `label`: int64 [B] tensor as in
    the reference call, or the python int 0 / 1 when every label is the same (train.py:139,152 builds zeros / ones)."""
    if torch.is_tensor(label):
        label = label.contiguous()
    if torch.is_grad_enabled() and any(p.requires_grad for p in posteriors):
        return AG.Nll3Fn.apply(posteriors[0], posteriors[1], posteriors[2], label)
    return ops.nll3([p.contiguous() for p in posteriors], label).view(())


class STYLERLoss(nn.Module):
    """loss.py:7-50."""

    def cal_mel_loss(self, mel, mel_postnet, mel_target, mel_mask, mel_len=None):
        lens = mel_len if mel_len is not None else mel_mask.sum(dim=1).to(torch.int64)
        return _masked_means([(mel, mel_target, 0, lens), (mel_postnet, mel_target, 0, lens)])

    def forward(self, log_d_predicted, log_d_target, p_predicted, p_target, e_predicted, e_target, mel, mel_postnet,
                mel_target, src_mask, mel_mask, src_len, mel_len, aug_posteriors, aug_label):
        lens = mel_len if mel_len is not None else mel_mask.sum(dim=1).to(torch.int64)
        slens = src_len if src_len is not None else src_mask.sum(dim=1).to(torch.int64)
        mel_loss, mel_postnet_loss, d_loss, p_loss, e_loss = _masked_means([
            (mel, mel_target, 0, lens), (mel_postnet, mel_target, 0, lens), (log_d_predicted, log_d_target, 1, slens),
            (p_predicted, p_target, 1, lens), (e_predicted, e_target, 1, lens)])
        return mel_loss, mel_postnet_loss, d_loss, p_loss, e_loss, _nll3(aug_posteriors, aug_label)


class DomainAdversarialTrainingLoss(nn.Module):
    """loss.py:53-68."""

    def forward(self, augmentation_posterior, aug_label):
        return _nll3(augmentation_posterior, aug_label)

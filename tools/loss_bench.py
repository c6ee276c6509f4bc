#!/usr/bin/env python
"""The seven masked-error means of a train step's loss head (one launch) and their backward, at the bench shape, cache-cold."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
dev = torch.device("cuda")
B, T, S = 48, 441, 60
NS = 8
g = torch.Generator().manual_seed(0)
mel_len = torch.randint(150, 442, (B,), generator=g).to(dev)
src_len = torch.randint(20, 61, (B,), generator=g).to(dev)
def mk():
    r = lambda *s: torch.randn(*s, device=dev)
    return [(r(B, T, 80), r(B, T, 80), 0, mel_len), (r(B, T, 80), r(B, T, 80), 0, mel_len), (r(B, S), r(B, S), 1, src_len),
            (r(B, T), r(B, T), 1, mel_len), (r(B, T), r(B, T), 1, mel_len), (r(B, T, 80), r(B, T, 80), 0, mel_len),
            (r(B, T, 80), r(B, T, 80), 0, mel_len)]
sets = [mk() for _ in range(NS)]
def t(fn, n=40):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % NS)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
res = [None] * NS
def fwd(i): res[i] = ops.masked_err_mean_multi(sets[i])
print(f"masked_err_mean_multi (7 terms, incl. 7 torch.zeros for the accumulators): {t(fwd):6.1f} us")
gs = torch.ones(1, device=dev)
def bwd(i):
    means, accs = res[i]
    ops.masked_err_bwd_multi([(a, b, acc, gs, k, l) for (a, b, k, l), acc in zip(sets[i], accs)])
for i in range(NS): fwd(i)
print(f"masked_err_bwd_multi: {t(bwd):6.1f} us")
z = [torch.zeros(ops.MASKED_ACC_DOUBLES, dtype=torch.float64, device=dev) for _ in range(7)]
def zeros7(i):
    for k in range(7): torch.zeros(ops.MASKED_ACC_DOUBLES, dtype=torch.float64, device=dev)
print(f"(7 torch.zeros alone: {t(zeros7):6.1f} us)")
# which terms cost what (forward kernel only, accumulators preallocated once and re-zeroed by ONE fill)
import ctypes
from styler_amd._lib import MaskedTerm
def run_terms(idx, label):
    accs = torch.zeros(len(idx), ops.MASKED_ACC_DOUBLES, dtype=torch.float64, device=dev)
    means = torch.empty(len(idx), device=dev)
    arrs = []
    for s_ in sets:
        arr = (MaskedTerm * len(idx))()
        for k, j in enumerate(idx):
            a, b, kind, lens = s_[j]
            m = arr[k]
            if a.dim() == 2: B_, L_, C_, lda, ldb = a.shape[0], a.shape[1], 1, 1, 1
            else: B_, L_, C_, lda, ldb = a.shape[0], a.shape[1], a.shape[2], a.shape[2], a.shape[2]
            m.a, m.b, m.acc, m.mean, m.len = a.data_ptr(), b.data_ptr(), accs[k].data_ptr(), means[k:k+1].data_ptr(), lens.data_ptr()
            m.lda, m.ldb, m.B, m.L, m.C, m.kind = lda, ldb, B_, L_, C_, kind
        arrs.append(arr)
    def f(i):
        accs.zero_()
        ops._chk(ops.lib.styler_masked_err_mean_multi(arrs[i], len(idx), ops._stream()), "x")
    print(f"{label:40s} {t(f):6.1f} us (incl. one fill)")
run_terms([0, 1, 5, 6], "four [48, 441, 80] terms")
run_terms([0], "one [48, 441, 80] term")
run_terms([2, 3, 4], "three small terms (C = 1)")
run_terms([3], "one [48, 441] term")
run_terms(list(range(7)), "all seven")

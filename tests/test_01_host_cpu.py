"""CPU-side tests: the C-ABI library loads and exports every declared symbol, the host mirror has the
reference's state-dict layout, the product never imports the oracle, and the multi-rank helpers work
under gloo with world_size 2."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "styler_hip.h")).read()
    declared = set(re.findall(r"^\s*int(?:64_t)?\s+(styler_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    from styler_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in include/styler_hip.h but not exported: {missing}"
    assert set(_lib.EXPORTED) == declared, set(_lib.EXPORTED) ^ declared
    assert _lib.ABI_VERSION == 1


def test_state_dict_layout_matches_reference_table():
    import json
    from styler_amd import STYLER
    sd = STYLER().state_dict()
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_shapes.json")))
    assert list(sd.keys()) == list(ref.keys())
    for k, (shape, dtype) in ref.items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == "torch." + dtype, k
    n_train = sum(p.numel() for p in STYLER().parameters() if p.requires_grad)
    assert n_train == 29482701


def test_dataparallel_call_sites_resolve_through_module():
    """every attribute path the reference's callers reach through `nn.DataParallel(STYLER()).module` (train.py:33,
    38-43,149-153; synthesize.py:62-63,116-128,171,195-197,202,313; evaluate.py:20,97-101) resolves on the bare model (its
    own `.module`) and on a real `nn.DataParallel` wrapper, and a `module.`-prefixed reference checkpoint loads into both."""
    import functools
    from styler_amd import STYLER
    paths = ["decode", "decoder", "style_modeling", "style_modeling.style_encoder.text_encoder",
             "style_modeling.style_encoder.audio_encoder", "style_modeling.style_encoder.encoder_input_cat",
             "style_modeling.style_encoder.speaker_linear", "style_modeling.style_encoder.speaker_linear_p",
             "style_modeling.duration_predictor", "style_modeling.pitch_predictor", "style_modeling.energy_predictor",
             "style_modeling.augmentation_classifier_d", "style_modeling.augmentation_classifier_p",
             "style_modeling.augmentation_classifier_e", "style_modeling.predict_inference", "style_modeling.pitch_linear",
             "style_modeling.length_regulator", "mel_linear", "postnet"]
    bare = STYLER()
    assert bare.module is bare and "module" not in dict(bare.named_children())
    wrapped = torch.nn.DataParallel(STYLER())
    for model in (bare, wrapped):
        for p in paths:
            obj = functools.reduce(getattr, p.split("."), model.module)
            assert callable(obj), p
        n = sum(q.numel() for q in model.module.decoder.parameters())      # utils.get_param_num (train.py:43)
        assert n > 0
    ref_sd = {"module." + k: torch.full_like(v, 3) if v.is_floating_point() else v.clone() for k, v in bare.state_dict().items()}
    assert len(ref_sd) == 328
    for model in (bare, wrapped):
        res = model.load_state_dict(ref_sd)                                 # synthesize.py:63 / train.py:54
        assert not res.missing_keys and not res.unexpected_keys
        assert float(model.module.mel_linear.weight.min()) == 3.0
    assert list(wrapped.state_dict()) == list(ref_sd)                       # train.py:222 writes these keys
    assert list(bare.state_dict(prefix="module.")) == list(ref_sd)
    bare.load_state_dict(bare.state_dict())                                 # bare keys still load


def test_product_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "styler_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|styler_oracle", txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_cpu_forward_fails_loudly():
    from styler_amd import STYLER
    m = STYLER().eval()
    t = torch.zeros(1, 4, dtype=torch.long)
    with pytest.raises(RuntimeError):
        m(t, torch.zeros(1, 8, 80), torch.zeros(1, 8, 80), torch.zeros(1, 8), torch.zeros(1, 8),
          torch.tensor([4]), torch.tensor([8]), speaker_embed=torch.zeros(1, 512))


def test_noam_schedule(golden):
    from styler_amd.optimizer import ScheduledOptim
    g = golden("noam_lr")
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], betas=(0.9, 0.98), eps=1e-9)
    for n, lr in zip(g["steps"], g["lr"]):
        so = ScheduledOptim(opt, 256, 4000, int(n) - 1)
        so._update_learning_rate()
        assert abs(opt.param_groups[0]["lr"] - float(lr)) < 1e-15 + 1e-12 * abs(lr)


def test_shard_indices_balanced():
    from styler_amd.dist import shard_indices
    lens = [5, 50, 7, 48, 20, 21, 3, 60]
    parts = [shard_indices(8, r, 2, lens) for r in range(2)]
    assert sorted(parts[0] + parts[1]) == list(range(8))
    loads = [sum(lens[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 12


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from styler_amd.dist import aggregate_throughput, allreduce_mean_, allreduce_sum_
rank, world = int(sys.argv[1]), 2
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
t, u = aggregate_throughput(1.0 + rank, 100 * (rank + 1))
assert (t, u) == (2.0, 300.0), (t, u)
g = torch.arange(10, dtype=torch.float32) * (rank + 1)
assert allreduce_mean_(g, bucket_bytes=16) == []
assert torch.allclose(g, torch.arange(10, dtype=torch.float32) * 1.5), g
# two-phase (overlapped) SUM reduction of a flat buffer: tail range first, head at step time -- same result as one
# pass; the 1 / world of the mean is folded into the clip + Adam kernel (grad_scale), not a pass over the buffer
flat = torch.arange(37, dtype=torch.float32) * (rank + 1)
tail = allreduce_sum_(flat[20:], bucket_bytes=24)
head = allreduce_sum_(flat[:20], bucket_bytes=24)
assert len(tail) == 3 and len(head) == 4
for w in tail + head:
    w.wait()
assert torch.allclose(flat, torch.arange(37, dtype=torch.float32) * 3.0), flat
# the optional bf16 transport (STYLER_ALLREDUCE_BF16): two launch points, cast down -> bf16 SUM -> cast back; the result
# is the bf16 sum of the two ranks' bf16-rounded values: three roundings of 2^-9, i.e. within 2^-7 of the fp32 sum (pinned)
from styler_amd.dist import Bf16Reducer, allreduce_preflight
g32 = (torch.linspace(-3.0, 5.0, 64) ** 3) * (rank + 1)
exact = (torch.linspace(-3.0, 5.0, 64) ** 3) * 3.0
red = Bf16Reducer(g32)
works = red.start(40, 64) + red.start(0, 40)
for w in works:
    w.wait()
red.finish()
want = (((torch.linspace(-3.0, 5.0, 64) ** 3) * 1.0).bfloat16() + ((torch.linspace(-3.0, 5.0, 64) ** 3) * 2.0).bfloat16()).float()
assert torch.equal(g32, want), (g32 - want).abs().max()
assert float(((g32 - exact).abs() / exact.abs().clamp_min(1e-2)).max()) <= 2.0 ** -7
pf = allreduce_preflight("cpu", nbytes=1 << 16, reps=2)
assert pf["ranks"] == 2 and pf["fp32"]["bytes"] == 1 << 16 and pf["fp32"]["ms"] > 0 and "bf16" in pf
dist.destroy_process_group()
print("ok", rank)
"""


def test_two_rank_gloo_helpers(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, ROOT], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_collate_matches_reference(golden):
    """styler_amd.data.collate_fn vs the reference Dataset.collate_fn (tests/golden/make_golden_collate.py)."""
    import numpy as np
    from golden.make_golden_collate import synthetic_items
    from styler_amd.data import collate_fn
    g = golden("collate")
    out = collate_fn(synthetic_items())
    assert len(out) == 4
    for k, sub in enumerate(out):
        for kk, vv in sub.items():
            ref = g[f"b{k}:{kk}"]
            if kk == "id":
                assert list(vv) == [str(x) for x in ref]
            else:
                assert np.asarray(vv).shape == ref.shape and np.array_equal(np.asarray(vv, dtype=ref.dtype), ref), (k, kk)


def test_derived_layout_specs_match_torch_permutes():
    """The strided-copy recipes (runtime.Seg) that the one-launch refresh of the derived weight layouts executes on the
    GPU, emulated element by element on the host against the torch expressions they replaced: conv kernel layout
    [n, kw, cin], tap-flipped transposed dX layout [cin, kw, n], fused row / transposed-column concatenations and the
    summed LSTM bias."""
    import numpy as np
    import torch
    from styler_amd.runtime import Seg, seg_conv_bwd, seg_conv_fwd, seg_rows, seg_transposed

    def run(segs, shape):
        out = np.full(int(np.prod(shape)), np.nan, dtype=np.float32)
        for sg in segs:
            src = sg.src.detach().numpy().reshape(-1)
            src2 = sg.src2.detach().numpy().reshape(-1) if sg.src2 is not None else None
            for a0 in range(sg.dims[0]):
                for a1 in range(sg.dims[1]):
                    for a2 in range(sg.dims[2]):
                        si = sg.src_off + a0 * sg.sstr[0] + a1 * sg.sstr[1] + a2 * sg.sstr[2]
                        v = src[si] + (src2[si] if src2 is not None else 0.0)
                        out[sg.dst_off + a0 * sg.dstr[0] + a1 * sg.dstr[1] + a2 * sg.dstr[2]] = v
        assert not np.isnan(out).any(), "a derived layout has elements no segment writes"
        return torch.from_numpy(out).view(shape)

    g = torch.Generator().manual_seed(0)
    w = torch.randn(6, 4, 5, generator=g)                                        # Conv1d weight [n, cin, kw]
    assert torch.equal(run([seg_conv_fwd(w)], (6, 20)), w.permute(0, 2, 1).reshape(6, 20))
    assert torch.equal(run([seg_conv_bwd(w)], (4, 30)), w.flip(2).permute(1, 2, 0).reshape(4, 30))
    lin = torch.randn(6, 4, generator=g)                                          # Linear weight [n, cin]
    assert torch.equal(run([seg_conv_fwd(lin)], (6, 4)), lin)
    assert torch.equal(run([seg_conv_bwd(lin)], (4, 6)), lin.t())
    q, k, v = (torch.randn(3, 4, generator=g) for _ in range(3))                  # fused QKV
    assert torch.equal(run([seg_rows(u, i * 3) for i, u in enumerate((q, k, v))], (9, 4)), torch.cat([q, k, v]))
    assert torch.equal(run([seg_transposed(u, i * 3, 9) for i, u in enumerate((q, k, v))], (4, 9)), torch.cat([q, k, v]).t())
    bi, bh, bir, bhr = (torch.randn(8, generator=g) for _ in range(4))            # b_ih + b_hh, forward | reverse
    segs = [Seg(bi, (8,), (1,), (1,), src2=bh), Seg(bir, (8,), (1,), (1,), dst_off=8, src2=bhr)]
    assert torch.allclose(run(segs, (16,)), torch.cat([bi + bh, bir + bhr]))
    oh = torch.randn(3, 7, 5, generator=g)                                        # one-hot conv weight [C, 257, 5] -> [5, 257, C]
    C, Q, K = oh.shape
    assert torch.equal(run([Seg(oh, (K, Q, C), (1, K, Q * K), (Q * C, C, 1))], (K, Q, C)), oh.permute(2, 1, 0).contiguous())


def _emulated_conv_gemm_pad(x, w, bias=None, *, kw, pad, act=0, prec=0, res=None, out=None):
    """What styler_conv_gemm_pad computes (include/styler_hip.h), in torch on the CPU, honouring strided row views."""
    import torch.nn.functional as F
    B, L, cin = x.shape
    n = w.shape[0]
    xp = F.pad(x, (0, 0, pad, kw - 1 - pad))
    cols = torch.cat([xp[:, j:j + L] for j in range(kw)], dim=-1)          # [B, L, kw*cin], tap-major like the weight
    y = cols @ w.float().t()
    if bias is not None:
        y = y + bias
    y = {0: lambda v: v, 1: torch.relu, 2: torch.tanh, 4: lambda v: F.leaky_relu(v, 0.1)}[act](y)
    if res is not None:
        y = y + res
    if out is None:
        return y
    out.copy_(y)
    return out


def test_hifigan_host_composition_matches_reference(golden, hifigan_state_dict, monkeypatch):
    """The vocoder's host logic -- weight-norm folding, the ConvTranspose -> 3-tap (phase, c_out) rearrangement, dilation
    by phase views, the 11-tap split, buffer reuse -- checked on the CPU with the two HIP entry points it calls
    replaced by their torch definitions.  (The HIP kernels themselves: tests/test_10_hip_parity.py, -m gpu.)"""
    import json
    import numpy as np
    import torch.nn.functional as F
    from styler_amd import hifigan, ops
    g = golden("hifigan")
    h = hifigan.config_v1()
    gen = hifigan.Generator(h)
    ref_shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "hifigan_state_dict_shapes.json")))
    assert {k: list(v.shape) for k, v in gen.state_dict().items()} == ref_shapes
    gen.load_state_dict(hifigan_state_dict)
    gen.fork_streams = False                             # HIP streams: GPU tests

    calls = {"gemm": 0, "leaky": 0}

    def gemm(*a, **k):
        calls["gemm"] += 1
        return _emulated_conv_gemm_pad(*a, **k)

    def leaky(a, b=None, c=None, *, scale=1.0, slope=0.1, out=None):
        calls["leaky"] += 1
        v = a + (0 if b is None else b) + (0 if c is None else c)
        v = F.leaky_relu(v * scale, slope)
        if out is None:
            return v
        out.copy_(v)
        return out

    monkeypatch.setattr(ops, "conv_gemm_pad", gemm)
    monkeypatch.setattr(ops, "leaky_sum", leaky)
    mel = torch.from_numpy(g["mel"]).transpose(1, 2).contiguous()
    for fold in (False, True):
        if fold:
            gen.remove_weight_norm()
            assert "conv_pre.weight" in gen.state_dict() and "conv_pre.weight_g" not in gen.state_dict()
        plan = gen._prepare(ops.PREC_F32)
        with torch.no_grad():
            wav = torch.stack([gen._item(mel[b], plan, ops.PREC_F32) for b in range(mel.shape[0])])
        err = float((wav - torch.from_numpy(g["wav"])[:, 0]).abs().max())
        assert err < 5e-6, err
    assert calls["gemm"] > 0 and calls["leaky"] > 0
    # a folded checkpoint loads into a fresh (weight-normed) generator
    gen2 = hifigan.Generator(h)
    gen2.load_state_dict(gen.state_dict())
    assert torch.equal(gen2.conv_post.weight, gen.conv_post.weight)
    with pytest.raises(RuntimeError):
        gen2(torch.zeros(1, 80, 4))                      # no CPU fallback


def test_feature_store_matches_reference(golden, tmp_path):
    """styler_amd.data.FeatureStore vs the reference `dataset.Dataset.__getitem__ / collate_fn` reading the same
    synthetic `.npy` store (tests/golden/make_golden_store.py regenerates it here from the same seed)."""
    import numpy as np
    from golden.make_golden_store import tokenizer, write_store
    from styler_amd.data import FeatureStore
    write_store(str(tmp_path))
    g = golden("store")
    ds = FeatureStore(str(tmp_path), tokenizer)
    assert len(ds) == int(g["n"])
    for idx in (0, 7, 19):
        item = ds[idx]
        keys = {k[len(f"item{idx}_"):] for k in g.files if k.startswith(f"item{idx}_")}
        assert set(item) == keys
        for k in keys:
            ref = g[f"item{idx}_{k}"]
            if k == "id":
                assert item[k] == str(ref)
            else:
                assert item[k].dtype == ref.dtype and np.array_equal(item[k], ref), k
    subs = ds.collate_fn([ds[i] for i in range(16)])
    assert len(subs) == int(g["n_sub"])
    for j in (0, 3):
        for k, v in subs[j].items():
            ref = g[f"sub{j}_{k}"]
            if k == "id":
                assert list(v) == [str(x) for x in ref]
            else:
                assert np.array_equal(v, ref), k


def test_batch_feeder_shards_prefetches_and_reshuffles(tmp_path):
    import numpy as np
    from golden.make_golden_store import tokenizer, write_store
    from styler_amd.data import BatchFeeder, FeatureStore, to_device
    write_store(str(tmp_path))
    ds = FeatureStore(str(tmp_path), tokenizer)
    feeders = [BatchFeeder(ds, "cpu", batch_size=2, rank=r, world=2, seed=3, depth=2) for r in range(2)]
    groups = [f.groups() for f in feeders]
    flat = np.concatenate([np.concatenate(g) for g in groups])
    # 5 groups of 4 on 2 ranks: disjoint, and EQUAL counts (every step ends in a collective): the odd group is dropped
    assert len(flat) == 16 and len(set(flat.tolist())) == 16
    assert [len(g) for g in groups] == [2, 2] and len(feeders[0]) == len(feeders[1]) == 4
    got = list(feeders[0])
    want = [to_device(sub, "cpu", pinned=False, pairs=feeders[0].pairs) for grp in groups[0]
                for sub in ds.collate_fn([ds[int(i)] for i in grp])]
    assert len(got) == len(want) == 4
    for (a, sa, ta), (b, sb, tb) in zip(got, want):
        assert (sa, ta) == (sb, tb) and a.keys() == b.keys()
        assert all(torch.equal(a[k], b[k]) for k in a)
        assert a["text"].dtype == torch.long and a["mel_target"].dtype == torch.float32
    assert feeders[0].epoch == 1                                         # next epoch: another permutation
    assert not all(np.array_equal(x, y) for x, y in zip(feeders[0].groups(), groups[0]))
    it = iter(feeders[1])                                                # abandoning an epoch mid-way stops the reader
    next(it)
    it.close()
    ordered = BatchFeeder(ds, "cpu", batch_size=2, shuffle=False)
    assert np.array_equal(np.concatenate(ordered.groups()), np.arange(20))

    class Broken(FeatureStore):
        def __getitem__(self, idx):
            raise OSError("unreadable feature file")
    with pytest.raises(OSError):
        list(BatchFeeder(Broken(str(tmp_path), tokenizer), "cpu", batch_size=2))


def test_split_channels_fn_gathers_slice_gradients(monkeypatch):
    """autograd.SplitChannelsFn (rt.fused_split): views out, ONE gathered gradient buffer back (unused slices zero);
    the strided-copy kernel is replaced by its torch definition here."""
    from styler_amd import autograd as AG, ops

    def copy_rows_multi(pairs):
        for src, dst in pairs:
            dst.zero_() if src is None else dst.copy_(src)
    monkeypatch.setattr(ops, "copy_rows_multi", copy_rows_multi)
    x = torch.randn(2, 3, 10, requires_grad=True)
    parts = AG.SplitChannelsFn.apply(x * 1.0, 2)
    assert len(parts) == 5 and all(p.shape == (2, 3, 2) for p in parts)
    ((parts[0] * 2).sum() + (parts[3] ** 2).sum() + parts[4][..., :1].sum()).backward()
    ref = torch.zeros_like(x)
    ref[..., 0:2] = 2
    ref[..., 6:8] = 2 * x.detach()[..., 6:8]
    ref[..., 8:9] = 1
    assert torch.allclose(x.grad, ref)


def test_c_host_links_and_validates_arguments(tmp_path):
    """The header is plain C99 and C++17, and a C host (tests/abi/abi_host.c) can load the library and gets the
    documented negative codes for invalid arguments -- before any HIP call, so no GPU is needed."""
    from styler_amd import _lib
    hdr = os.path.join(ROOT, "include", "styler_hip.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                   check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr], check=True)
    exe = str(tmp_path / "abi_host")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi", "abi_host.c"), "-o", exe, "-ldl"], check=True)
    out = subprocess.run([exe, _lib.LIB_PATH], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip() == "abi=1 null=-1 align=-2 kw=-1 plan=-1"


# ---- checkpoint round trip (train.py:54,61-66,221-224) -----------------------------------------------------------------
def _flat_layout(params):
    align = lambda k: (k + 3) & ~3
    return sum(align(p.numel()) for p in params)


def _flat_adam_step(p, g, m, v, lr, step, b1=0.9, b2=0.98, eps=1e-9):
    """adam_kernel (csrc/misc_bwd.hip) restated in torch on flat buffers, no clipping."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2s = 1 - b1 ** step, (1 - b2 ** step) ** 0.5
    p.sub_((lr / bc1) * m / (v.sqrt() / bc2s + eps))


def test_checkpoint_round_trip_with_torch_adam():
    """A checkpoint written by the REFERENCE layout -- {'model': nn.DataParallel(model).state_dict() (module.-prefixed),
    'optimizer': torch.optim.Adam(model.parameters(), ...).state_dict()} -- loads into the flat moment buffers, one more
    (flat) Adam step matches torch's next step, and the dict written back loads into a fresh torch.optim.Adam."""
    import copy
    from styler_amd import STYLER, hparams as hp
    from styler_amd import checkpoint as C
    torch.manual_seed(3)
    model = STYLER()
    all_params = list(model.parameters())
    train_params = [p for p in all_params if p.requires_grad]
    unused = {id(p) for n, p in model.named_parameters() if "pitch_norm_linear" in n}
    opt = torch.optim.Adam(model.parameters(), betas=hp.betas, eps=hp.eps, weight_decay=hp.weight_decay, lr=1e-3)
    g = torch.Generator().manual_seed(5)

    def fake_grads():
        for p in train_params:           # the forward never calls pitch_norm_linear: grad stays None, torch skips it
            p.grad = None if id(p) in unused else torch.randn(p.shape, generator=g) * 0.01
    for _ in range(2):
        fake_grads()
        opt.step()
    ckpt = {"model": {"module." + k: v.clone() for k, v in model.state_dict().items()},
            "optimizer": copy.deepcopy(opt.state_dict())}
    assert len(ckpt["model"]) == 328 and all(k.startswith("module.") for k in ckpt["model"])
    assert len(ckpt["optimizer"]["state"]) == len(train_params) - len(unused)

    # ---- load into a fresh model + flat buffers ----
    model2 = STYLER()
    C.load_model_state_dict(model2, ckpt["model"])
    for a, b in zip(model.state_dict().values(), model2.state_dict().values()):
        assert torch.equal(a, b)
    params2 = [p for p in model2.parameters() if p.requires_grad]
    n = _flat_layout(params2)
    flat_m, flat_v = torch.full((n,), 7.0), torch.full((n,), 7.0)
    steps = C.flat_from_adam_state(ckpt["optimizer"], model2, params2, flat_m, flat_v)
    assert steps == 2
    off = 0
    for p_ref, p in zip(train_params, params2):
        k = p.numel()
        st = opt.state.get(p_ref)
        if st is None:
            assert float(flat_m[off:off + k].abs().max()) == 0.0 and float(flat_v[off:off + k].abs().max()) == 0.0
        else:
            assert torch.equal(flat_m[off:off + k].view(p.shape), st["exp_avg"])
            assert torch.equal(flat_v[off:off + k].view(p.shape), st["exp_avg_sq"])
        off += (k + 3) & ~3

    # ---- one more step: torch on the original, the flat formula on the loaded state ----
    fake_grads()
    flat_p, flat_g = torch.zeros(n), torch.zeros(n)
    off = 0
    for p_ref, p in zip(train_params, params2):
        k = p.numel()
        flat_p[off:off + k] = p.detach().reshape(-1)
        if p_ref.grad is not None:
            flat_g[off:off + k] = p_ref.grad.reshape(-1)
        off += (k + 3) & ~3
    opt.step()
    _flat_adam_step(flat_p, flat_g, flat_m, flat_v, 1e-3, steps + 1)
    off = 0
    for p_ref in train_params:
        k = p_ref.numel()
        assert torch.allclose(flat_p[off:off + k].view(p_ref.shape), p_ref.detach(), rtol=1e-5, atol=1e-7)
        off += (k + 3) & ~3

    # ---- write back in torch's layout and load it into a fresh torch.optim.Adam ----
    sd = C.adam_state_from_flat(model2, params2, flat_m, flat_v, steps + 1, 1e-3)
    assert sorted(sd["state"]) == sorted(ckpt["optimizer"]["state"])            # no entry for the never-used MLP
    opt3 = torch.optim.Adam(model2.parameters(), betas=hp.betas, eps=hp.eps, weight_decay=hp.weight_decay, lr=1e-3)
    opt3.load_state_dict(sd)
    by_index = dict(enumerate(model2.parameters()))
    for idx, st in opt.state_dict()["state"].items():
        got = opt3.state[by_index[idx]]
        assert int(got["step"]) == 3
        assert torch.allclose(got["exp_avg"], st["exp_avg"], rtol=1e-5, atol=1e-9)        # torch lerps, the kernel FMAs
        assert torch.allclose(got["exp_avg_sq"], st["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    # torch 1.6 (the reference's pin) stored `step` as a python int: accepted too
    legacy = copy.deepcopy(ckpt["optimizer"])
    for st in legacy["state"].values():
        st["step"] = int(st["step"])
    assert C.flat_from_adam_state(legacy, model2, params2, flat_m, flat_v) == 2
    assert set(C.model_state_dict(model2)) == set(ckpt["model"])


def test_feeder_groups_equal_across_ranks():
    """Every rank must run the same number of steps per epoch (each ends in a collective): with 5 groups and 2 ranks the
    odd group is dropped, the ranks' groups are disjoint."""
    from styler_amd.data import BatchFeeder

    class Store:
        def __len__(self):
            return 5 * 4 + 3                                   # 5 full groups of batch_size^2 = 4 items, 3 left over
    feeders = [BatchFeeder(Store(), "cpu", batch_size=2, rank=r, world=2, seed=1) for r in range(2)]
    groups = [f.groups() for f in feeders]
    assert len(groups[0]) == len(groups[1]) == 2 and len(feeders[0]) == len(feeders[1]) == 4
    seen = [int(i) for gs in groups for g in gs for i in g]
    assert len(seen) == len(set(seen)) == 16


def test_noam_lr_at_step_zero():
    from styler_amd.optimizer import ScheduledOptim, noam_lr
    assert noam_lr(0) == 0.0
    so = ScheduledOptim(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]), 256, 4000, 0)
    assert so._get_lr_scale() == 0.0                           # optimizer.py:21-25 at n = 0: min(inf, 0) = 0


def test_feeder_shape_buckets(tmp_path):
    """`bucket=(s_step, t_step)`: every sub-batch is padded up to the grid (zeros, lengths untouched), the reported maxima are
    the padded extents, and the number of distinct shapes an epoch produces collapses."""
    import numpy as np
    from golden.make_golden_store import tokenizer, write_store
    from styler_amd.data import BatchFeeder, FeatureStore, bucket_up
    write_store(str(tmp_path))
    ds = FeatureStore(str(tmp_path), tokenizer)
    exact = list(BatchFeeder(ds, "cpu", batch_size=2, shuffle=False, depth=2))
    bucketed = list(BatchFeeder(ds, "cpu", batch_size=2, shuffle=False, depth=2, bucket=(8, 64)))
    assert len(exact) == len(bucketed)
    for (a, sa, ta), (b, sb, tb) in zip(exact, bucketed):
        assert (sb, tb) == (bucket_up(sa, 8), bucket_up(ta, 64)) and sb % 8 == 0 and tb % 64 == 0
        assert b["text"].shape[1] == sb and b["mel_target"].shape[1] == tb and b["f0"].shape[1] == tb and b["log_D"].shape[1] == sb
        assert torch.equal(a["src_len"], b["src_len"]) and torch.equal(a["mel_len"], b["mel_len"])
        assert torch.equal(b["text"][:, :sa], a["text"]) and float(b["text"][:, sa:].abs().sum()) == 0
        assert torch.equal(b["mel_target"][:, :ta], a["mel_target"]) and float(b["mel_target"][:, ta:].abs().sum()) == 0
        assert float(b["log_D"][:, sa:].abs().sum()) == 0                     # log(0 + 1): the padded durations stay zero
    assert len({(s, t) for _, s, t in bucketed}) < len({(s, t) for _, s, t in exact})
    assert bucket_up(990, 64) == 1001 and bucket_up(1001, 64) == 1001         # train mode: never past the position table


def test_feeder_pairs_layout_equals_add_pair_inputs(tmp_path):
    """`BatchFeeder(pairs=True)` / `to_device(pairs=True)`: the collated pair_* tensors are the concatenation
    of the two halves `training.add_pair_inputs` stacks on the device -- exact and bucketed; a feed WITHOUT them still
    carries everything the step needs (the graphed step then stacks them itself and skips keys it does not own)."""
    from golden.make_golden_store import tokenizer, write_store
    from styler_amd.data import BatchFeeder, FeatureStore
    from styler_amd.training import PAIR_KEYS, add_pair_inputs
    write_store(str(tmp_path))
    ds = FeatureStore(str(tmp_path), tokenizer)
    for bucket in (None, (8, 64)):
        plain = list(BatchFeeder(ds, "cpu", batch_size=2, shuffle=False, depth=2, bucket=bucket, pairs=False))
        paired = list(BatchFeeder(ds, "cpu", batch_size=2, shuffle=False, depth=2, bucket=bucket, pairs=True))
        assert len(plain) == len(paired) > 0
        for (a, sa, ta), (b, sb, tb) in zip(plain, paired):
            assert (sa, ta) == (sb, tb) and set(b) == set(a) | set(PAIR_KEYS)
            ref = add_pair_inputs(dict(a))
            for k, (h0, h1) in PAIR_KEYS.items():
                assert b[k].dtype == ref[k].dtype and torch.equal(b[k], ref[k]), k
                assert torch.equal(b[k], torch.cat([a[h0], a[h1]], 0)), k
            for k in a:
                assert torch.equal(a[k], b[k]), k
    from styler_amd import rt
    assert BatchFeeder(ds, "cpu", batch_size=2).pairs == bool(rt.pair_audio)      # the default follows the training step


def test_torch_library_schemas_and_fake_kernels():
    """torch.ops.styler.* exist with fake (meta) kernels: shapes propagate without a GPU (tracing / torch.compile front end)."""
    import styler_amd.torch_ops as T
    from torch._subclasses.fake_tensor import FakeTensorMode
    assert all(hasattr(torch.ops.styler, n) for n in T.OPS)
    with FakeTensorMode():
        y = torch.ops.styler.conv_gemm(torch.empty(2, 10, 256), torch.empty(512, 256, 5), torch.empty(512), 1, 0)
        assert tuple(y.shape) == (2, 10, 512)
        o, lse = torch.ops.styler.attention(torch.empty(2, 10, 768), torch.empty(2, dtype=torch.int64), 0)
        assert tuple(o.shape) == (2, 10, 256) and tuple(lse.shape) == (2, 4, 10)
        y, s = torch.ops.styler.add_layernorm(torch.empty(2, 10, 256), None, torch.empty(256), torch.empty(256), None)
        assert tuple(y.shape) == (2, 10, 256)
        y, ml = torch.ops.styler.length_regulate(torch.empty(2, 5, 64), torch.empty(2, 5, dtype=torch.int64), 33)
        assert tuple(y.shape) == (2, 33, 64) and tuple(ml.shape) == (2,)
        mel, en, ei, fl = torch.ops.styler.stft_mel(torch.empty(3, 22050), None)
        assert tuple(mel.shape) == (3, 87, 80) and tuple(fl.shape) == (3,)
        # round 6: one forward + one backward operator per fused kernel
        assert len(T.OPS) == 28
        dx, dw, db = torch.ops.styler.conv_gemm_bwd(torch.empty(2, 10, 512), torch.empty(2, 10, 256), torch.empty(512, 256, 5),
                                                    torch.empty(2, 10, 512), 1, 0)
        assert tuple(dx.shape) == (2, 10, 256) and tuple(dw.shape) == (512, 256, 5) and tuple(db.shape) == (512,)
        assert tuple(torch.ops.styler.attention_bwd(torch.empty(2, 10, 768), torch.empty(2, 10, 256), torch.empty(2, 10, 256),
                                                    torch.empty(2, 4, 10), torch.empty(2, dtype=torch.int64), 0).shape) == (2, 10, 768)
        assert tuple(torch.ops.styler.length_regulate_bwd(torch.empty(2, 33, 64), torch.empty(2, 5, dtype=torch.int64), 5).shape) == (2, 5, 64)
        y, st = torch.ops.styler.groupnorm_relu(torch.empty(2, 30, 320), torch.empty(320), torch.empty(320))
        assert tuple(st.shape) == (2, 20, 2)
        o = torch.ops.styler.batchnorm_act(torch.empty(4, 30, 512), torch.empty(512), torch.empty(512), torch.empty(512),
                                           torch.empty(512), 2, 0.5, 3, 2)
        assert [tuple(t.shape) for t in o] == [(4, 30, 512), (2, 512), (2, 512), (512,), (512,)]
        o = torch.ops.styler.lstm_bidir(torch.empty(2, 10, 512), torch.empty(2, 256, 64), 64)
        assert [tuple(t.shape) for t in o] == [(2, 10, 128), (2, 10, 512), (2, 10, 128)]
        y, s_ = torch.ops.styler.linear_ln(torch.empty(2, 10, 1024), torch.empty(256, 1024), torch.empty(256), torch.empty(2, 10, 256),
                                           torch.empty(256), torch.empty(256), None)
        assert tuple(y.shape) == (2, 10, 256)
        assert torch.ops.styler.nll3(torch.empty(4, 2), torch.empty(4, 2), torch.empty(4, 2), torch.empty(4, dtype=torch.int64)).shape == ()
        assert tuple(torch.ops.styler.mel_calibrate(torch.empty(2, 40, 1152), torch.empty(2, dtype=torch.int64),
                                                    torch.empty(2, dtype=torch.int64), 7).shape) == (2, 7, 1152)
        assert tuple(torch.ops.styler.aug_classifier_tail(torch.empty(3, 6, 256), torch.empty(256), torch.empty(256), torch.empty(2, 256),
                                                          torch.empty(2)).shape) == (3, 2)

"""Data-parallel train step on real device code with more than one rank (train.py:33 replaced by one process per
GPU + gradient all-reduce).  The 1-GPU box cannot run RCCL with two ranks, so both ranks share cuda:0 and the collectives
go over gloo: everything except the transport is the N > 1 path (utterance sharding, per-rank padding, parameter
broadcast, two-graph step with the decoder-side all-reduce between the replays, 1 / world folded into clip + Adam)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_step_equals_single_rank_accumulation(tmp_path, ref_state_dict, monkeypatch):
    """Two ranks, each on its shard of one global batch, against ONE rank that accumulates the two shards as micro-batches
    (acc_steps = 2: loss / 2 each -> the mean of the per-shard gradients, which is what the all-reduce mean computes; per-
    shard BatchNorm statistics, as per replica in the reference's DataParallel).  Same summed gradient, same parameters
    after the update; the step really ran as two graphs; the ranks drew their weights from rank 0."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import global_batch, shard_batch
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), "2", port,
                               str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(2))
    assert r0["graphs"] == r1["graphs"] == 2, "the split (overlapped all-reduce) capture fell back to one graph"
    assert sorted(r0["idx"] + r1["idx"]) == list(range(8))
    assert torch.equal(r0["flat_g"], r1["flat_g"]) and torch.equal(r0["flat_p"], r1["flat_p"]) and r0["lr"] == r1["lr"]
    assert (r0["seed"], r1["seed"]) == (0, 1)                       # one dropout stream per rank
    assert r0["info"]["allreduce_world"] == 2 and r0["info"]["allreduce_bytes"] == r0["flat_g"].numel() * 4

    from styler_amd import STYLER, hparams as hp, rt
    from styler_amd.training import TrainState, train_step
    dev = torch.device("cuda:0")
    gb = global_batch()
    rt.disable_dropout = True
    monkeypatch.setattr(hp, "acc_steps", 2)
    try:
        m = STYLER()
        m.load_state_dict(ref_state_dict)
        m = m.to(dev).train()
        st = TrainState(m)
        for r in (r0, r1):
            _, lr = train_step(m, st, {k: v.to(dev) for k, v in shard_batch(gb, r["idx"]).items()})
        torch.cuda.synchronize()
        mean_g = st.flat_g.cpu()                                    # 0.5 * (g_shard0 + g_shard1)
        err_g = float((0.5 * r0["flat_g"] - mean_g).abs().max()) / float(mean_g.abs().max())
        assert err_g <= 1e-5, err_g
        assert lr == r0["lr"]
        err_p = float((r0["flat_p"] - st.flat_p.cpu()).abs().max())
        assert err_p <= 1e-6, err_p
        st.close()
    finally:
        rt.disable_dropout = False

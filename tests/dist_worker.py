"""Worker of tests/test_15_dist_gpu.py: one rank of a two-rank data-parallel train step on ONE GPU (both ranks on cuda:0,
collectives over gloo -- RCCL refuses two ranks on one device; the control flow is the N > 1 path of bench.py).

    python dist_worker.py <rank> <world> <port> <out_dir> [graph|acc2|buckets|buckets_sync]
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def shard_batch(batch, idx):
    """Items `idx` of a padded batch, re-padded to their own max lengths (what a rank's collate would produce)."""
    idx = torch.as_tensor(idx)
    sb = {k: v[idx] for k, v in batch.items()}
    S, T = int(sb["src_len"].max()), int(sb["mel_len"].max())
    for k in ("text", "D", "log_D"):
        sb[k] = sb[k][:, :S].contiguous()
    for k in ("mel_target", "mel_aug", "f0", "f0_norm", "f0_norm_aug", "energy", "energy_input", "energy_input_aug"):
        sb[k] = sb[k][:, :T].contiguous()
    return sb


def global_batch():
    from closed_form import make_batch
    return make_batch(8, 10, 30, 2, 9, seed=91)


def micro_batches(gb, idx):
    """A rank's shard cut into two micro-batches (gradient accumulation, train.py:175-178)."""
    h = len(idx) // 2
    return [shard_batch(gb, idx[:h]), shard_batch(gb, idx[h:])]


def second_batch():
    from closed_form import make_batch
    return make_batch(8, 24, 44, 3, 11, seed=92)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "graph"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    if mode.startswith("rccl1"):         # ONE rank on the real transport (backend "nccl" = RCCL), as bench.py initialises it
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from conftest import _reference_state_dict
    from styler_amd import STYLER, hparams as hp, rt
    from styler_amd.dist import shard_indices
    from styler_amd.training import GraphedStepCache, GraphedTrainStep, TrainState, train_step

    rt.disable_dropout = True
    model = STYLER()
    sd = _reference_state_dict()
    if rank != 0:                        # rank 0's weights must win: TrainState broadcasts them
        sd = {k: (v + 0.01 if v.is_floating_point() and "running" not in k and "position_enc" not in k and "_bins" not in k
                  else v) for k, v in sd.items()}
    model.load_state_dict(sd)
    model = model.to(dev).train()
    state = TrainState(model)
    gb = global_batch()
    idx = shard_indices(gb["text"].shape[0], rank, world, gb["mel_len"].tolist())
    res = {"idx": idx, "seed": rt.seed, "info": state.allreduce_info()}
    if mode.startswith("rccl1"):
        # the graphed step with its cut(s) and the eager step with the hook-driven launch point, collectives FORCED on one rank
        from styler_amd import dist as sdist
        assert dist.get_backend() == "nccl" and sdist.FORCE_COLLECTIVES and world == 1
        local = {k: v.to(dev) for k, v in shard_batch(gb, idx).items()}
        calls = []
        orig_ar = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (calls.append(int(a[0].numel())), orig_ar(*a, **k))[1]
        if mode == "rccl1_graph":
            step = GraphedTrainStep(model, state, local, split=True)
            res["graphs"] = len(step.graphs)
            calls.clear()
            losses, lr = step(local)
        else:
            losses, lr = train_step(model, state, local)
        res["collective_numels"] = list(calls)
        dist.all_reduce = orig_ar
    elif mode == "graph":
        local = {k: v.to(dev) for k, v in shard_batch(gb, idx).items()}
        step = GraphedTrainStep(model, state, local)              # N > 1: two graphs, tail all-reduce between the replays
        res["graphs"] = len(step.graphs)
        losses, lr = step(local)
    elif mode == "acc2":
        # eager steps with gradient accumulation AND the hook-driven overlapped all-reduce (round-2 advisor finding: the
        # hook must fire on the last micro-batch of the window only)
        hp.acc_steps = 2
        assert state.overlap_allreduce
        fired = []
        orig = state.on_decoder_grads_ready
        state.on_decoder_grads_ready = lambda: (fired.append(state._accum), orig())[1]
        fired_text = []
        orig_text = state.on_text_grads_ready
        state.on_text_grads_ready = lambda: (fired_text.append(state._accum), orig_text())[1]
        for mb in micro_batches(gb, idx):
            losses, lr = train_step(model, state, {k: v.to(dev) for k, v in mb.items()})
        res["hook_fired_at"] = fired
        res["text_hook_fired_at"] = fired_text
    elif mode in ("buckets", "buckets_sync"):
        # two consecutive optimisation steps through the graph cache, every rank with its own padded shapes per step;
        # "buckets_sync": the ranks exchange their shapes and capture every missing shape of the job in the same step
        cache = GraphedStepCache(model, state, sync_misses=(mode == "buckets_sync"))
        gb2 = second_batch()
        idx2 = shard_indices(gb2["text"].shape[0], rank, world, gb2["mel_len"].tolist())
        res["idx2"] = idx2
        for g_, i_ in ((gb, idx), (gb2, idx2)):
            losses, lr = cache({k: v.to(dev) for k, v in shard_batch(g_, i_).items()})
        res["graphs"] = [len(s_.graphs) for s_ in cache.steps.values()]
        res["misses"], res["prefetched"], res["keys"] = cache.misses, cache.prefetched, sorted(cache.steps)
    else:
        raise SystemExit(f"unknown mode {mode}")
    torch.cuda.synchronize()
    res.update(flat_g=state.flat_g.cpu(), flat_p=state.flat_p.cpu(), lr=lr, losses=[float(x) for x in losses],
               steps=state.n_current_steps)
    torch.save(res, os.path.join(out, f"rank{rank}.pt"))
    state.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Worker of tests/test_dist_gpu.py: one rank of a two-rank data-parallel train step on ONE GPU (both ranks on cuda:0,
collectives over gloo -- RCCL refuses two ranks on one device; the control flow is the N > 1 path of bench.py).

    python dist_worker.py <rank> <world> <port> <out_dir>
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


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from conftest import _reference_state_dict
    from styler_amd import STYLER, rt
    from styler_amd.dist import shard_indices
    from styler_amd.training import GraphedTrainStep, TrainState

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
    local = {k: v.to(dev) for k, v in shard_batch(gb, idx).items()}
    step = GraphedTrainStep(model, state, local)                  # N > 1: two graphs, tail all-reduce between the replays
    n_graphs = len(step.graphs)
    losses, lr = step(local)
    torch.cuda.synchronize()
    torch.save({"idx": idx, "graphs": n_graphs, "flat_g": state.flat_g.cpu(), "flat_p": state.flat_p.cpu(), "lr": lr,
                "losses": [float(x) for x in losses], "seed": rt.seed, "info": state.allreduce_info()},
               os.path.join(out, f"rank{rank}.pt"))
    state.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

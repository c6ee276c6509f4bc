#!/usr/bin/env python
"""bench.py -- headline benchmark of the STYLER hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--mode fwd] [--prec bf16|fp32] [--no-graph]

One "step" = one pass of the hot path (STYLER.forward, teacher-forced, eval) over one synthetic
VCTK-shape batch resident in HBM (B=48, src_len~U{20..60},
D~U{2..13}, 80-bin mel, clean branch only, bf16 MFMA operands / fp32 accumulate).  The metric is
BASELINE.json's: valid mel-frames per second, reported as the WHOLE-JOB aggregate over N GPUs (each rank
runs its own batch: data parallel, weak scaling, no data-path collective in the forward).

Rank 0 prints ONE JSON line.  It also carries
  roofline     : the dominant kernel (conv_gemm_kernel<2,2,bf16>: 128x128 MFMA tile engine) -- algorithmic
                 FLOPs of its launches / their HIP-event durations measured live in the timed steps;
  cpu_baseline : the oracle (plain PyTorch-CPU restatement) timed on this host's cores on the same batch.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}      # dense, MI355X_MICROARCH.md
VARIANT_NAMES = {0: "conv_gemm_kernel<1,1,f32>", 1: "conv_gemm_kernel<2,2,f32>",
                 2: "conv_gemm_kernel<1,1,bf16>", 3: "conv_gemm_kernel<2,2,bf16>"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--prec", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--mode", default="fwd", choices=["fwd", "train"],
                    help="fwd: C2 eval forward; train: full reference step (dual decode + DAT pass + losses + backward "
                         "+ grad all-reduce + clip + Adam), train.py:135-186")
    ap.add_argument("--dual", action="store_true", help="also run the noisy-branch decode (styler.py:55)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--prof-steps", type=int, default=5, help="extra eager steps with HIP-event GEMM brackets")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import styler_amd
    from styler_amd import ops, rt
    from closed_form import make_batch

    torch.manual_seed(0)                       # identical random-init weights on every rank
    train = args.mode == "train"
    model = styler_amd.STYLER().to(dev)
    model = model.train() if train else model.eval()
    model.clean_only = (not args.dual) and not train
    rt.set_precision(args.prec)
    rt.strict_inputs = False                   # no host sync inside the forward

    batch = make_batch(args.batch, 20, 60, 2, 13, seed=1234 + rank)
    frames = int(batch["mel_len"].sum())
    S, T = batch["text"].shape[1], batch["mel_target"].shape[1]
    bd = {k: v.to(dev) for k, v in batch.items()}

    if train:
        from styler_amd.training import TrainState, train_step
        state = TrainState(model)
        args.no_graph = True                   # the tape-driven step launches eagerly (lr / step count are host scalars)

    def step():
        if train:
            return train_step(model, state, bd)
        return model(bd["text"], bd["mel_target"], bd["mel_aug"], bd["f0_norm"], bd["energy_input"],
                     bd["src_len"], bd["mel_len"], bd["D"], bd["f0"], bd["energy"], S, T,
                     speaker_embed=bd["speaker_embed"])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with (torch.enable_grad() if train else torch.no_grad()):
        for _ in range(3):                      # builds derived weights (bf16 shadows etc.)
            out = step()
        torch.cuda.synchronize()
        graph = None
        if not args.no_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = step()
        run = graph.replay if graph is not None else step

        for _ in range(args.warmup):
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        barrier()
        elapsed = time.perf_counter() - t0

        # ---- live roofline measurement: HIP events around every GEMM launch, eager steps ----
        prof = ops.GemmProfiler()
        ops.gemm_profiler = prof
        for _ in range(args.prof_steps):
            step()
        torch.cuda.synchronize()
        ops.gemm_profiler = None
        gsum = prof.summary()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    fr = torch.tensor([float(frames)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_frames = float(fr.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = total_frames * args.steps / elapsed
        dom = 3 if args.prec == "bf16" else 1
        d = gsum.get(dom, {"launches": 0, "flops": 0.0, "ms": 1.0})
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["launches"] else 0.0
        peak = MFMA_PEAK_TFLOPS[args.prec]
        roofline = {"bound": "mfma", "kernel": VARIANT_NAMES[dom], "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                    "launches_per_step": d["launches"] // max(1, args.prof_steps),
                    "avg_launch_us": round(d["ms"] * 1e3 / max(1, d["launches"]), 2),
                    "gemm_ms_per_step_all_variants": round(sum(v["ms"] for v in gsum.values()) / max(1, args.prof_steps), 3)}
        cpu = None
        if not args.no_cpu:
            cpu = cpu_baseline(model, batch, S, T, model.clean_only, train=train)
        print(json.dumps({
            "metric": "mel_frames_per_sec", "value": round(value, 1),
            "unit": "valid mel-frames/s (80-bin mel, whole job)", "per_gpu": round(value / world, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.prec,
            "data": "synthetic (seeded VCTK-shape batch, random-init weights)",
            "config": {"workload": (f"C3 per-rank: full train step (dual decode + DAT pass + 10 losses + backward + "
                                    f"grad all-reduce + clip + Adam), B={args.batch}/GPU, S={S}, T={T}, valid frames="
                                    f"{frames}/GPU" if train else
                                    f"C2: STYLER.forward eval teacher-forced, {'dual' if args.dual else 'clean'}-branch, "
                                    f"B={args.batch}/GPU, S={S}, T={T}, valid frames={frames}/GPU"),
                       "launch": "eager" if graph is None else "hipGraph replay", "parallelism": f"dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu}))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(model, batch, S, T, clean_only, sample_items=16, max_threads=16, train=False):
    """Oracle forward (PyTorch-CPU eager fp32) on a bounded sample of the same workload: the first
    `sample_items` utterances of the batch, re-padded to their own max lengths, 1 warm-up + timed runs
    bounded to ~20 s.  Threads are capped (torch's intra-op pool degrades badly past ~16 threads on the
    small per-op shapes of this model); the count actually used is reported as `cores`."""
    from oracle import styler_oracle as O
    threads = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(threads)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    n_items = min(sample_items, batch["text"].shape[0])
    sb = {k: v[:n_items] for k, v in batch.items()}
    S2, T2 = int(sb["src_len"].max()), int(sb["mel_len"].max())
    for k in ("text", "D", "log_D"):
        sb[k] = sb[k][:, :S2]
    for k in ("mel_target", "mel_aug", "f0", "f0_norm", "f0_norm_aug", "energy", "energy_input", "energy_input_aug"):
        sb[k] = sb[k][:, :T2]
    frames = int(sb["mel_len"].sum())

    if train:
        sd = {k: (v.requires_grad_(True) if v.is_floating_point() and "position_enc" not in k and "_bins" not in k
                  and "running_" not in k else v) for k, v in sd.items()}

    def run():
        if train:                              # forward + DAT pass + losses + backward (no optimiser) on the CPU port
            for v in sd.values():
                v.grad = None
            O.train_losses(sd, sb, training=True)[0].backward()
            return
        with torch.no_grad():
            O.styler_forward(sd, sb["text"], sb["mel_target"], sb["mel_aug"], sb["f0_norm"], sb["energy_input"],
                             sb["src_len"], sb["mel_len"], sb["D"], sb["f0"], sb["energy"], S2, T2,
                             speaker_embed=sb["speaker_embed"], noisy_branch=not clean_only)
    t0 = time.perf_counter()
    run()
    warm = time.perf_counter() - t0
    n, t0 = 0, time.perf_counter()
    while n < 5 and (time.perf_counter() - t0) + warm < 20.0 and warm < 15.0:
        run()
        n += 1
    dt = (time.perf_counter() - t0) / n if n else warm
    return {"value": round(frames / dt, 1), "unit": "valid mel-frames/s", "cores": threads, "kind": "port",
            "sample": f"first {n_items} utterances of the batch ({frames} valid frames), {max(n, 1)} timed forward(s), "
                      f"{dt:.2f} s each, torch {torch.__version__} CPU fp32, {threads} threads of {os.cpu_count()} cores"}


if __name__ == "__main__":
    main()

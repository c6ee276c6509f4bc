#!/usr/bin/env python
"""bench.py -- headline benchmark of the STYLER hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--mode train|fwd] [--prec bf16|fp32] [--no-aux] [--no-graph]

Metric (BASELINE.json): valid mel-frames per second on a seeded synthetic VCTK-shape batch resident in HBM
(B=48 per GPU, src_len~U{20..60}, D~U{2..13}, 80-bin mel), reported as the WHOLE-JOB
aggregate over N GPUs (one process per GPU, each rank its own batch: weak scaling).

One "step" is one pass of the hot path over one batch:
  --mode train (default): the full reference optimisation step, train.py:135-186 -- forward with both decodes,
        clean + noisy losses, the DAT pass, backward, gradient all-reduce (RCCL, N > 1), clip_grad_norm_(1.0),
        Adam with the Noam schedule (BASELINE config 3 per-rank shape).  --prec bf16 (default, BASELINE config 2's
        dtype): bf16 MFMA operands in every GEMM (forward, dX and weight gradients) and in attention, fp32 accumulate,
        fp32 norms / softmax / losses / optimiser state.  --prec fp32: exact-fp32 MFMA everywhere (the 1e-3 parity mode).
  --mode fwd: BASELINE config 2 -- eval forward, teacher-forced, clean branch only, replayed from a hipGraph.
  "aux" in the same JSON line (default on, --no-aux to skip): the SAME train step in fp32 parity mode ("train_fp32") and
        the config-2 forward in the main precision ("forward_c2"), so that the arithmetic whose parity is 1e-3 has a
        driver-timed number next to the throughput mode (both are pinned to the oracle at this shape by
        tests/test_11_oracle_c2c3.py).

`value` comes from EXACTLY --steps timed steps between two barriers; "repeat" reports further blocks of the same length
(median / min / max ms per step) so that the spread of the short contract window is visible.

`python bench.py --gpus N` launches its own N ranks (re-exec under torch.distributed.run on a free local port) unless a
launcher already did (WORLD_SIZE set); rank 0 prints ONE JSON line, last.  It also carries
  roofline     : the dominant kernel family of the step (the two large-tile forward + dX GEMM engines) -- algorithmic FLOPs of
                 its launches / their HIP-event durations (events bracket each launch on the launch stream during extra eager
                 steps); `gemm256` = the 256 x 256 LDS-DMA engine alone; `traffic` = HBM-side bytes per launch from this
                 round's separate rocprofv3 --pmc passes of this command (`traffic_source` names the committed file: PMC
                 collection cannot run inside the timed process), null if that file has no record for the kernel;
  roofline.hbm : achieved HBM-side rate of the memory-bound kernels (LengthRegulator, LayerNorm, GroupNorm, BatchNorm) at the
                 step's shapes and storage formats, cache-cold (rotating operand sets inside one hipGraph), N = 1 only;
  aux          : train_fp32 (the same step in the fp32 parity mode), forward_c2, forward_c4 (BASELINE config 4, teacher-forced
                 and free-running, with the LengthRegulator's rate at that shape), N = 1 only for forward_c4;
  cpu_baseline : the oracle (plain PyTorch-CPU restatement) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import re
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0}      # dense, MI355X_MICROARCH.md
VARIANT_NAMES = {0: "conv_gemm_kernel<1,1,f32>", 1: "conv_gemm_kernel<2,2,f32>",
                 2: "conv_gemm_kernel<1,1,bf16>", 3: "conv_gemm_kernel<2,2,bf16>", 4: "conv_gemm256_kernel"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--prec", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--mode", default="train", choices=["fwd", "train"],
                    help="fwd: C2 eval forward; train: full reference step (dual decode + DAT pass + losses + backward "
                         "+ grad all-reduce + clip + Adam), train.py:135-186")
    ap.add_argument("--dual", action="store_true", help="also run the noisy-branch decode (styler.py:55)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--split-graph", action="store_true",
                    help="train: capture the two-graph step (all-reduce overlap) even on one rank (default for N > 1)")
    ap.add_argument("--prof-steps", type=int, default=3, help="extra eager steps with HIP-event GEMM brackets")
    ap.add_argument("--aux", action="store_true", help="(default on) kept for compatibility")
    ap.add_argument("--no-aux", action="store_true", help="skip the aux legs (fp32-mode train step, config-2 forward)")
    ap.add_argument("--no-hbm", action="store_true", help="skip the HBM-bound kernel rooflines (roofline.hbm)")
    ap.add_argument("--repeat", type=int, default=4, help="extra timed blocks of --steps steps (spread diagnostic)")
    ap.add_argument("--shape", default="vctk", choices=["vctk", "c4"],
                    help="vctk: src_len~U{20..60}, D~U{2..13} (C1-C3); c4: long-form S=300, T=2000 (eval forward only)")
    return ap.parse_args()


def run(args, mode, prec, rank, world, dev, dist, with_cpu=True, with_roofline=True):
    """Time `args.steps` steps of `mode` ("fwd" | "train") in precision `prec` after `args.warmup` warm-up steps.  Returns
    the result dict of rank 0 (None elsewhere)."""
    import styler_amd
    from styler_amd import ops, rt
    from styler_amd.dist import aggregate_throughput
    from closed_form import make_batch

    torch.manual_seed(0)                       # identical random-init weights on every rank
    train = mode == "train"
    model = styler_amd.STYLER().to(dev)
    model = model.train() if train else model.eval()
    model.clean_only = (not args.dual) and not train
    rt.set_precision(prec)
    rt.strict_inputs = False                   # no host sync inside the step

    if args.shape == "c4":
        batch = make_batch(args.batch, 300, 300, 5, 8, seed=1234 + rank, fix_src=300, fix_mel=2000)
    else:
        batch = make_batch(args.batch, 20, 60, 2, 13, seed=1234 + rank)
    frames = int(batch["mel_len"].sum())
    S, T = batch["text"].shape[1], batch["mel_target"].shape[1]
    bd = {k: v.to(dev) for k, v in batch.items()}
    use_graph = (not args.no_graph) and not train
    state = None
    if train:
        from styler_amd.training import GraphedTrainStep, TrainState, add_pair_inputs, train_step
        state = TrainState(model)
        add_pair_inputs(bd)                    # the feed's layout of the AudioEncoder inputs (data.BatchFeeder collates it)

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

    def timed_block():
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            go()
        barrier()
        return time.perf_counter() - t0

    with (torch.enable_grad() if train else torch.no_grad()):
        for _ in range(3):                      # builds derived weights (bf16 shadows etc.)
            step()
        torch.cuda.synchronize()
        graph = None
        if use_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
        go = graph.replay if graph is not None else step
        if train and not args.no_graph:         # forward + losses + backward replayed from hipGraphs
            try:
                graph = GraphedTrainStep(model, state, bd, split=True if args.split_graph else None)
                go = graph
            except Exception as e:              # keep the measurement alive: eager launches (reported as "eager")
                import warnings
                warnings.warn(f"hipGraph capture of the train step failed ({type(e).__name__}: {e}); launching eagerly")
                torch.cuda.synchronize()
                graph, go = None, step

        for _ in range(args.warmup):
            go()
        elapsed = timed_block()                 # THE measurement: exactly --steps steps between two barriers
        blocks = [timed_block() for _ in range(max(0, args.repeat))]      # spread diagnostic, same length each
        host_ms = None
        if graph is None:                       # eager launch: how long the host needs to ENQUEUE one step (diagnostic)
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            step()
            host_ms = (time.perf_counter() - h0) * 1e3
            torch.cuda.synchronize()

        # ---- live roofline measurement: HIP events around every MFMA-GEMM launch, extra eager steps ----
        gsum = {}
        if with_roofline and args.prof_steps > 0:
            prof = ops.GemmProfiler()
            ops.gemm_profiler = prof
            # The brackets time a kernel from its first to its last wave: next to a co-running side stream (rt.text_stream,
            # round 5: rt.pred_stream) a launch's bracket also holds the time it SHARED the chip.  The profiling steps therefore
            # run every stream's work serially -- `frac` is the kernel family alone on the chip; `frac_rocprof` (the committed
            # trace of the default command) keeps the in-graph durations with the co-runners, i.e. it is lower by the overlap.
            from styler_amd import rt as _rt
            keep_streams = (_rt.pred_stream, _rt.text_stream)
            _rt.pred_stream = _rt.text_stream = False
            try:
                for _ in range(args.prof_steps):
                    step()
                torch.cuda.synchronize()
            finally:
                _rt.pred_stream, _rt.text_stream = keep_streams
            ops.gemm_profiler = None
            gsum = prof.summary(packed_fraction=frames / float(args.batch * T))
            # what an EMPTY event bracket measures on this box (round-3 verdict: the brackets' own cost sat in the
            # family's time, 3.76 ms bracketed against 3.35 ms in the rocprof trace): subtracted per launch in `roofline`
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
            for e0, e1 in pairs:
                e0.record()
                e1.record()
            torch.cuda.synchronize()
            el = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)
            gsum["_bracket_ms"] = el[len(el) // 2]

    elapsed, total_frames = aggregate_throughput(elapsed, frames, dev)
    blocks = [aggregate_throughput(b, frames, dev)[0] for b in blocks]
    n_graphs = len(getattr(graph, "graphs", ())) if (train and graph is not None) else (1 if graph is not None else 0)
    if state is not None:
        ar = state.allreduce_info()
        if world > 1:                               # what a gradient all-reduce costs on this job's links (after the timed
            from styler_amd.dist import allreduce_preflight      # region: it must not perturb the measurement)
            # ... and where the step's own all-reduce sits: five extra steps with events around the collective waits
            state.ar_events = []
            for _ in range(5):
                go()
            torch.cuda.synchronize()
            ar["allreduce_step_timing"] = state.allreduce_timing()
            state.ar_events = None
            ar["allreduce_preflight"] = allreduce_preflight(dev)
        state.close()
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    value = total_frames * args.steps / elapsed
    workload = (f"C3 per-rank shape: full reference train step (dual decode + DAT pass + 10 losses + backward + grad "
                f"all-reduce + clip + Adam, train.py:135-186), B={args.batch}/GPU, S={S}, T={T}, valid frames={frames}/GPU"
                if train else
                f"C2: STYLER.forward eval teacher-forced, {'dual' if args.dual else 'clean'}-branch, "
                f"B={args.batch}/GPU, S={S}, T={T}, valid frames={frames}/GPU")
    res = {"value": round(value, 1), "ms_per_step": round(ms, 4), "workload": workload, "dtype": prec,
           "launch": ("hipGraph replay" + (" (2 graphs, all-reduce between)" if train and n_graphs == 2 else ""))
           if graph is not None else "eager", "graphs": n_graphs}
    if blocks:
        per = sorted(b / args.steps * 1e3 for b in blocks + [elapsed])
        res["repeat"] = {"blocks": len(per), "steps_per_block": args.steps, "ms_per_step_median": round(per[len(per) // 2], 4),
                         "ms_per_step_min": round(per[0], 4), "ms_per_step_max": round(per[-1], 4)}
    if train:
        res["allreduce"] = ar
    if host_ms is not None:
        res["host_enqueue_ms_per_step"] = round(host_ms, 2)
    if gsum:
        res["roofline"] = roofline_of(gsum, train, prec, max(1, args.prof_steps))
    if with_cpu and not args.no_cpu and world == 1:     # reported at N = 1 only (rank 0's host cores)
        res["cpu_baseline"] = cpu_baseline(model, batch, S, T, model.clean_only, train=train)
    return res


def rocprof_family_ms(patterns):
    """ms per step of the kernels matching `patterns` in the COMMITTED rocprofv3 kernel trace of the default command (hipGraph
    replay), or None: (total_ms of the matching rows) / (traced steps).  The trace also holds the un-stepped warm-up passes
    of the command (forward + backward without an optimiser step), so the traced steps are the calls of a kernel that runs
    ONCE in every forward + backward -- `pack_plan_kernel` -- not the calls of `adam_kernel` (round 4 divided by the latter:
    28 instead of 31, `frac_rocprof` 0.287 where the trace says 0.32)."""
    for name in ("r06_train_bf16_graph_kernel_stats.txt", "r05_train_bf16_graph_kernel_stats.txt", "r04_train_bf16_graph_kernel_stats.txt",
                 "r03_train_bf16_graph_kernel_stats.txt"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        tot, steps = 0.0, 0
        for line in open(path):
            m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.]+)\s+[0-9.]+\s+[0-9.]+\s+[0-9.]+\s+[0-9.]+\s*$", line.rstrip())
            if not m:
                continue
            kname, calls, total = m.group(1), int(m.group(2)), float(m.group(3))
            if kname.strip().startswith("pack_plan_kernel"):
                steps = calls
            elif kname.strip().startswith("adam_kernel") and not steps:
                steps = calls
            if any(p in kname for p in patterns):
                tot += total
        if steps:
            return tot / steps, f"profiles/{name}"
    return None, None


def _family(gsum, key, name, traffic_label, prec, ps, rocprof_patterns=None):
    peak = MFMA_PEAK_TFLOPS[prec]
    keys = key if isinstance(key, (tuple, list)) else (key,)
    d = {"launches": 0, "flops": 0.0, "ms": 0.0}
    for k in keys:
        for f in d:
            d[f] += gsum.get(k, {}).get(f, 0)
    if not d["launches"]:
        d["ms"] = 1.0
    # `achieved` / `frac`: algorithmic FLOPs over the event-bracketed time MINUS what an empty bracket measures on this box
    # (`bracket_us`, per launch); `frac_raw` keeps the uncorrected figure; `frac_rocprof` divides the same FLOPs by the
    # family's time in the committed rocprofv3 kernel trace of the default command (hipGraph replay).
    brk = gsum.get("_bracket_ms", 0.0)
    ms_cal = max(d["ms"] - brk * d["launches"], 0.5 * d["ms"])
    raw = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["launches"] else 0.0
    achieved = d["flops"] / (ms_cal * 1e-3) / 1e12 if d["launches"] else 0.0
    traffic, source = pmc_traffic(traffic_label, prec)
    out = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak,
           "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "frac_raw": round(raw / peak, 4),
           "bracket_us": round(brk * 1e3, 2), "traffic": traffic, "traffic_source": source,
           "launches_per_step": d["launches"] // ps, "avg_launch_us": round(ms_cal * 1e3 / max(1, d["launches"]), 2),
           "kernel_ms_per_step": round(ms_cal / ps, 3)}
    if rocprof_patterns and prec == "bf16":
        rms, src = rocprof_family_ms(rocprof_patterns)
        if rms:
            out["frac_rocprof"] = round(d["flops"] / ps / (rms * 1e-3) / 1e12 / peak, 4)
            out["rocprof_ms_per_step"] = round(rms, 3)
            out["rocprof_source"] = src
            out["frac_basis"] = ("frac: the family alone on the chip (profiling steps launch every stream's work serially); "
                                 "frac_rocprof: durations inside the replayed graph, where 15 % of the step has a side-stream "
                                 "kernel co-running (a launch's duration includes the time it shared the chip)")
    return out


def roofline_of(gsum, train, prec, ps):
    """The dominant MFMA kernel family of the step: `conv_gemm_kernel<2,2,..>`, the forward + dX engine (3.7 ms per replayed
    step in profiles/r02_train_bf16_graph_kernel_stats.txt against 2.3 ms for the weight-gradient engine incl. its grouped launch).  The
    weight-gradient engine is reported next to it (`weight_gradient`): in the eager profiling steps, where every gradient
    is launched stand-alone with its own split-K reduce instead of through the grouped / deferred path of the graph, its
    bracketed time is about as large, and the two used to trade places from run to run."""
    # the large-tile forward + dX engines: conv_gemm_kernel<2,2,...> (128 x 128) and, in bf16 mode, conv_gemm256_kernel (256 x 256
    # LDS-DMA, csrc/gemm256.hip) -- one family: which of the two takes a launch is a dispatch decision (styler_conv_gemm_engine)
    big = (3, 4) if prec == "bf16" else 1
    name = "conv_gemm_kernel<2,2,bf16> + conv_gemm256_kernel" if prec == "bf16" else VARIANT_NAMES[1]
    r = _family(gsum, big, name, "train_conv_gemm_2x2_bf16" if train else "fwd_conv_gemm_2x2_bf16", prec, ps,
                rocprof_patterns=("conv_gemm_kernel<2, 2, true", "conv_gemm256_kernel") if train else None)
    if prec == "bf16" and 4 in gsum:
        g = gsum[4]
        r["gemm256"] = {"kernel": "conv_gemm256_kernel", "launches_per_step": g["launches"] // ps,
                        "avg_launch_us": round(g["ms"] * 1e3 / max(1, g["launches"]), 2),
                        "achieved": round(g["flops"] / (g["ms"] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                        "frac": round(g["flops"] / (g["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[prec], 4)}
    r["all_mfma_gemm_ms_per_step"] = round(sum(v["ms"] for k, v in gsum.items() if k != "_bracket_ms") / ps, 3)
    if train:
        wg = "wgrad_bf16" if prec == "bf16" else "wgrad"
        name = ("wgrad_dma_kernel / wgrad_tr_kernel<KW,TA,TB>" if prec == "bf16" else "wgrad_kernel<KW>") + " (all tap counts, stand-alone launches)"
        r["weight_gradient"] = _family(gsum, wg, name, "train_wgrad_bf16", prec, ps,
                                       rocprof_patterns=("wgrad_tr_kernel", "wgrad_dma_kernel", "wgrad_tr_group_kernel"))
    return r


HBM_PEAK_TBS = 8.0                                        # MI355X_MICROARCH.md: HBM3E, ~8 TB/s


def _graph_loop_us(make_call, nsets, reps=4, replays=3):
    """Average duration (us) of one launch of a memory-bound entry point, cold in the 256 MB Infinity Cache: `make_call(i)`
    returns a closure over the i-th of `nsets` disjoint operand sets (their total footprint exceeds the cache, so no launch
    finds its operands resident from the previous one); the reps x nsets launches are captured into ONE hipGraph (no host
    enqueue gaps between 10-20 us kernels) and the replay is bracketed by HIP events on its stream."""
    from styler_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    calls = [make_call(i) for i in range(nsets)]
    keep = []
    # as inside a training step: statistics workspaces come from ONE slab cleared once per graph (not a memset per launch),
    # LayerNorm's parameter-gradient slots are folded later by the step's multi-tensor reduce (not timed here)
    slab, arena = ops.ZeroSlab(), ops.WgradArena()
    arena.buf = torch.empty(4, device=dev)
    ops.zero_slab, ops.wgrad_arena = slab, arena
    try:
        slab.begin(dev)
        for c in calls:
            keep.append(c())
        torch.cuda.synchronize()
        keep.clear()
        slab.total *= reps
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                slab.begin(dev)
                for _ in range(reps):
                    for c in calls:
                        keep.append(c())        # outputs stay alive through the capture: every launch writes its own block
            g.replay()
            side.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(replays):
                g.replay()
            e1.record(side)
            side.synchronize()
        torch.cuda.current_stream().wait_stream(side)
    finally:
        ops.zero_slab, ops.wgrad_arena = None, None
    us = e0.elapsed_time(e1) * 1e3 / (replays * reps * nsets)
    del g, keep
    return us


def hbm_rooflines(dev, B, S, T, frames, tag):
    """Achieved HBM-side rate of the memory-bound kernels north_star names (LengthRegulator, LayerNorm) and of the other
    normalisation kernels of the step, measured in THIS process.  `bytes` = the COMPULSORY traffic of one launch (every
    operand read once, every result written once, in the storage types the bf16-mode step uses; SURVEY 8d's per-position
    figures where the kernel moves exactly those), `frac` = bytes / avg_us / 8 TB/s.  Operand sets rotate so that the
    working set exceeds the Infinity Cache (`_graph_loop_us`)."""
    from styler_amd import ops, rt
    out = []
    g = torch.Generator().manual_seed(7)

    def nsets_for(nbytes):
        return int(max(2, min(16, -(-(640 << 20) // max(1, nbytes)))))

    def add(kernel, shape, nbytes, make_call, survey=None, nsets=None):
        us = _graph_loop_us(make_call, nsets or nsets_for(nbytes))
        rec = {"kernel": kernel, "shape": shape, "bytes": int(nbytes), "avg_us": round(us, 2),
               "achieved_TBs": round(nbytes / us / 1e6, 3), "frac": round(nbytes / us / 1e6 / HBM_PEAK_TBS, 4)}
        if survey:
            rec["bytes_model"] = survey
        out.append(rec)

    # LengthRegulator (modules.py:396-423): write 5120 B per output frame (zero fill included) + read 5120 B per phoneme
    d = torch.randint(2, 14, (B, S), generator=g) if tag != "c4" else torch.randint(5, 9, (B, S), generator=g)
    csum, _, _ = ops.duration_scan(B, S, dev, dur=d.to(dev))
    lr_bytes = B * T * 5120 + B * S * 5120 + B * S * 4
    xs = [torch.randn(B, S, 1280, device=dev) for _ in range(nsets_for(lr_bytes))]
    add("length_regulate_kernel", f"[B={B},S={S}]->[T={T}] x 1280 ch fp32", lr_bytes,
        lambda i: (lambda: ops.length_regulate(xs[i], csum, T)), "5120 B/frame written + 5120 B/phoneme read (SURVEY 8d)")
    dys = [torch.randn(B, T, 1280, device=dev) for _ in range(nsets_for(lr_bytes))]
    add("length_regulate_bwd_kernel", f"[B={B},T={T}]->[S={S}] x 1280 ch fp32", lr_bytes,
        lambda i: (lambda: ops.length_regulate_bwd(dys[i], csum, S)), "5120 B/frame read + 5120 B/phoneme written")
    del xs, dys
    if tag == "c4":
        return out
    # LayerNorm at the decoder's row count (clean + noisy decode packed: 2 x valid frames), in the storage formats of the
    # throughput-mode step: GEMM output fp32, residual stream / saved pre-norm sum / gradients along the stream bf16
    rows = 2 * frames
    gam, bet = torch.randn(256, device=dev), torch.randn(256, device=dev)
    bf = torch.bfloat16
    n = nsets_for(rows * 2560)
    a = [torch.randn(1, rows, 256, device=dev) for _ in range(n)]
    r = [torch.randn(1, rows, 256, device=dev).to(bf) for _ in range(n)]
    so = [torch.empty(1, rows, 256, device=dev, dtype=bf) for _ in range(n)]
    yo = [torch.empty(1, rows, 256, device=dev, dtype=bf) for _ in range(n)]
    add("add_layernorm_kernel (train: dropout(x) + res -> y, pre-norm sum kept; bf16 stream)",
        f"rows={rows} x 256: x fp32, res / y / sum bf16", rows * 2560,
        lambda i: (lambda: ops.add_layernorm(a[i], gam, bet, res=r[i], sum_out=so[i], out=yo[i], in_drop_p=0.2, in_drop_seed=5)),
        "x 1024 B + res 512 B read, y 512 B + sum 512 B written per row (all-fp32 form: 4096 B; eval form 3072 B, SURVEY 8d)", nsets=n)
    for i in range(n):
        ops.add_layernorm(a[i], gam, bet, res=r[i], sum_out=so[i], out=yo[i])
    dyb = [torch.randn(1, rows, 256, device=dev).to(bf) for _ in range(n)]
    dg, db = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    add("layernorm_bwd_kernel (sum, dy -> dx, dx through the dropout mask; bf16 stream)", f"rows={rows} x 256, all four bf16",
        rows * 2048,
        lambda i: (lambda: ops.layernorm_bwd(so[i], dyb[i], gam, bet, dg, db, in_drop_p=0.2, in_drop_seed=5)),
        "2 reads + 2 writes of 512 B per row (all-fp32 form: 4096 B)", nsets=n)
    del a, r, so, yo, dyb
    # GroupNorm + ReLU of the AudioEncoder (main + DAT pass stacked: 2B items), C = 320; throughput-mode storage: the conv
    # output is bf16 (rt.bf16_z) whenever the item fits the single-pass kernels, as at this shape
    zdt = torch.bfloat16 if (rt.bf16_z and ops.groupnorm_z_bf16_ok(T)) else torch.float32
    zb = 2 if zdt == torch.bfloat16 else 4
    zname = "bf16" if zb == 2 else "fp32"
    Bg, C = 2 * B, 320
    gnb = Bg * T * C
    n = nsets_for(gnb * (zb + 2))
    gx = [torch.randn(Bg, T, C, device=dev).to(zdt) for _ in range(n)]
    gy = [torch.empty(Bg, T, C, device=dev, dtype=torch.bfloat16) for _ in range(n)]
    gst = [torch.empty(Bg, C // 16, 2, device=dev) for _ in range(n)]
    ggam, gbet = torch.randn(C, device=dev), torch.randn(C, device=dev)
    add("gn_fused_kernel (GroupNorm + ReLU, single pass)", f"[{Bg},{T},{C}] {zname} -> bf16", gnb * (zb + 2),
        lambda i: (lambda: ops.groupnorm_relu(gx[i], ggam, gbet, out=gy[i], stats=gst[i])),
        f"1 read ({zb} B) + 1 write (2 B) per element; SURVEY 8d's two-pass fp32 figure is 2 reads + 1 write of 4 B", nsets=n)
    gdy = [torch.randn(Bg, T, C, device=dev).to(torch.bfloat16) for _ in range(n)]
    for i in range(n):
        ops.groupnorm_relu(gx[i], ggam, gbet, out=gy[i], stats=gst[i])
    gdg, gdb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    add("gn_bwd_fused_kernel", f"[{Bg},{T},{C}] x {zname} + dy bf16 -> dx bf16", gnb * (zb + 4),
        lambda i: (lambda: ops.groupnorm_relu_bwd(gx[i], gdy[i], ggam, gbet, gst[i], gdg, gdb, dx_bf16=True)),
        f"x read once ({zb} B), dy read once (2 B), dx written (2 B)", nsets=n)
    del gx, gy, gdy
    # PostNet BatchNorm (train statistics) + tanh + dropout, clean + noisy mel as two segments
    rows_b, Cb = 2 * B * T, 512
    zdt = torch.bfloat16 if rt.bf16_z else torch.float32
    zb = 2 if zdt == torch.bfloat16 else 4
    zname = "bf16" if zb == 2 else "fp32"
    n = nsets_for(rows_b * Cb * (2 * zb + 2))
    bx = [torch.randn(B * 2, T, Cb, device=dev).to(zdt) for _ in range(n)]
    bgam, bbet = torch.randn(Cb, device=dev), torch.randn(Cb, device=dev)
    add("bn_colstats + bn_apply (BatchNorm train + tanh + dropout)", f"[{rows_b},{Cb}] {zname} -> bf16, 2 segments",
        rows_b * Cb * (2 * zb + 2),
        lambda i: (lambda: ops.batchnorm_train(bx[i], bgam, bbet, None, None, ops.ACT_TANH, 0.5, 11, segs=2, out_bf16=True)),
        f"2 reads of x ({zb} B each: statistics, then apply) + 1 write of 2 B (SURVEY 8d: the fp32 form)", nsets=n)
    del bx
    return out


def forward_c4(args, dev, steps=5):
    """BASELINE config 4 (synthetic long form: B = 128, S = 300, T = 2000, eval mode), both ways SURVEY 8d asks for:
    teacher-forced (durations given: the LengthRegulator stress, replayed from a hipGraph) and free-running
    (synthesize.py:348-349: durations predicted, rounded on the device, T from one host read as in the reference -- eager).
    Random-init weights predict log-durations around 0, i.e. empty mels; the duration head's bias is set to log(7.67) and
    its weight scaled by 1/4 so that the free-running durations fall in 5..9 frames per phoneme (sum ~ 2000 per item)."""
    import math
    import styler_amd
    from styler_amd import rt
    from closed_form import make_batch
    torch.manual_seed(0)
    model = styler_amd.STYLER().to(dev).eval()
    with torch.no_grad():
        head = model.style_modeling.duration_predictor.linear_layer
        head.weight.mul_(0.25)
        head.bias.fill_(math.log(7.67))
    rt.weights_epoch += 1
    rt.set_precision(args.prec)
    rt.strict_inputs = False
    B = 128
    batch = make_batch(B, 300, 300, 5, 8, seed=4321, fix_src=300, fix_mel=2000)
    bd = {k: v.to(dev) for k, v in batch.items()}
    S, T = bd["text"].shape[1], bd["mel_target"].shape[1]

    def teacher():
        return model(bd["text"], bd["mel_target"], bd["mel_aug"], bd["f0_norm"], bd["energy_input"], bd["src_len"],
                     bd["mel_len"], bd["D"], bd["f0"], bd["energy"], S, T, speaker_embed=bd["speaker_embed"])

    def free():
        return model(bd["text"], bd["mel_target"], bd["mel_target"], bd["f0_norm"], bd["energy_input"], bd["src_len"],
                     bd["mel_len"], None, None, None, S, None, speaker_embed=bd["speaker_embed"])

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    res = {"workload": f"C4: STYLER.forward eval, dual branch, B={B}, S={S}, T={T} (synthetic long form)", "dtype": args.prec,
           "steps": steps}
    with torch.no_grad():
        for _ in range(2):
            teacher()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            teacher()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            teacher()
        g.replay()
        ms = timed(g.replay, steps)
        fr = int(batch["mel_len"].sum())
        res["teacher_forced"] = {"ms_per_step": round(ms, 3), "valid_frames": fr, "value": round(fr / ms * 1e3, 1),
                                 "launch": "hipGraph replay"}
        del g
        out = free()
        torch.cuda.synchronize()
        mel_len = out[7]
        fr = int(mel_len.sum().item())
        for _ in range(2):
            free()
        ms = timed(free, steps)
        res["free_running"] = {"ms_per_step": round(ms, 3), "valid_frames": fr, "T": int(out[0][0].shape[1]),
                               "frames_per_phoneme": round(fr / float(B * S), 3), "value": round(fr / ms * 1e3, 1),
                               "launch": "eager (one host read of max(mel_len) per forward, as modules.py:360)"}
    res["hbm"] = hbm_rooflines(dev, B, S, T, int(batch["mel_len"].sum()), "c4")
    del model
    torch.cuda.empty_cache()
    return res


def forward_c5(args, dev, steps=5):
    """BASELINE config 5 (end-to-end wav -> mel, inference batch 256): a ragged batch of B = 256 utterances (66 150 / 77 175 /
    88 200 samples) goes wav -> STFT -> mel / energy -> energy_rescaling (-> DeepSpeaker embedding) ->
    STYLER.forward (eval, teacher-forced, both branches) without leaving the GPU (styler_amd/pipeline.py).  Three timings:
    the audio front end alone with its own MFMA roofline (the STFT is a framing GEMM: 2 x 1024 x 1026 FLOP per frame, plus the
    80 x 513 mel projection -- audio/stft.py:59-75,156-158), front end + forward with a given speaker embedding, and the same
    with the embedding computed from the wavs (DeepSpeaker ResCNN, random-init: its parity is unpinned).  The
    front end runs in the fp32 arithmetic the parity tests pin (tests/test_12_c5_pipeline.py) and, next to it, in bf16."""
    import styler_amd
    from styler_amd import rt
    from styler_amd.deepspeaker import DeepSpeaker
    from styler_amd.pipeline import WavFrontEnd, forward_from_wavs
    torch.manual_seed(0)
    B, S = 256, 60
    lengths = (66150, 77175, 88200)
    g = torch.Generator().manual_seed(1234)
    n = torch.tensor(lengths)[torch.randint(0, 3, (B,), generator=g)]
    n[0] = lengths[2]
    wav = (torch.rand(B, lengths[2], generator=g) - 0.5) * (torch.arange(lengths[2])[None] < n[:, None])
    mel_len = 1 + n // 256
    T = int(mel_len.max())
    src_len = torch.randint(20, S + 1, (B,), generator=g)
    src_len[0] = S
    text = torch.randint(1, 152, (B, S), generator=g) * (torch.arange(S)[None] < src_len[:, None])
    # durations: every phoneme at least one frame, the remainder dealt evenly (sum = mel_len per item)
    base = (mel_len[:, None] // src_len[:, None]).expand(B, S)
    extra = (torch.arange(S)[None] < (mel_len % src_len)[:, None]).long()
    D = (base + extra) * (torch.arange(S)[None] < src_len[:, None])
    p_norm = torch.rand(B, T, generator=g) * (torch.arange(T)[None] < mel_len[:, None])
    f0 = (80.0 + 300.0 * torch.rand(B, T, generator=g)) * (torch.arange(T)[None] < mel_len[:, None])
    spk = torch.randn(B, 512, generator=g)
    spk = spk / spk.norm(dim=1, keepdim=True)
    d = lambda t: t.to(dev)
    wav_d, n_d, text_d, src_d, D_d, p_d, f0_d, spk_d = d(wav), d(n), d(text), d(src_len), d(D), d(p_norm), d(f0), d(spk)
    frames = int(mel_len.sum())
    model = styler_amd.STYLER().to(dev).eval()
    rt.strict_inputs = False

    def timed(fn, k):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3

    res = {"workload": f"C5: ragged wav batch B={B} ({'/'.join(str(x) for x in lengths)} samples) -> STFT -> mel/energy -> "
                       f"STYLER.forward eval, teacher-forced, dual branch, S={S}, T={T}, valid frames={frames}",
           "steps": steps, "valid_frames": frames}
    flops_per_frame = 2.0 * 1024 * 1026 + 2.0 * 513 * 80
    with torch.no_grad():
        for prec in ("fp32", "bf16"):
            rt.set_precision(prec)
            fe = WavFrontEnd().to(dev)
            ms = timed(lambda: fe(wav_d, n_d), 2 * steps)
            padded = B * T                                    # the GEMM runs over the padded frame rectangle
            res[f"front_end_{prec}"] = {
                "ms": round(ms, 3), "value": round(frames / ms * 1e3, 1),
                "roofline": {"bound": "mfma", "kernel": "STFT framing GEMM + magnitude + mel projection (styler_stft_mel_varlen)",
                             "achieved": round(padded * flops_per_frame / (ms * 1e-3) / 1e12, 2),
                             "peak": MFMA_PEAK_TFLOPS[prec], "unit": "TFLOP/s",
                             "frac": round(padded * flops_per_frame / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[prec], 4),
                             "flops_model": "2*1024*1026 + 2*513*80 per frame of the padded [B, T] rectangle; time = the whole "
                                            "front end (reflect-pad/framing, GEMM, magnitude + energy, mel GEMM + log)"}}
        rt.set_precision(args.prec)
        fe = WavFrontEnd().to(dev)                            # features in fp32 arithmetic, the model in args.prec
        def e2e(front):
            rt.set_precision("fp32")
            feats = front(wav_d, n_d)
            rt.set_precision(args.prec)
            mel = feats["mel"]
            return model(text_d, mel, mel, p_d, feats["e_input"], src_d, feats["mel_len"], D_d, f0_d, feats["energy"], S, T,
                         speaker_embed=feats.get("speaker_embed", spk_d))
        ms = timed(lambda: e2e(fe), steps)
        res["wav_to_mel_given_speaker"] = {"ms_per_step": round(ms, 3), "value": round(frames / ms * 1e3, 1), "launch": "eager",
                                           "dtype": f"front end fp32, model {args.prec}"}
        # the same pipeline replayed from ONE hipGraph (the launch mode of the other legs; VERDICT round 4, weak #10 iii)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                e2e(fe)
            torch.cuda.current_stream().wait_stream(side)
            g5 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g5):
                e2e(fe)
            g5.replay()
            msg = timed(g5.replay, steps)
            res["wav_to_mel_given_speaker_graph"] = {"ms_per_step": round(msg, 3), "value": round(frames / msg * 1e3, 1),
                                                     "launch": "hipGraph replay", "dtype": f"front end fp32, model {args.prec}"}
            del g5
        except Exception as e:                      # (never the bench line's problem: the eager leg above stands)
            res["wav_to_mel_given_speaker_graph"] = {"error": f"{type(e).__name__}: {str(e)[:120]}"}
        fe_ds = WavFrontEnd(DeepSpeaker().to(dev)).to(dev)
        ms = timed(lambda: e2e(fe_ds), steps)
        res["wav_to_mel_deepspeaker"] = {"ms_per_step": round(ms, 3), "value": round(frames / ms * 1e3, 1), "launch": "eager",
                                         "dtype": f"front end + DeepSpeaker fp32, model {args.prec}",
                                         "note": "DeepSpeaker ResCNN random-init, parity unpinned (DESIGN 3.8)"}
    rt.set_precision(args.prec)
    del model
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` launches its own ranks (one process per GPU, train.py:33's DataParallel replaced):
        # re-exec this command under torch.distributed.run on a free local port; rank 0 of the children prints the JSON
        # line (the other ranks' stdout goes to stderr), this parent prints nothing and hands the exit code through
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        raise SystemExit(subprocess.call(cmd, env=env))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if rank != 0:
        sys.stdout.flush()
        os.dup2(2, 1)                                   # nothing but rank 0's JSON line belongs on the job's stdout
    if world > 1 or os.environ.get("STYLER_FORCE_PG"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if os.environ.get("STYLER_TEST_SHARED_GPU"):
            # test hook for 1-GPU boxes: every rank on cuda:0, collectives over gloo (RCCL refuses two ranks on one
            # device) -- exercises the N > 1 control flow (sharding, two-graph step, all-reduce between the replays,
            # max-over-ranks timing); the numbers it prints are NOT a scaling measurement
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist is not None and world > 1:
        # Fail LOUDLY before anything is timed unless every rank of the job is there on the expected transport: a 1-element SUM
        # all-reduce must count `world` ranks, on RCCL (backend "nccl") unless the shared-GPU test hook asked for gloo.  A job that
        # silently ran on fewer ranks, or on a fallback backend, would print a scaling number that means nothing.
        want = "gloo" if os.environ.get("STYLER_TEST_SHARED_GPU") else "nccl"
        if dist.get_backend() != want:
            raise SystemExit(f"bench.py --gpus {args.gpus}: process group backend is {dist.get_backend()!r}, expected {want!r}")
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)
        torch.cuda.synchronize()
        if int(seen.item()) != world or world != args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: the rank-count all-reduce saw {int(seen.item())} ranks "
                             f"(WORLD_SIZE={world}); refusing to time a partial job")

    main_res = run(args, args.mode, args.prec, rank, world, dev, dist)
    aux = {}
    if not args.no_aux:
        keep = args.steps, args.warmup, args.repeat
        args.steps, args.warmup, args.repeat = max(5, args.steps // 2), min(args.warmup, 3), 0
        legs = []
        if args.mode == "train" and args.prec == "bf16":
            legs.append(("train_fp32", "train", "fp32"))        # the 1e-3 parity arithmetic, same step
            legs.append(("train_bf16x3", "train", "bf16x3"))    # ... and the same bounds on the bf16 matrix cores (3 products)
        legs.append(("forward_c2", "fwd", args.prec) if args.mode == "train" else ("train_c3", "train", args.prec))
        for name, mode, prec in legs:
            r = run(args, mode, prec, rank, world, dev, dist, with_cpu=False,
                    with_roofline=(name not in ("train_fp32", "train_bf16x3")))
            if r is not None:
                aux[name] = {k: r[k] for k in ("value", "ms_per_step", "workload", "dtype", "launch", "roofline") if k in r}
                aux[name]["steps"] = args.steps
        args.steps, args.warmup, args.repeat = keep
        if world == 1 and args.mode == "train" and args.shape == "vctk":
            aux["forward_c4"] = forward_c4(args, dev)
            aux["forward_c5"] = forward_c5(args, dev)
    hbm = None
    if rank == 0 and world == 1 and not args.no_hbm and args.shape == "vctk":
        from closed_form import make_batch
        b0 = make_batch(args.batch, 20, 60, 2, 13, seed=1234)
        hbm = hbm_rooflines(dev, args.batch, b0["text"].shape[1], b0["mel_target"].shape[1], int(b0["mel_len"].sum()), "c3")
    if rank == 0:
        cfg = {"workload": main_res["workload"], "launch": main_res["launch"], "graphs": main_res["graphs"],
               "parallelism": f"dp{world}", "host_enqueue_ms_per_step": main_res.get("host_enqueue_ms_per_step")}
        if "allreduce" in main_res:
            cfg.update(main_res["allreduce"])
        line = {
            "metric": "mel_frames_per_sec", "value": main_res["value"],
            "unit": "valid mel-frames/s (80-bin mel, whole job)", "per_gpu": round(main_res["value"] / world, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.prec,
            "data": "synthetic (seeded VCTK-shape batch, random-init weights)",
            "config": cfg, "repeat": main_res.get("repeat"),
            "roofline": main_res.get("roofline"), "cpu_baseline": main_res.get("cpu_baseline")}
        if hbm is not None and line["roofline"] is not None:
            line["roofline"]["hbm"] = hbm
        if aux:
            line["aux"] = aux
            x3 = aux.get("train_bf16x3")
            if x3:
                # the SAME step in the arithmetic that meets north_star's 1e-3 bound (bf16x3: fp32-class products as three bf16
                # MFMA products), next to the throughput-mode headline (VERDICT round 4, next #4c).  Its roofline is
                # the whole step's: algorithmic FLOPs (265 MFLOP per valid frame of a train step; every GEMM product
                # costs three MFMA products in this mode, so the matrix cores execute ~3x that) over the bf16 dense peak.
                alg = x3["value"] * 265e6 / 1e12
                line["value_parity"], line["ms_per_step_parity"], line["dtype_parity"] = x3["value"], x3["ms_per_step"], "bf16x3"
                line["roofline_parity"] = {"bound": "mfma", "kernel": "whole bf16x3 train step (all GEMM engines, 3 bf16 products per product)",
                                           "achieved": round(alg, 1), "executed": round(3 * alg, 1), "peak": MFMA_PEAK_TFLOPS["bf16"],
                                           "unit": "TFLOP/s", "frac": round(alg / MFMA_PEAK_TFLOPS["bf16"], 4),
                                           "frac_executed": round(3 * alg / MFMA_PEAK_TFLOPS["bf16"], 4),
                                           "flops_model": "265 MFLOP per valid mel frame (SURVEY 8d, train step) x frames/s"}
    # The JSON line must be the LAST line of the job's stdout.  RCCL writes a version banner to the C-level stdout of every
    # process that creates a communicator; behind a pipe it sits in the stdio buffer until exit, i.e. it used to land AFTER
    # the line.  So: tear the group down first, push whatever C code buffered out, and only then print (ranks other than 0
    # had their stdout pointed at stderr in main()).
    if dist is not None:
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(line), flush=True)


def pmc_traffic(want, prec):
    """HBM-side bytes per launch of the dominant kernel family `want`, from the SEPARATE rocprofv3 --pmc passes of this same command
    (tools/pmc_run.sh -> tools/pmc_traffic.py, committed as profiles/r02_pmc_traffic.jsonl for the kernels of THIS round;
    FETCH_SIZE doubled per the gfx950 correction).  PMC collection cannot run inside the timed process, so the figure is
    read from that file; null when the file has no record for the kernel or was taken for another precision."""
    if prec != "bf16":
        return None, None
    for name in ("r06_pmc_traffic.jsonl", "r05_pmc_traffic.jsonl", "r04_pmc_traffic.jsonl", "r03_pmc_traffic.jsonl", "r02_pmc_traffic.jsonl"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        for line in open(path):
            try:
                rec = json.loads(line)
            except ValueError:
                continue
            if rec.get("label") == want:
                return rec["traffic_bytes_per_launch"], f"committed file profiles/{name} (separate rocprofv3 --pmc pass of this command)"
    return None, None


def cpu_baseline(model, batch, S, T, clean_only, thread_counts=(8, 16, 32), train=False, passes=5, budget_s=75.0):
    """The oracle (kind "port": plain PyTorch-CPU eager fp32 restatement of the reference) on the BENCH BATCH itself, timed on
    this host's cores as BASELINE.md section 3 lays out: both figures -- the eval forward (teacher-forced; dual-branch unless the
    config is clean-only) and the full train step = forward (both decodes) + DAT pass + ten losses + backward +
    clip_grad_norm_(1.0) + Adam (train.py:135-186) -- each as the MEDIAN of `passes` timed passes behind one warm-up pass.
    The intra-op thread count is probed first on a small sample (torch's pool stops scaling on these per-op shapes well
    before the core count of a 256-core host); `cores` = the threads the measurement used.  `value` is the figure of the
    headline workload (train step in train mode, else the forward).  A wall-clock budget bounds the leg: if a pass is slower
    than expected the pass count drops (never below 3) and `sample` says so."""
    from oracle import styler_oracle as O
    import statistics
    t_leg = time.perf_counter()
    sd0 = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    sb = {k: v.detach().cpu() for k, v in batch.items()}
    frames = int(sb["mel_len"].sum())
    n_items = int(sb["text"].shape[0])

    def fwd(b, S_, T_):
        with torch.no_grad():
            O.styler_forward(sd0, b["text"], b["mel_target"], b["mel_aug"], b["f0_norm"], b["energy_input"], b["src_len"],
                             b["mel_len"], b["D"], b["f0"], b["energy"], S_, T_, speaker_embed=b["speaker_embed"],
                             noisy_branch=not clean_only)

    # ---- thread-count probe: one forward of the first 8 utterances per candidate (after one untimed call)
    ncpu = os.cpu_count() or 1
    k = min(8, n_items)
    pb = {key: v[:k] for key, v in sb.items()}
    S2, T2 = int(pb["src_len"].max()), int(pb["mel_len"].max())
    for key in ("text", "D", "log_D"):
        pb[key] = pb[key][:, :S2]
    for key in ("mel_target", "mel_aug", "f0", "f0_norm", "f0_norm_aug", "energy", "energy_input", "energy_input_aug"):
        pb[key] = pb[key][:, :T2]
    probe = []
    for threads in sorted({min(t, ncpu) for t in thread_counts}):
        torch.set_num_threads(threads)
        fwd(pb, S2, T2)
        t0 = time.perf_counter()
        fwd(pb, S2, T2)
        probe.append((time.perf_counter() - t0, threads))
    threads = min(probe)[1]
    torch.set_num_threads(threads)

    def timed(fn, n, deadline):
        fn()                                             # warm-up pass
        ts = []
        for i in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if i + 1 >= 3 and time.perf_counter() + ts[-1] > deadline:
                break
        return statistics.median(ts), len(ts)

    res = {}
    dt, n = timed(lambda: fwd(sb, S, T), passes, t_leg + (0.3 if train else 1.0) * budget_s)
    res["fwd"] = {"value": round(frames / dt, 1), "s_per_pass": round(dt, 3), "timed_passes": n,
                  "what": f"eval forward, teacher-forced, {'clean' if clean_only else 'dual'}-branch"}
    if train:
        sd = {key: (v.clone().requires_grad_(True) if v.is_floating_point() and "position_enc" not in key and "_bins" not in key
                    and "running_" not in key else v) for key, v in sd0.items()}
        params = [v for v in sd.values() if v.requires_grad]
        import styler_amd.hparams as hp
        opt = torch.optim.Adam(params, lr=1e-4, betas=hp.betas, eps=hp.eps, weight_decay=hp.weight_decay)

        def step():
            opt.zero_grad(set_to_none=True)
            O.train_losses(sd, sb, training=True)[0].backward()
            torch.nn.utils.clip_grad_norm_(params, hp.grad_clip_thresh)
            opt.step()

        dt, n = timed(step, passes, t_leg + budget_s)
        res["train"] = {"value": round(frames / dt, 1), "s_per_pass": round(dt, 3), "timed_passes": n,
                        "what": "forward (both decodes) + DAT pass + 10 losses + backward + clip_grad_norm_ + Adam"}
    head = res["train" if train else "fwd"]
    tried = "; ".join(f"{t} threads {d * 1e3:.0f} ms" for d, t in sorted(probe, key=lambda x: x[1]))
    out = {"value": head["value"], "unit": "valid mel-frames/s", "cores": threads, "kind": "port",
           "sample": (f"the bench batch itself: {n_items} utterances, {frames} valid frames; 1 warm-up + {head['timed_passes']} timed "
                      f"passes, median; torch {torch.__version__} CPU fp32, {threads} intra-op threads on a {ncpu}-core host "
                      f"(probe, one forward of the first {k} utterances: {tried}); leg took {time.perf_counter() - t_leg:.0f} s")}
    out.update(res)
    return out


if __name__ == "__main__":
    main()

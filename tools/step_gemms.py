#!/usr/bin/env python
"""Every styler_conv_gemm launch of one eager training step (bench shape, bf16 mode): shape, engine, HIP-event time.

Aggregated per (shape, engine): launches per step, average us, TFLOP/s, share of the forward + dX GEMM time.  Side streams are
off (each launch alone on the chip); the empty-bracket cost of the box is subtracted.  `python tools/step_gemms.py [steps]`."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch


def main():
    import styler_amd
    from styler_amd import ops, rt
    from styler_amd.training import TrainState, add_pair_inputs, train_step
    from closed_form import make_batch
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = styler_amd.STYLER().to(dev).train()
    rt.set_precision("bf16")
    rt.strict_inputs = False
    batch = make_batch(48, 20, 60, 2, 13, seed=1234)
    frames = int(batch["mel_len"].sum())
    T = batch["mel_target"].shape[1]
    bd = {k: v.to(dev) for k, v in batch.items()}
    state = TrainState(model)
    add_pair_inputs(bd)
    rt.pred_stream = rt.text_stream = False
    for _ in range(3):
        train_step(model, state, bd)
    torch.cuda.synchronize()
    prof = ops.GemmProfiler()
    ops.gemm_profiler = prof
    for _ in range(steps):
        train_step(model, state, bd)
    torch.cuda.synchronize()
    ops.gemm_profiler = None
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for e0, e1 in pairs:
        e0.record(); e1.record()
    torch.cuda.synchronize()
    brk = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)[100]
    frac = frames / float(48 * T)
    recs = [r for r in prof.records if not isinstance(r[0], str)]
    assert len(recs) == len(prof.shapes)
    agg = {}
    for (var, flops, e0, e1, packed), shp in zip(recs, prof.shapes):
        d = agg.setdefault((shp, var, packed), [0, 0.0, 0.0])
        d[0] += 1
        d[1] += max(e0.elapsed_time(e1) - brk, 0.0)
        d[2] += flops * (frac if packed else 1.0)
    tot = sum(d[1] for d in agg.values())
    print(f"# bracket {brk * 1e3:.1f} us subtracted per launch; {steps} steps; forward + dX GEMM time {tot / steps:.3f} ms / step")
    print(f"# {'B':>3s} {'L':>6s} {'cin':>5s} {'n':>5s} kw io act      eng pk  n/step    avg_us   TFLOP/s   ms/step  share")
    for (shp, var, packed), d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        B, L, cin, n, kw, io, act = shp
        print(f"  {B:3d} {L:6d} {cin:5d} {n:5d} {kw:2d} {io:2d} {act:#8x} {var:3d} {int(packed):2d} {d[0] / steps:7.1f} {d[1] / d[0] * 1e3:9.1f} "
              f"{d[2] / (d[1] * 1e-3) / 1e12 if d[1] else 0:9.1f} {d[1] / steps:9.3f} {d[1] / tot * 100:6.1f}%")


if __name__ == "__main__":
    main()

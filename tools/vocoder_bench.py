"""HiFi-GAN generator throughput on one MI355X: samples/s and real-time factor per utterance
length, eager launches vs one hipGraph replay, with the CPU oracle timed on a short clip beside it.

    python tools/vocoder_bench.py [--frames 300 1000] [--batch 1] [--prec bf16] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def flops_per_sample(h):
    """Algorithmic FLOPs per output sample (2 per MAC): conv_pre + ups + resblocks + conv_post."""
    total_rate = 1
    for u in h["upsample_rates"]:
        total_rate *= u
    c = h["upsample_initial_channel"]
    f = 2.0 * 80 * c * 7 / total_rate
    rate = total_rate
    for u, k in zip(h["upsample_rates"], h["upsample_kernel_sizes"]):
        rate //= u                                   # samples per row after this stage = rate
        f += 2.0 * c * (c // 2) * k / u / rate       # transposed conv: k/u taps per output row
        c //= 2
        for ks, dil in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
            f += 2.0 * c * c * ks * 2 * len(dil) / rate
    return f + 2.0 * c * 7


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, nargs="+", default=[300, 1000])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--prec", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    from styler_amd import ops, utils
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    voc = utils.get_vocoder(device=dev)
    voc.prec = ops.PREC_BF16 if args.prec == "bf16" else ops.PREC_F32
    h = dict(voc.h)
    fps = flops_per_sample(h)
    for T in args.frames:
        mel = torch.randn(args.batch, 80, T, device=dev) * 2 - 4
        row = {"frames": T, "batch": args.batch, "prec": args.prec, "flop_per_sample": round(fps)}
        for mode in ("eager", "graph"):
            voc.use_graph = mode == "graph"
            for _ in range(3):
                wav = voc(mel)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                wav = voc(mel)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.iters * 1e3
            n = wav.numel()
            row[mode] = {"ms": round(ms, 3), "samples_per_s": round(n / ms * 1e3), "x_realtime": round(n / 22050 / ms * 1e3, 1),
                         "tflops": round(n * fps / ms / 1e9, 2)}
        print(json.dumps(row), flush=True)
    if not args.no_cpu:
        from oracle import styler_oracle as O                  # checker / CPU baseline only
        sd = {k: v.detach().cpu() for k, v in voc.state_dict().items()}
        mel = torch.randn(1, 80, 100) * 2 - 4
        torch.set_num_threads(min(16, os.cpu_count()))       # tiny convs oversubscribe a 100+-core host
        O.hifigan_generator(sd, mel[:, :, :10])
        t0 = time.perf_counter()
        with torch.no_grad():
            w = O.hifigan_generator(sd, mel)
        dt = time.perf_counter() - t0
        print(json.dumps({"cpu_oracle": {"frames": 100, "seconds": round(dt, 2), "samples_per_s": round(w.numel() / dt),
                                         "cores": min(16, os.cpu_count())}}), flush=True)


if __name__ == "__main__":
    main()

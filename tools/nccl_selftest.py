"""One-rank RCCL sanity check of the exact collectives bench.py / TrainState use (float64 MAX/SUM, fp32 SUM on a
bucket view of a flat buffer, barrier)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
flat = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
w = dist.all_reduce(flat[1000:500000], op=dist.ReduceOp.SUM, async_op=True); w.wait()
dist.barrier(); torch.cuda.synchronize()
print("rccl ok", float(t), float(flat[1000]))
dist.destroy_process_group()

"""One-rank RCCL smoke test of the calls bench.py's multi-GPU path makes (the N-rank run itself is the driver's):
process group on the nccl backend bound to cuda:0, gather / reduce / all_reduce(MAX) / barrier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29517"))
import torch
import torch.distributed as dist
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
send = torch.arange(12, dtype=torch.float32, device=dev).view(4, 3)
big = torch.zeros((4, 3), dtype=torch.float32, device=dev)
dist.gather(send, list(big.split(4)), dst=0)
assert torch.equal(big, send)
acc = torch.ones((8, 8, 4), device=dev)
dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 1.5 and bool((acc == 1).all())
dist.destroy_process_group()
print("rccl one-rank self-check ok:", torch.cuda.get_device_name(0))

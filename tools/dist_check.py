"""One-rank RCCL group on cuda:0: the torch.distributed calls bench.py makes when it runs on several GPUs
(init with device_id, barrier, float64 MAX all-reduce, all_gather_into_tensor, destroy)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29533"; os.environ["RANK"]="0"; os.environ["WORLD_SIZE"]="1"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out = torch.empty((4,), dtype=torch.float32, device=dev)
dist.all_gather_into_tensor(out, torch.arange(4, dtype=torch.float32, device=dev))
torch.cuda.synchronize()
print("ok", float(t.item()), out.tolist())
dist.destroy_process_group()

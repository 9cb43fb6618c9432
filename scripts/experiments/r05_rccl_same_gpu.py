"""Can RCCL carry two ranks that share ONE GPU?  (Would let the C-level RCCL communicator run with more than one rank on the one-GPU box.)"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    t = torch.full((1024,), float(rank + 1), device="cuda:0", dtype=torch.float64)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce ->", t[0].item(), flush=True)
    dist.destroy_process_group()
except Exception as ex:
    print("rank", rank, "FAILED:", repr(ex)[:600], flush=True)
    sys.exit(3)

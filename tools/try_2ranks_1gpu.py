import os, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl")
x = torch.full((4,), float(rank), device="cuda")
out = [torch.zeros_like(x) for _ in range(world)]
dist.all_gather(out, x)
print(rank, [o.tolist() for o in out], flush=True)
dist.destroy_process_group()

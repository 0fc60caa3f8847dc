import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29655")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.ones(4,device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
print("JSONLINE")
dist.destroy_process_group()

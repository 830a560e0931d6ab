import os, sys, json
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
import torch, torch.distributed as dist
import bench
log = f"/tmp/m360_rccl_probe_{os.getpid()}.log"
os.environ["NCCL_DEBUG"], os.environ["NCCL_DEBUG_SUBSYS"], os.environ["NCCL_DEBUG_FILE"] = "INFO", "INIT,GRAPH", log
dev = torch.device("cuda:0")
dist.init_process_group("nccl", device_id=dev, world_size=1, rank=0)
x = torch.ones(4, device=dev); dist.all_reduce(x); torch.cuda.synchronize()
print(json.dumps(bench.rccl_log_summary(log)))
print(open(log).read()[:1500] if os.path.exists(log) else "NO LOG FILE")
dist.destroy_process_group()

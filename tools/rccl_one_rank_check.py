import os, sys
sys.path.insert(0, "/root/repo")
os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29701", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
import torch, torch.distributed as dist
from vvc_amd import sharding
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
sharding.barrier(dist)
print("max", sharding.max_over_ranks(dist, [1.5, 2.5]))
print("gather", sharding.gather_floats(dist, 3.25))
print("sum", sharding.sum_over_ranks_u64(dist, 12345))
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank ok")

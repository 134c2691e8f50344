"""Can two ranks on ONE MI355X run the job's RCCL all-gather (backend "nccl")?  (The pool gives this build one GPU; DESIGN section 6 states
the RCCL path as unexecuted.)  Run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port <free port>
tools/rccl_same_gpu_probe.py      -- prints what RCCL says."""
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    x = torch.full((4, 18), float(rank), device="cuda:0")
    out = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(out, x)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_gather over RCCL on one GPU ok:", [float(o[0, 0]) for o in out], flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:400]}", flush=True)
    sys.exit(3)

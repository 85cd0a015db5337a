"""GPU box probe: can two ranks of one RCCL group share ONE GPU (so that a 1-GPU box could run bench.py --gpus 2 for real)?"""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((4,), float(dist.get_rank()), device="cuda")
    out = torch.empty(8, device="cuda")
    dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    print("RANK", dist.get_rank(), "OK", out.tolist())
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print("RANK", os.environ.get("RANK"), "FAILED", repr(e)[:300])
    sys.exit(1)

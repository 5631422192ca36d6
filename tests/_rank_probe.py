"""Rank process started by bench.launch_ranks in tests/test_cpu_launcher.py: joins a gloo group from the
environment the launcher set, all-reduces its rank and writes what it saw. Never touches a GPU."""
import json
import os
import sys

import torch
import torch.distributed as dist

out_dir, fail_rank = sys.argv[1], int(sys.argv[2])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if rank == fail_rank:
    sys.exit(7)
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank)])
dist.all_reduce(t)
with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
    json.dump({"rank": rank, "local_rank": int(os.environ["LOCAL_RANK"]), "world": dist.get_world_size(),
               "sum": float(t.item()), "addr": os.environ["MASTER_ADDR"], "launched": os.environ.get("VX_BENCH_LAUNCHED")}, f)
dist.barrier()
dist.destroy_process_group()

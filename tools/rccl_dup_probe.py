#!/usr/bin/env python3
"""Does RCCL on this box accept two ranks on ONE GPU (PSK_SHARE_GPU=1)?  If it does, the whole N = 2 RCCL path can be
exercised on a one-GPU box.  Run as two processes: RANK=0/1 WORLD_SIZE=2 LOCAL_RANK=0/1 PSK_SHARE_GPU=1."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phenotypeseeker_amd import dist  # noqa: E402

g = dist.Group()
try:
    g.init()
    print("rank", g.rank, "joined", g.backend, "sum", g.allreduce_sum(g.rank + 1), "gather", g.allgather_bytes(b"r%d" % g.rank), flush=True)
    g.close()
except Exception as e:   # noqa: BLE001
    print("rank", g.rank, "FAILED:", repr(e)[:300], flush=True)

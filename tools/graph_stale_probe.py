"""Diagnostic: inside a replayed HIP graph, does a kernel see a device word that was rewritten just before the replay
(by an H2D copy from pinned memory / by a fill kernel) -- through s_load, a vector load, an agent-scope load?"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import _lib as L  # noqa: E402

lib = ctypes.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
B = 1024
src = torch.zeros(1, dtype=torch.int64, device=dev)
outs = [torch.zeros(B, dtype=torch.int64, device=dev) for _ in range(3)]
pin = torch.zeros(1, dtype=torch.int64).pin_memory()
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731


def launch():
    rc = lib.vx_debug_echo(p(src), p(outs[0]), p(outs[1]), p(outs[2]), B,
                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0


big = torch.zeros(64 << 20, dtype=torch.float32, device=dev)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    launch()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    launch()
    big.add_(1.0)          # some unrelated traffic inside the graph
    launch()
for how in ("h2d", "fill", "h2d", "fill"):
    stale = [0, 0, 0]
    for it in range(1, 41):
        val = 1000 * it + (7 if how == "h2d" else 3)
        if how == "h2d":
            pin[0] = val
            src.copy_(pin, non_blocking=True)
        else:
            src.fill_(val)
        g.replay()
        torch.cuda.synchronize()
        for j in range(3):
            stale[j] += int((outs[j] != val).sum())
    print(f"update by {how}: stale words of {40 * B}: scalar {stale[0]}, vector {stale[1]}, agent {stale[2]}", flush=True)
# the same without a graph
for how in ("h2d", "fill"):
    stale = [0, 0, 0]
    for it in range(1, 41):
        val = 5000 * it + 1
        if how == "h2d":
            pin[0] = val
            src.copy_(pin, non_blocking=True)
        else:
            src.fill_(val)
        launch()
        torch.cuda.synchronize()
        for j in range(3):
            stale[j] += int((outs[j] != val).sum())
    print(f"eager, update by {how}: stale scalar {stale[0]}, vector {stale[1]}, agent {stale[2]}", flush=True)

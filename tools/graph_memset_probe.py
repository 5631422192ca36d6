"""Diagnostic: does a hipMemsetAsync captured in a HIP graph run (in order) on every replay?"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import _lib as L  # noqa: E402

lib = ctypes.CDLL(L.LIB_PATH)
lib.vx_debug_memset.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")


def memset(t, value, nbytes):
    rc = lib.vx_debug_memset(ctypes.c_void_p(t.data_ptr()), value, nbytes,
                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0


for nbytes in (1024, 4 << 20, (4 << 20) + 4, 100 << 20, 300 << 20):
    for value in (0, 0xff):
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            memset(buf, value, nbytes)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            buf.add_(1)                  # kernel before (makes the buffer differ from the fill)
            memset(buf, value, nbytes)   # the node under test
            out.copy_(buf)               # kernel after: must see the fill
        bad = []
        for it in range(3):
            buf.fill_(5)
            out.fill_(9)
            g.replay()
            torch.cuda.synchronize()
            bad.append((int((buf != value).sum()), int((out != value).sum())))
        print(f"memset {nbytes} bytes value {value:#x}: wrong bytes (buffer, reader) per replay {bad}", flush=True)
        del g, buf, out

"""Time the dense tail (necks + attention + head, stock PyTorch-ROCm ops) on VISTA-shaped maps.
usage: python tools/dense_tail_probe.py [batch] [fp32|bf16] [channels_last]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vista_amd.dense_tail import OHS_Multiview

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dtype = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
cl = len(sys.argv) > 3 and sys.argv[3] == "channels_last"
dev = torch.device("cuda:0")
torch.manual_seed(0)
t0 = time.perf_counter()
det = OHS_Multiview().to(dev).eval().to(dtype)
bev = torch.randn(batch, 640, 160, 160, device=dev, dtype=dtype)
rv = torch.randn(batch, 640, 160, 12, device=dev, dtype=dtype)
if cl:
    det = det.to(memory_format=torch.channels_last)
    bev, rv = bev.contiguous(memory_format=torch.channels_last), rv.contiguous(memory_format=torch.channels_last)
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    with torch.no_grad():
        t1 = time.perf_counter()
        out = det.dense_tail(bev, rv)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"benchmark={bench} first call {t2 - t1:.2f} s", flush=True)
        for _ in range(2):
            det.dense_tail(bev, rv)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(5):
            det.dense_tail(bev, rv)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
    print(f"benchmark={bench} batch={batch} {dtype} channels_last={cl}: {(t4 - t3) / 5 * 1e3:.2f} ms/step "
          f"({(t4 - t3) / 5 / batch * 1e3:.2f} ms/frame)", flush=True)
print([tuple(o["box_preds"].shape) for o in out][:2], "setup", f"{t1 - t0:.1f}s")

"""Time the rotated BEV IoU / NMS kernels. usage: python tools/nms_probe.py [n_boxes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vista_amd import iou3d_nms as m

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(0)
c = rng.uniform(-30, 30, (n, 2)); wh = rng.uniform(0.6, 5.0, (n, 2)); a = rng.uniform(-np.pi, np.pi, (n, 1))
boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2, a], 1).astype(np.float32)).cuda()
iou = torch.empty((n, n), device="cuda"); keep = torch.empty(n, dtype=torch.int64, device="cuda")
for name, fn in (("iou matrix", lambda: m.boxes_iou_bev_gpu(boxes, boxes, iou)), ("nms (mask + scan + count sync)", lambda: m.nms_gpu(boxes, keep, 0.2))):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): r = fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: n={n} {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (result {r})")

"""Diagnostic: per-layer weight-gradient error of the GPU training step (module path with torch BatchNorm, fused
libvx BatchNorm path) and of the fp32 CPU oracle, all against the fp64 oracle (2 SMALL frames), beside the number of
ReLU gates at or above each layer that differ from the fp64 run's: the error is at fp32 rounding (1e-6) down to the
first flipped gate and steps up by three to four orders of magnitude there -- in the CPU oracle as on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_a_parity import _gpu_train_step, _oracle_train_step, _tiny_frames
from vista_amd.synth import backbone_state_dict
dev = torch.device("cuda:0")
clouds, mean, coords, grid, _ = _tiny_frames(dev, 2)
sd = backbone_state_dict("SpMiddleResNetFHD")
runs = {}
for name in ("module", "fused"):
    net, out, loss, gates = _gpu_train_step(sd, mean, coords, grid, dev, fused=name == "fused")
    runs[name] = ({n: p.grad.cpu().numpy().astype(np.float64) for n, p in net.named_parameters()}, gates)
numel = out.numel()
gates32, gates64 = [], []
_, g32 = _oracle_train_step(sd, mean, coords, grid, numel, torch.float32, gates32)
_, g64 = _oracle_train_step(sd, mean, coords, grid, numel, torch.float64, gates64)


def flips(gates):
    out, later = {}, 0
    for (name, m64), m in zip(gates64[::-1], gates[::-1]):
        later += int((m64 != (m[1] if isinstance(m, tuple) else m)).sum())
        out[name] = later
    return out


f32, fm, ff = flips(gates32), flips(runs["module"][1]), flips(runs["fused"][1])
print(f"{'layer':28s} {'cpu fp32':>9s} {'flips':>6s} {'gpu module':>10s} {'flips':>6s} {'gpu fused':>10s} {'flips':>6s}   (max abs error / max |g64|)")
for n in g64:
    if n.endswith("weight") and g64[n].ndim == 5:
        s = np.abs(g64[n]).max()
        e = lambda g: np.abs(g[n] - g64[n]).max() / s  # noqa: E731
        print(f"{n:28s} {e(g32):9.1e} {f32[n]:6d} {e(runs['module'][0]):10.1e} {fm[n]:6d} {e(runs['fused'][0]):10.1e} {ff[n]:6d}")

import sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import spconv_oracle as so, voxelize_np as ov
from vista_amd import models
from vista_amd.synth import backbone_state_dict, lidar_cloud, GEOMETRIES, grid_size
rg, vs, T, mv = GEOMETRIES["SMALL"]
clouds = [lidar_cloud(seed=20 + b, sweeps=2) for b in range(2)]
ovx = [ov.voxelize(c, vs, rg, T, mv) for c in clouds]
mean = np.concatenate([ov.mean_vfe(v, n) for v, c, n in ovx]); coords = ov.collate_coords([c for v, c, n in ovx])
grid = grid_size(rg, vs).tolist()
sd = backbone_state_dict("SpMiddleResNetFHD")
def run(dtype):
    ref = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5), sd).to(dtype).train()
    x = torch.from_numpy(mean).to(dtype); c = coords
    shape = [int(v) for v in (np.array(grid[::-1]) + [1, 0, 0])]
    cache = {}
    def conv64(x, w, nbr):
        K = nbr.shape[0]; wk = w.reshape(w.shape[0], K, w.shape[-1])
        y = torch.zeros(nbr.shape[1], w.shape[0], dtype=dtype)
        for k in range(K):
            o = np.flatnonzero(nbr[k] >= 0)
            if o.size:
                y = y.index_add(0, torch.from_numpy(o), x[torch.from_numpy(nbr[k][o]).long()] @ wk[:, k, :].T)
        return y
    def run_conv(conv, x, c, shape):
        if conv.subm:
            key = (conv.indice_key, c.shape[0])
            if key not in cache: cache[key] = so.rulebook_subm(c, shape, conv.kernel_size)
            y = conv64(x, conv.weight, cache[key]); oc, osh = c, shape
        else:
            oc, nbr, osh = so.rulebook_conv(c, shape, conv.kernel_size, conv.stride, conv.padding)
            y = conv64(x, conv.weight, nbr)
        if conv.bias is not None: y = y + conv.bias
        return y, oc, osh
    for m in ref.middle_conv:
        if isinstance(m, models.SparseBasicBlock):
            idt = x
            y, _, _ = run_conv(m.conv1, x, c, shape); y = torch.relu(m.bn1(y))
            y, _, _ = run_conv(m.conv2, y, c, shape); x = torch.relu(m.bn2(y) + idt)
        elif hasattr(m, "subm"): x, c, shape = run_conv(m, x, c, shape)
        else: x = m(x)
    loss = (x ** 2).sum() / (2*256*16*16); loss.backward()
    return loss.item(), {n: p.grad.double() for n, p in ref.named_parameters()}
l32, g32 = run(torch.float32); l64, g64 = run(torch.float64)
print(l32, l64)
for n in g64:
    if n.endswith("conv1.bias") or n.endswith("conv2.bias"): continue
    print(f"{n:40s} fp32-vs-fp64 relmax {float((g32[n]-g64[n]).abs().max()/g64[n].abs().max()):.3e}")

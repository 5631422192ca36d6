"""Generate tests/golden/attention_*.npz by running the REFERENCE's own cross-view attention neck
(det3d/models/necks/attention.py) on CPU in the build container. The file is loaded by path; its one
package-relative import (`from ..registry import NECKS`, a decorator registry) is satisfied by a stand-in
whose register_module returns the class unchanged, so the arithmetic that runs is the reference's.
The fixture holds inputs, the module's state_dict and its outputs (data only).
Usage: python tools/gen_golden_attention.py   (needs /root/reference; never runs on the GPU box)"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_reference_attention():
    for name in ("det3d", "det3d.models", "det3d.models.necks"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    reg = types.ModuleType("det3d.models.registry")

    class _Reg:
        @staticmethod
        def register_module(cls):
            return cls
    reg.NECKS = _Reg()
    sys.modules["det3d.models.registry"] = reg
    spec = importlib.util.spec_from_file_location("det3d.models.necks.attention",
                                                  os.path.join(REF, "det3d/models/necks/attention.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    att = load_reference_attention()
    cases = {
        "small_1head": dict(embed_dim=32, num_heads=1, bev_size=(16, 16), bev_block_res=(4, 4), rv_size=(16, 3),
                            rv_block_res=(4, 3), hidden_channels=48, batch=2, seed=1),
        "small_2head": dict(embed_dim=48, num_heads=2, bev_size=(24, 16), bev_block_res=(6, 8), rv_size=(24, 4),
                            rv_block_res=(6, 2), hidden_channels=40, batch=3, seed=2),
    }
    for name, c in cases.items():
        torch.manual_seed(c["seed"])
        m = att.Cross_Attention_Decouple(c["embed_dim"], c["embed_dim"], c["embed_dim"], c["num_heads"], c["bev_size"],
                                         c["bev_block_res"], c["rv_size"], c["rv_block_res"], c["hidden_channels"]).eval()
        # non-trivial LayerNorm / bias parameters
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.05 * torch.randn_like(p))
        bev = torch.randn(c["batch"], c["embed_dim"], *c["bev_size"])
        rv = torch.randn(c["batch"], c["embed_dim"], *c["rv_size"])
        with torch.no_grad():
            outs, maps = m((bev.clone(), rv.clone()))
        data = {"bev": bev.numpy(), "rv": rv.numpy(), "out_sem": outs[0].numpy(), "out_geo": outs[1].numpy(),
                "map_sem": maps[0].numpy(), "map_geo": maps[1].numpy(),
                "cfg": np.array(repr({k: v for k, v in c.items() if k not in ("batch", "seed")}))}
        for k, v in m.state_dict().items():
            data["sd/" + k] = v.numpy()
        np.savez_compressed(os.path.join(OUT, f"attention_{name}.npz"), **data)
        print(name, {k: v.shape for k, v in data.items() if not k.startswith("sd/")})


if __name__ == "__main__":
    main()

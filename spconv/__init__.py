"""Drop-in `spconv` package for the VISTA sparse backbone path, backed by libvx (gfx950 HIP).

The reference imports `spconv` at det3d/models/detectors/voxelnet.py:7 (`spconv.SparseConvTensor`)
and `spconv.pytorch` at det3d/models/backbones/scn.py:4-6, multiview.py:8-9. Only the surface those
call sites use is provided; there is no CUDA path, no backend dispatch and no CPU fallback.
"""
from . import pytorch  # noqa: F401
from .pytorch import (SparseConv3d, SparseConvTensor, SparseModule,  # noqa: F401
                      SparseSequential, SubMConv3d)

__version__ = "2.1.0+vx"

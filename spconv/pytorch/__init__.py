from . import ops  # noqa: F401  (scn.py:5 imports the module; the reference never calls into it)
from .conv import SparseConv3d, SparseConvolution, SubMConv3d  # noqa: F401
from .core import SparseConvTensor  # noqa: F401
from .modules import SparseModule, SparseSequential, is_spconv_module  # noqa: F401

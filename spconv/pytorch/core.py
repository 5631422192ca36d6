"""SparseConvTensor: the container the reference builds at scn.py:373 / multiview.py:94."""
import numpy as np
import torch

from vista_amd import ops as vx


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None,
                 indice_dict=None, benchmark=False):
        """features [N, C]; indices int32 [N, 4] = (batch, z, y, x); spatial_shape = (D, H, W)
        (a numpy int64 array at scn.py:370); batch_size a Python int."""
        if features.dim() != 2:
            raise ValueError(f"features must be [N, C], got {tuple(features.shape)}")
        if indices.dim() != 2 or indices.shape[1] != 4:
            raise ValueError(f"indices must be [N, 4], got {tuple(indices.shape)}")
        if indices.dtype != torch.int32:
            raise ValueError("indices must be int32")
        if indices.shape[0] != features.shape[0]:
            raise ValueError("features and indices disagree on N")
        if indices.device != features.device:
            raise ValueError("features and indices must live on the same device")
        self._features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in np.asarray(spatial_shape).reshape(-1)]
        if len(self.spatial_shape) != 3:
            raise ValueError("only 3-D sparse tensors are supported")
        self.batch_size = int(batch_size)
        self.indice_dict = {} if indice_dict is None else indice_dict
        self.grid = grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark
        self.benchmark_record = {}
        # libvx extra: rank table of `indices` (set by strided convs, whose outputs are sorted rows);
        # lets SubMConv3d build its rulebook without a hash table. None for user-built tensors.
        self.rank_table = None

    @property
    def features(self):
        return self._features

    @features.setter
    def features(self, val):
        raise ValueError("you can't set features directly, use 'x = x.replace_feature(your_new_feature)' "
                         "to generate a new SparseConvTensor instead")

    def replace_feature(self, feature):
        """Shallow copy that shares indices and the rulebook cache."""
        t = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.grid,
                             self.voxel_num, self.indice_dict, self.benchmark)
        t.benchmark_record = self.benchmark_record
        t.rank_table = self.rank_table
        return t

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    @property
    def sparity(self):
        return self.indices.shape[0] / self.spatial_size / self.batch_size

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        out = _ToDense.apply(self._features, self.indices, self.batch_size, tuple(self.spatial_shape))
        if not channels_first:
            out = out.permute(0, 2, 3, 4, 1).contiguous()
        return out

    @property
    def shape(self):
        return [self.batch_size, *self.spatial_shape, self._features.shape[1]]

    def __dir__(self):  # scn.py:21 probes for "replace_feature" through __dir__()
        return list(super().__dir__())


class _ToDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices, batch_size, spatial_shape):
        ctx.save_for_backward(indices)
        ctx.spatial_shape = spatial_shape
        return vx.to_dense(features, indices, batch_size, spatial_shape)

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        return vx.from_dense(grad, idx, ctx.spatial_shape), None, None, None

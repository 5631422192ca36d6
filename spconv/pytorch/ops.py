"""Functional entry points (the reference imports this module at scn.py:5 but never calls it)."""
from vista_amd.ops import (conv_dgrad, conv_fwd, conv_out_shape, nbr_to_pairs,  # noqa: F401
                           rulebook_conv, rulebook_subm, to_dense)


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation=None):
    return conv_out_shape(input_size, kernel_size, stride, padding)


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     out_padding=0, subm=False, transpose=False, **_):
    """-> (out_indices, indice_pairs int32[K,2,n_out], indice_pair_num int32[K])."""
    if transpose or (dilation not in (1, [1, 1, 1], (1, 1, 1))):
        raise NotImplementedError("transposed / dilated sparse convolutions are outside the VISTA path")
    rb = (rulebook_subm(indices, batch_size, spatial_shape, ksize) if subm
          else rulebook_conv(indices, batch_size, spatial_shape, ksize, stride, padding))
    pairs, num = rb.pairs()
    return rb.out_indices, pairs, num

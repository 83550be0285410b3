"""
elasticdeform_amd -- MI355X-native elastic grid deformation.

Drop-in for the hot path of gvtulder/elasticdeform: ``deform_random_grid``, ``deform_grid`` and
``deform_grid_gradient`` (reference: elasticdeform/__init__.py:1) with the same signatures, backed
by hand-written HIP kernels for gfx950 behind the C ABI in include/edhip.h.
``elasticdeform_amd.torch`` is the on-device PyTorch autograd wrapper (reference:
elasticdeform/torch.py).

    import elasticdeform_amd as elasticdeform
    Y = elasticdeform.deform_random_grid(X, sigma=25, points=3)

As in the reference, the function ``deform_grid`` shadows the submodule of the same name.
``deform_grid_batch`` / ``deform_grid_gradient_batch`` (no counterpart in the reference) take a
batch of samples with one control grid each.
"""
from .deform_grid import (deform_grid, deform_grid_gradient, deform_random_grid,  # noqa: F401
                          deform_grid_batch, deform_grid_gradient_batch, set_arithmetic,
                          set_reduced_precision, set_crop_identity, set_gradient_accumulation,
                          set_field_strength)

from ._lib import release_scratch  # noqa: F401,E402  (frees the library's cached device scratch)

__version__ = '0.1.0'

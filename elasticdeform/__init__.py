"""
``import elasticdeform`` -- the reference's import name (/root/reference/elasticdeform/__init__.py:1)
served by the MI355X-native build: a thin alias of :mod:`elasticdeform_amd`, so that code written
against gvtulder/elasticdeform runs unchanged on the HIP kernels::

    import elasticdeform
    Y = elasticdeform.deform_random_grid(X, sigma=25, points=3)
    import elasticdeform.torch as etorch          # on-device autograd wrapper

As in the reference, the function ``deform_grid`` shadows the submodule of the same name.
There is no CPU fallback behind this name either.
"""
from elasticdeform_amd import (deform_grid, deform_grid_gradient, deform_random_grid,  # noqa: F401
                               deform_grid_batch, deform_grid_gradient_batch, set_arithmetic,
                               set_reduced_precision, set_crop_identity, set_gradient_accumulation, set_field_strength,
                               release_scratch, __version__)

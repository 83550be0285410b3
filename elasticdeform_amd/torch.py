"""
PyTorch wrapper: ``elasticdeform_amd.torch.deform_grid(X, displacement, *args, **kwargs)`` --
drop-in for ``elasticdeform.torch.deform_grid`` (/root/reference/elasticdeform/torch.py:33-66).

Same autograd contract as the reference's ``ElasticDeform`` Function (torch.py:5-29): gradients
flow to the inputs ``X`` only (the displacement gets none), a list / tuple of inputs gives a
tuple of outputs.  The difference is the one this build exists for: the reference copies every
tensor to the host, runs one CPU thread and copies back (torch.py:13-16,25-29); here CUDA tensors
stay in HBM and forward / backward are HIP kernels enqueued on the current stream.
"""
from __future__ import absolute_import

import torch

# the package re-exports the functions, and `deform_grid` the function shadows the submodule
from . import deform_grid as _deform_grid_fn
from . import deform_grid_gradient as _deform_grid_gradient_fn
from . import deform_grid_batch as _deform_grid_batch_fn
from . import deform_grid_gradient_batch as _deform_grid_gradient_batch_fn


class ElasticDeform(torch.autograd.Function):
    """forward: deform_grid; backward: deform_grid_gradient (torch.py:5-29)."""

    @staticmethod
    def forward(ctx, displacement, deform_args, deform_kwargs, *xs):
        ctx.save_for_backward(displacement)
        ctx.deform_args = deform_args
        ctx.deform_kwargs = deform_kwargs
        ctx.x_shapes = [tuple(x.shape) for x in xs]
        ys = _deform_grid_fn([x.detach() for x in xs], displacement.detach(),
                             *deform_args, **deform_kwargs)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        displacement, = ctx.saved_tensors
        dxs = _deform_grid_gradient_fn([dy.detach() for dy in dys], displacement.detach(),
                                       *ctx.deform_args, X_shape=ctx.x_shapes,
                                       **ctx.deform_kwargs)
        return (None, None, None) + tuple(dxs)


def deform_grid(X, displacement, *args, **kwargs):
    """
    Elastic deformation with a deformation grid, wrapped for PyTorch with a custom gradient.

    X : torch.Tensor or list / tuple of torch.Tensors; displacement : tensor or array of control
    point displacements; remaining arguments as for ``elasticdeform_amd.deform_grid``.
    Returns a tensor, or a tuple of tensors for a list / tuple input (torch.py:56-66).
    """
    if not isinstance(X, (list, tuple)):
        X_list = [X]
    else:
        X_list = X
    displacement = torch.as_tensor(displacement)
    y = ElasticDeform.apply(displacement, args, kwargs, *X_list)
    if isinstance(X, (list, tuple)):
        return y
    else:
        return y[0]


def random_displacement(naxis, points=3, sigma=25, batch=None, device=None, dtype=torch.float64,
                        generator=None):
    """
    Random control-point displacements drawn ON THE DEVICE: ``randn(naxis, *points) * sigma``, the
    distribution of ``deform_random_grid`` (/root/reference/elasticdeform/deform_grid.py:42-48),
    from torch's device generator (Philox) instead of NumPy's host RNG -- no host round trip and
    no H2D copy per sample.  With ``batch=B`` the result has a leading batch axis: one grid per
    sample of a batch.
    """
    if not isinstance(points, (list, tuple)):
        points = [points] * naxis
    assert len(points) == naxis
    shape = (naxis,) + tuple(int(p) for p in points)
    if batch is not None:
        shape = (int(batch),) + shape
    return torch.randn(shape, device=device, dtype=dtype, generator=generator) * sigma


def deform_random_grid(X, sigma=25, points=3, order=3, mode='constant', cval=0.0, crop=None,
                       prefilter=True, axis=None, affine=None, rotate=None, zoom=None,
                       generator=None):
    """
    ``elasticdeform.deform_random_grid`` (deform_grid.py:6-49) for tensors that live on the GPU:
    same arguments and meaning, but the random grid is drawn on the device of ``X``
    (:func:`random_displacement`) and the result stays there, with the autograd contract of
    :func:`deform_grid`.  ``generator``: an optional ``torch.Generator`` of that device.
    """
    from . import _host
    Xs = list(X) if isinstance(X, (list, tuple)) else [X]
    _, deform_shape = _host.normalize_axis_list(axis, Xs)
    displacement = random_displacement(len(deform_shape), points, sigma, device=Xs[0].device,
                                       generator=generator)
    import importlib
    _dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    pts = points if isinstance(points, (list, tuple)) else [points] * len(deform_shape)
    with _dgm._random_grid_hint(sigma, pts, deform_shape):       # (a strong field by construction -> the z-walk route)
        return deform_grid(X, displacement, order, mode, cval, crop, prefilter, axis, affine, rotate, zoom)


class ElasticDeformBatch(torch.autograd.Function):
    """deform_grid_batch / deform_grid_gradient_batch as an autograd pair: gradients flow to X."""

    @staticmethod
    def forward(ctx, x, displacements, deform_kwargs):
        ctx.save_for_backward(displacements)
        ctx.deform_kwargs = deform_kwargs
        ctx.x_shape = tuple(x.shape[1:])
        return _deform_grid_batch_fn(x.detach(), displacements.detach(), **deform_kwargs)

    @staticmethod
    def backward(ctx, dy):
        displacements, = ctx.saved_tensors
        dx = _deform_grid_gradient_batch_fn(dy.detach(), displacements.detach(), X_shape=ctx.x_shape,
                                            **ctx.deform_kwargs)
        return dx, None, None


def deform_grid_batch(X, displacements, **kwargs):
    """
    Batched :func:`deform_grid` with one control grid per sample: ``X`` is ``(B, ...)``,
    ``displacements`` is ``(B, naxis, n_0, ...)`` (e.g. from :func:`random_displacement` with
    ``batch=B``); keyword arguments as for ``elasticdeform_amd.deform_grid_batch``.  Differentiable
    with respect to ``X``.
    """
    return ElasticDeformBatch.apply(X, torch.as_tensor(displacements, device=X.device), kwargs)


def deform_random_grid_batch(X, sigma=25, points=3, axis=None, generator=None, **kwargs):
    """Per-sample random deformation of a batch ``X`` of shape ``(B, ...)``: draws ``B`` grids on
    the device and applies one to each sample (the augmentation step of a data loader, without a
    host round trip).  ``axis`` counts the axes of one sample."""
    from . import _host
    _, deform_shape = _host.normalize_axis_list(axis, [X[0]])
    disp = random_displacement(len(deform_shape), points, sigma, batch=X.shape[0], device=X.device,
                               generator=generator)
    import importlib
    _dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    pts = points if isinstance(points, (list, tuple)) else [points] * len(deform_shape)
    with _dgm._random_grid_hint(sigma, pts, deform_shape):
        return deform_grid_batch(X, disp, axis=axis, **kwargs)

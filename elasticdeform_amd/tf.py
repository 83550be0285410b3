"""
TensorFlow wrapper of the hot path (reference: elasticdeform/tf.py:5-72):
``deform_grid(X, displacement, *args, **kwargs)`` as a TensorFlow op with a custom gradient -- the forward pass is
:func:`elasticdeform_amd.deform_grid`, the gradient with respect to the images is
:func:`elasticdeform_amd.deform_grid_gradient`, and, as in the reference, there is no gradient with respect to the
displacement (it comes back as NaN of the displacement's shape, elasticdeform/tf.py:51).

TensorFlow (TF-ROCm) is NOT part of the image this build is made and tested in: importing this module without
TensorFlow raises ImportError, exactly like the reference's module.  The control flow is exercised by
tests/test_tf_wrapper.py against a minimal stand-in for the four TensorFlow entry points used here
(``custom_gradient``, ``py_function``, ``experimental.dlpack``, eager tensors with ``.numpy()``); it has never run under a
real TensorFlow.

Data movement: inside ``tf.py_function`` the arguments are eager tensors.  A tensor that lives on the GPU is handed to
the kernels without leaving HBM (DLPack -> torch CUDA tensor, and back the same way); whatever cannot go that way
(CPU tensors, a TensorFlow without DLPack) takes the numpy route -- one H2D and one D2H copy, the reference's own
path (``x.numpy()``, elasticdeform/tf.py:33-35).
"""
import numpy
import tensorflow

from . import deform_grid as _deform_grid_fn            # (the function shadows the submodule, as in the reference)
from . import deform_grid_gradient as _deform_grid_gradient_fn


def _dlpack():
    exp = getattr(tensorflow, 'experimental', None)
    return getattr(exp, 'dlpack', None) if exp is not None else None


def _unwrap(t):
    """eager tensor -> (array for the kernels, True when it went by DLPack): a torch CUDA tensor that shares the
    TensorFlow tensor's memory where that is possible, else a numpy array."""
    dl = _dlpack()
    device = str(getattr(t, 'device', '') or '')
    if dl is not None and 'GPU' in device.upper():
        try:
            import torch.utils.dlpack
            return torch.utils.dlpack.from_dlpack(dl.to_dlpack(t)), True
        except Exception:       # noqa: BLE001 -- any failure of the zero-copy route: the numpy route is always there
            pass
    return (t.numpy() if hasattr(t, 'numpy') else numpy.asarray(t)), False


def _wrap(a):
    """result of the kernels -> something tf.py_function accepts: a TensorFlow tensor sharing a torch CUDA tensor's
    memory (DLPack), or the numpy array as it is."""
    if isinstance(a, numpy.ndarray):
        return a
    dl = _dlpack()
    if dl is not None and getattr(a, 'is_cuda', False):
        try:
            import torch.utils.dlpack
            return dl.from_dlpack(torch.utils.dlpack.to_dlpack(a.contiguous()))
        except Exception:       # noqa: BLE001
            pass
    return a.cpu().numpy()


def _uniform(arrays, displacement):
    """The kernels want one array family per call: if anything took the numpy route, everything does."""
    vals, flags = zip(*arrays) if arrays else ((), ())
    d, dflag = displacement
    if all(flags) and dflag:
        return list(vals), d
    to_np = lambda v: v if isinstance(v, numpy.ndarray) else v.cpu().numpy()      # noqa: E731
    return [to_np(v) for v in vals], to_np(d)


def deform_grid(X, displacement, *args, **kwargs):
    """
    Elastic deformation with a deformation grid, as a TensorFlow op with a custom gradient
    (elasticdeform/tf.py:5-72).  ``X``: a tensor or a list of tensors; ``displacement``: a tensor or a numpy array;
    every other argument as in :func:`elasticdeform_amd.deform_grid`.  Returns a tensor, or a list for a list.
    """
    single = not isinstance(X, (list, tuple))
    volumes = (X,) if single else tuple(X)
    n = len(volumes)

    @tensorflow.custom_gradient
    def op(disp, *vols):
        def forward(disp_t, *vol_t):
            xs, d = _uniform([_unwrap(v) for v in vol_t], _unwrap(disp_t))
            outs = _deform_grid_fn(xs, d, *args, **kwargs)
            return [_wrap(o) for o in outs]

        ys = tensorflow.py_function(forward, (disp,) + tuple(vols), [v.dtype for v in vols], name='EdhipDeformGrid')

        def backward(*dys):
            def gradient(*packed):
                dy_t, disp_t, vol_t = packed[:n], packed[n], packed[n + 1:]
                shapes = [tuple(int(s) for s in v.shape) for v in vol_t]
                gs, d = _uniform([_unwrap(g) for g in dy_t], _unwrap(disp_t))
                dxs = _deform_grid_gradient_fn(gs, d, *args, X_shape=shapes, **kwargs)
                d_np = d if isinstance(d, numpy.ndarray) else d.cpu().numpy()
                # no gradient with respect to the control points: NaN of their shape, as the reference returns it
                return [numpy.nan * d_np] + [_wrap(g) for g in dxs]

            return tensorflow.py_function(gradient, tuple(dys) + (disp,) + tuple(vols),
                                          [disp.dtype] + [v.dtype for v in vols], name='EdhipDeformGridGrad')

        return ys, backward

    result = op(displacement, *volumes)
    return result[0] if single else result

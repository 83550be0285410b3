"""elasticdeform_amd.tf (reference: elasticdeform/tf.py:5-72, tests/test_deform_grid.py:367-468).

TensorFlow is not in this image.  Without it the module must fail to import, like the reference's (CPU test); its
control flow -- custom gradient, the two py_function bodies, argument packing, list / single returns, the NaN gradient
of the displacement -- runs here against a MINIMAL STAND-IN for the TensorFlow entry points it uses, with the product's
GPU compute underneath.  With a real TensorFlow installed the last test runs the wrapper under it instead."""
import importlib
import sys
import types

import numpy as np
import pytest

try:
    import tensorflow as _real_tf            # noqa: F401
    HAVE_TF = True
except Exception:                            # noqa: BLE001
    HAVE_TF = False


@pytest.mark.skipif(HAVE_TF, reason="TensorFlow is installed")
def test_import_without_tensorflow_fails_like_the_reference():
    sys.modules.pop("elasticdeform_amd.tf", None)
    with pytest.raises(ImportError):
        importlib.import_module("elasticdeform_amd.tf")


class _T(object):
    """stand-in for an eager tensor"""
    def __init__(self, a):
        self._a = np.asarray(a)
        self.dtype = self._a.dtype
        self.shape = self._a.shape
        self.device = "/job:localhost/replica:0/task:0/device:CPU:0"

    def numpy(self):
        return self._a


def _fake_tensorflow(tape):
    tf = types.ModuleType("tensorflow")

    def py_function(func, inp, Tout, name=None):
        outs = func(*[t if isinstance(t, _T) else _T(t) for t in inp])
        assert len(outs) == len(Tout)
        return [_T(np.asarray(o).astype(dt)) for o, dt in zip(outs, Tout)]

    def custom_gradient(fn):
        def wrapped(*a):
            y, grad = fn(*[t if isinstance(t, _T) else _T(t) for t in a])
            tape.append(grad)
            return y
        return wrapped
    tf.py_function = py_function
    tf.custom_gradient = custom_gradient
    return tf


@pytest.mark.gpu
def test_wrapper_control_flow_on_a_stand_in():
    import elasticdeform_amd as ed
    tape = []
    saved = sys.modules.get("tensorflow")
    sys.modules["tensorflow"] = _fake_tensorflow(tape)
    sys.modules.pop("elasticdeform_amd.tf", None)
    try:
        etf = importlib.import_module("elasticdeform_amd.tf")
        rng = np.random.default_rng(0)
        X = rng.random((20, 24, 28)).astype(np.float32)
        Y = rng.random((20, 24, 28))
        disp = rng.standard_normal((3, 3, 3, 3)) * 2.0
        kw = dict(order=3, mode="mirror")
        # single tensor in -> single tensor out
        y = etf.deform_grid(_T(X), _T(disp), **kw)
        np.testing.assert_array_equal(y.numpy(), ed.deform_grid(X, disp, **kw))
        dy = rng.random(X.shape).astype(np.float32)
        grads = tape[-1](_T(dy))
        assert len(grads) == 2 and grads[0].shape == disp.shape and np.isnan(grads[0].numpy()).all()
        np.testing.assert_allclose(grads[1].numpy(), ed.deform_grid_gradient(dy, disp, **kw), rtol=1e-5, atol=1e-6)
        # list in -> list out, per-input orders, a crop (X_shape comes from the forward tensors)
        kw2 = dict(order=[3, 1], mode="constant", crop=(slice(2, 18), slice(0, 24), slice(5, 20)))
        ys = etf.deform_grid([_T(X), _T(Y)], disp, **kw2)
        want = ed.deform_grid([X, Y], disp, **kw2)
        assert isinstance(ys, list) and len(ys) == 2 and ys[1].dtype == np.float64
        for a, b in zip(ys, want):
            np.testing.assert_array_equal(a.numpy(), b)
        dys = [rng.random(w.shape).astype(w.dtype) for w in want]
        grads = tape[-1](*[_T(d) for d in dys])
        wg = ed.deform_grid_gradient(dys, disp, X_shape=[X.shape, Y.shape], **kw2)
        assert len(grads) == 3 and np.isnan(grads[0].numpy()).all()
        np.testing.assert_allclose(grads[1].numpy(), wg[0], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(grads[2].numpy(), wg[1], rtol=1e-10, atol=1e-12)
    finally:
        sys.modules.pop("elasticdeform_amd.tf", None)
        if saved is not None:
            sys.modules["tensorflow"] = saved
        else:
            sys.modules.pop("tensorflow", None)


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_TF, reason="TensorFlow (TF-ROCm) is not installed in this image")
def test_wrapper_under_a_real_tensorflow():
    import tensorflow as tf
    import elasticdeform_amd as ed
    sys.modules.pop("elasticdeform_amd.tf", None)
    etf = importlib.import_module("elasticdeform_amd.tf")
    rng = np.random.default_rng(1)
    X = rng.random((20, 24, 28)).astype(np.float32)
    disp = rng.standard_normal((3, 3, 3, 3)) * 2.0
    xt = tf.constant(X)
    with tf.GradientTape() as tape:
        tape.watch(xt)
        y = etf.deform_grid(xt, tf.constant(disp), order=3, mode="mirror")
        loss = tf.reduce_sum(y)
    np.testing.assert_allclose(y.numpy(), ed.deform_grid(X, disp, order=3, mode="mirror"), rtol=1e-5, atol=1e-6)
    g = tape.gradient(loss, xt)
    np.testing.assert_allclose(g.numpy(), ed.deform_grid_gradient(np.ones_like(X), disp, order=3, mode="mirror"),
                               rtol=1e-5, atol=1e-5)

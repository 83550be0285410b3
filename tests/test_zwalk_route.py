"""The z-walk forward route (csrc/deform_k1z.hip) on geometries the default routing does not give it: with
``set_field_strength('strong')`` (EDHIP_FLAG_STRONG_FIELD) it serves every float32 call with three deformed axes,
spline orders 1-3 and a control grid its geometry kernel holds in LDS -- small volumes, partial tiles, batches, step
axes, crops, affine maps, strong fields whose tiles are taken in halves or left to the fix-up kernel.  Everything
against the oracle (tests/golden for the 256^3 cases of the default routing live in test_gpu_parity.py).  The route is
a function of the call's arguments and of that setting alone: repeated calls return the same bits.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ed_oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import elasticdeform_amd as ed  # noqa: E402

TOL = dict(rtol=1e-5, atol=3e-5)       # (prefiltered white noise of unit range reaches ~2.5: the 1e-5 budget scales with it)


@pytest.fixture(autouse=True)
def strong_field():
    prev = ed.set_field_strength("strong")
    yield
    ed.set_field_strength(prev)


@pytest.mark.parametrize("mode", ["nearest", "wrap", "reflect", "mirror", "constant"])
def test_general_tiles_every_mode(mode):
    """Tiles at the array's faces, partial tiles on every axis, displacements far larger than the array's margin
    (whole tiles fold, clamp, wrap or turn constant), a crop, an affine map -- orders 1-3."""
    rng = np.random.default_rng(len(mode) * 7 + 3)
    shape = (53, 70, 91)
    X = rng.random(shape).astype(np.float32)
    aff = np.eye(3, 4)
    aff[:, :3] += rng.standard_normal((3, 3)) * 0.05
    aff[:, 3] = rng.standard_normal(3) * 2
    for order in (1, 2, 3):
        for sigma, extra in ((9.0, {}), (4.0, dict(crop=(slice(5, 50), slice(0, 70), slice(11, 80)))), (6.0, dict(affine=aff))):
            disp = rng.standard_normal((3, 4, 3, 5)) * sigma
            kw = dict(order=order, mode=mode, cval=-0.75, **extra)
            np.testing.assert_allclose(ed.deform_grid(X, disp, **kw), orc.deform_grid(X, disp, **kw),
                                       err_msg="order %d sigma %g %s" % (order, sigma, list(extra)), **TOL)


@pytest.mark.parametrize("points", [(9, 9, 9), (13, 11, 13)])
def test_windows_outside_their_sampled_box_are_redone(points):
    """A control grid of 9-13 points on a 64^3 volume bends the field inside a tile beyond the margin of the sampled
    boxes: windows fall outside their tile's box and the fix-up kernel redoes those voxels from global memory."""
    rng = np.random.default_rng(sum(points) + 1)
    shape = (64, 64, 64)
    X = rng.random(shape).astype(np.float32)
    for order, mode in ((3, "mirror"), (1, "constant"), (2, "nearest"), (3, "reflect")):
        disp = rng.standard_normal((3,) + points) * 3.0
        kw = dict(order=order, mode=mode, cval=0.5)
        np.testing.assert_allclose(ed.deform_grid(X, disp, **kw), orc.deform_grid(X, disp, **kw),
                                   err_msg="order %d %s" % (order, mode), **TOL)
    Xc = rng.random((3,) + shape).astype(np.float32)          # channels (step axes) through the same boxes
    disp = rng.standard_normal((3,) + points) * 3.0
    kw = dict(order=3, mode="mirror", axis=(1, 2, 3))
    np.testing.assert_allclose(ed.deform_grid(Xc, disp, **kw), orc.deform_grid(Xc, disp, **kw), **TOL)


@pytest.mark.parametrize("sigma", [2.0, 8.0, 16.0])
def test_strong_fields_halves_and_unfit_tiles(sigma):
    """96 x 104 x 120, 5^3 grid: at sigma 8 tiles are taken as two z halves, at 16 many fit neither way (fix-up kernel);
    forward against the oracle, repeated calls bit-identical (whatever box size the spill feedback picks meanwhile),
    and the gradient call that follows -- on the boxes this route hands over -- against the oracle too."""
    rng = np.random.default_rng(int(sigma * 10))
    shape = (96, 104, 120)
    X = rng.random(shape).astype(np.float32)
    dY = rng.random(shape).astype(np.float32)
    disp = rng.standard_normal((3, 5, 5, 5)) * sigma
    Xd, dYd, dd = (torch.from_numpy(a).cuda() for a in (X, dY, disp))
    kw = dict(order=3, mode="mirror")
    first = ed.deform_grid(Xd, dd, **kw)
    np.testing.assert_allclose(first.cpu().numpy(), orc.deform_grid(X, disp, **kw), **TOL)
    for rep in range(4):
        torch.cuda.synchronize()
        assert torch.equal(ed.deform_grid(Xd, dd, **kw), first), rep
        g = ed.deform_grid_gradient(dYd, dd, **kw)
    gw = orc.deform_grid_gradient(dY, disp, **kw)
    truth = orc.deform_grid_gradient(dY.astype(np.float64), disp, **kw)
    err, ref_err = np.abs(g.cpu().numpy() - truth).max(), np.abs(gw - truth).max()
    scale = max(1.0, float(np.abs(truth).max()))
    assert err <= 4 * ref_err + 4 * np.finfo(np.float32).eps * scale, (err, ref_err, scale)


def test_batch_and_single_calls_agree():
    """One control grid per sample through the batch entry point (one geometry launch for all samples): the bits of
    the per-sample calls, and the oracle's values."""
    rng = np.random.default_rng(12)
    B, shape = 3, (40, 48, 56)
    X = rng.random((B,) + shape).astype(np.float32)
    disp = rng.standard_normal((B, 3, 4, 4, 4)) * 3.0
    Xd, dd = torch.from_numpy(X).cuda(), torch.from_numpy(disp).cuda()
    kw = dict(order=3, mode="reflect")
    got = ed.deform_grid_batch(Xd, dd, **kw)
    for b in range(B):
        one = ed.deform_grid(Xd[b], dd[b], **kw)
        assert torch.equal(got[b], one), b
        np.testing.assert_allclose(one.cpu().numpy(), orc.deform_grid(X[b], disp[b], **kw), **TOL)


def test_routes_agree_to_float_rounding():
    """The same call on the default route (x-strip kernel for this shape) and on the z-walk route: both within the
    oracle's tolerance, and within a few float32 ulp of each other."""
    rng = np.random.default_rng(5)
    X = rng.random((64, 72, 80)).astype(np.float32)
    disp = rng.standard_normal((3, 5, 5, 5)) * 2.5
    Xd, dd = torch.from_numpy(X).cuda(), torch.from_numpy(disp).cuda()
    kw = dict(order=3, mode="mirror")
    z = ed.deform_grid(Xd, dd, **kw)
    ed.set_field_strength("auto")
    s = ed.deform_grid(Xd, dd, **kw)
    ed.set_field_strength("strong")
    assert float((z - s).abs().max()) <= 2e-5
    np.testing.assert_allclose(z.cpu().numpy(), orc.deform_grid(X, disp, **kw), **TOL)
    np.testing.assert_allclose(s.cpu().numpy(), orc.deform_grid(X, disp, **kw), **TOL)
    with pytest.raises(ValueError):
        ed.set_field_strength("mild")


def test_deform_random_grid_hints_by_its_own_arguments():
    """deform_random_grid knows sigma and the control-point spacing: a field that is strong by construction takes the
    z-walk route without the caller's setting (here: 'auto'), a mild one the default routing -- same values as the
    oracle on the grid NumPy's global RNG draws, as in the reference (deform_grid.py:42-48)."""
    import importlib
    dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    ed.set_field_strength("auto")
    assert dgm._random_grid_hint(25, [3, 3], (200, 300)).strong             # the README example
    assert dgm._random_grid_hint(15, [5, 5, 5], (256, 256, 256)).strong
    assert not dgm._random_grid_hint(5, [5, 5, 5], (256, 256, 256)).strong  # the benchmark's field
    assert not dgm._random_grid_hint(25, [3, 1, 3], (1, 1, 1)).strong
    seen = []
    orig = dgm._route_flags

    def spy():
        f = orig()
        seen.append(f)
        return f
    dgm._route_flags = spy
    try:
        rng = np.random.default_rng(3)
        X = rng.random((64, 72, 80)).astype(np.float32)
        for sigma, strong in ((12.0, True), (1.0, False)):
            del seen[:]
            np.random.seed(1234)
            got = ed.deform_random_grid(X, sigma=sigma, points=4, order=3, mode="mirror")
            assert seen and all(bool(f) == strong for f in seen), (sigma, seen)
            np.random.seed(1234)
            disp = np.random.randn(3, 4, 4, 4) * sigma
            np.testing.assert_allclose(got, orc.deform_grid(X, disp, order=3, mode="mirror"), **TOL)
        # on the device (torch wrapper): the hint is set for the forward call, the values are those of the drawn grid
        import elasticdeform_amd.torch as etorch
        del seen[:]
        g = torch.Generator(device="cuda")
        g.manual_seed(7)
        Xd = torch.from_numpy(X).cuda()
        y = etorch.deform_random_grid(Xd, sigma=12.0, points=4, order=3, mode="mirror", generator=g)
        assert seen and all(seen)
        g.manual_seed(7)
        disp = etorch.random_displacement(3, 4, 12.0, device=Xd.device, generator=g).cpu().numpy()
        np.testing.assert_allclose(y.cpu().numpy(), orc.deform_grid(X, disp, order=3, mode="mirror"), **TOL)
    finally:
        dgm._route_flags = orig

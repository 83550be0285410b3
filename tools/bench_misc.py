#!/usr/bin/env python3
"""Timings of other configurations (not the headline bench): HIP events, data resident."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import elasticdeform_amd as ed
import cases as C

def timed(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]

dev = "cuda"
res = {}
rng = np.random.default_rng(0)
def T(a): return torch.from_numpy(a).to(dev)
# 2-D
X2 = T(rng.random((2048, 2048), dtype=np.float32)); d2 = T(rng.standard_normal((2, 3, 3)) * 25)
for order in (0, 1, 3):
    res["2d_2048_f32_o%d_fwd_ms" % order] = timed(lambda: ed.deform_grid(X2, d2, order=order))
res["2d_2048_f32_o3_grad_ms"] = timed(lambda: ed.deform_grid_gradient(X2, d2, order=3))
Xr, dr, kwr = C.cfg1_inputs(); Xr, dr = T(Xr), T(dr)
res["cfg1_200x300_fwd_ms"] = timed(lambda: ed.deform_grid(Xr, dr, **kwr))
# 3-D variants
X3 = T(rng.random((256, 256, 256), dtype=np.float32)); d3 = T(rng.standard_normal((3, 5, 5, 5)) * 5)
for order in (0, 1, 2, 3, 5):
    res["3d_256_f32_o%d_mirror_fwd_ms" % order] = timed(lambda: ed.deform_grid(X3, d3, order=order, mode="mirror"), 5)
for mode in ("constant", "nearest", "wrap", "reflect"):
    res["3d_256_f32_o3_%s_fwd_ms" % mode] = timed(lambda: ed.deform_grid(X3, d3, order=3, mode=mode), 5)
d10 = T(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * 10)
res["3d_256_f32_o3_mirror_sigma10_fwd_ms"] = timed(lambda: ed.deform_grid(X3, d10, order=3, mode="mirror"), 5)
res["3d_256_f32_o3_mirror_sigma10_grad_ms"] = timed(lambda: ed.deform_grid_gradient(X3, d10, order=3, mode="mirror"), 5)
L3 = T(rng.integers(0, 4, (256, 256, 256)).astype(np.int32))
res["3d_256_i32_o0_nearest_fwd_ms"] = timed(lambda: ed.deform_grid(L3, d3, order=0, mode="nearest"), 5)
X64 = X3.double()
res["3d_256_f64_o3_fast_fwd_ms"] = timed(lambda: ed.deform_grid(X64, d3, order=3, mode="mirror"), 3)
ed.set_arithmetic("exact")
res["3d_256_f64_o3_exact_fwd_ms"] = timed(lambda: ed.deform_grid(X64, d3, order=3, mode="mirror"), 3)
res["3d_256_f32_o3_exact_fwd_ms"] = timed(lambda: ed.deform_grid(X3, d3, order=3, mode="mirror"), 3)
ed.set_arithmetic("auto")
# cfg3 128^3 autograd
import elasticdeform_amd.torch as et
Xa = T(rng.random((128, 128, 128), dtype=np.float32)).requires_grad_(); da = T(rng.standard_normal((3, 5, 5, 5)) * 2.5); dY = torch.rand_like(Xa)
def fb():
    y = et.deform_grid(Xa, da, order=3, mode="mirror"); y.backward(dY); Xa.grad = None
res["cfg3_128_autograd_fwd_bwd_ms"] = timed(fb)
# cfg4
Xs, d4, kw4 = C.cfg4_inputs(); Xs = [T(x) for x in Xs]; d4 = T(d4)
res["cfg4_multi_crop64_affine_ms"] = timed(lambda: ed.deform_grid(Xs, d4, **kw4), 5)
kw4np = dict(kw4); kw4np["prefilter"] = False
res["cfg4_multi_crop64_affine_noprefilter_ms"] = timed(lambda: ed.deform_grid(Xs, d4, **kw4np), 5)
# channels (step axis): 4 x 128^3 with shared displacement
Xc = T(rng.random((4, 128, 128, 128), dtype=np.float32))
res["4ch_128_axis123_fwd_ms"] = timed(lambda: ed.deform_grid(Xc, da, order=3, mode="mirror", axis=(1, 2, 3)), 5)
# cfg4-like crop out of a 512^3 volume (crop-aware prefilter pays off with the volume size)
if os.environ.get("BIG"):
    Xb = T(rng.random((512, 512, 512), dtype=np.float32))
    cropb = (slice(224, 288),) * 3
    res["512_crop64_o3_fwd_ms"] = timed(lambda: ed.deform_grid(Xb, d3, order=3, mode="constant", crop=cropb), 5)
    dYb = T(rng.random((64, 64, 64), dtype=np.float32))
    res["512_crop64_o3_grad_ms"] = timed(lambda: ed.deform_grid_gradient(dYb, d3, order=3, mode="constant", crop=cropb, X_shape=(512, 512, 512)), 5)
# cfg5-like: a batch of 128^3 volumes with one grid each, forward + gradient (per-GPU share 64)
import elasticdeform_amd.torch as et2
Xb = torch.rand((32, 128, 128, 128), device=dev)
Db = et2.random_displacement(3, 5, 2.5, batch=32, device=dev)
dYb = torch.rand_like(Xb)
def loop_fg():
    for b in range(32):
        ed.deform_grid(Xb[b], Db[b], order=3, mode="mirror"); ed.deform_grid_gradient(dYb[b], Db[b], order=3, mode="mirror")
def batch_fg():
    ed.deform_grid_batch(Xb, Db, order=3, mode="mirror"); ed.deform_grid_gradient_batch(dYb, Db, order=3, mode="mirror")
res["cfg5_32x128_fwd_grad_loop_ms"] = timed(loop_fg, 3)
res["cfg5_32x128_fwd_grad_batch_ms"] = timed(batch_fg, 3)
print(json.dumps(res, indent=1))

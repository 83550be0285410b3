import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import elasticdeform_amd as ed
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.random((256, 256, 256), dtype=np.float32)).cuda()
d = torch.from_numpy(rng.standard_normal((3, 5, 5, 5)) * 5).cuda()
dY = torch.rand_like(X)
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[iters // 2]
for order in (1, 0):
    print("order", order, "fwd %.3f ms" % timed(lambda: ed.deform_grid(X, d, order=order, mode="mirror")),
          "grad %.3f ms" % timed(lambda: ed.deform_grid_gradient(dY, d, order=order, mode="mirror")))

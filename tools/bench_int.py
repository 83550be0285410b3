import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import elasticdeform_amd as ed
def timed(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[iters // 2]
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(a).cuda()
d3 = T(rng.standard_normal((3, 5, 5, 5)) * 5); d2 = T(rng.standard_normal((2, 5, 5)) * 20)
for dt in (np.int16, np.uint8):
    X3 = T((rng.random((256, 256, 256)) * 200).astype(dt)); X2 = T((rng.random((4096, 4096)) * 200).astype(dt))
    for order in (0, 1, 3):
        print("%s 3d_256 o%d fwd %.3f ms" % (dt.__name__, order, timed(lambda: ed.deform_grid(X3, d3, order=order, mode="nearest"))))
        print("%s 2d_4096 o%d fwd %.3f ms" % (dt.__name__, order, timed(lambda: ed.deform_grid(X2, d2, order=order, mode="nearest"))))

import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import elasticdeform_amd as ed
import elasticdeform_amd.torch as et
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.random((32, 32, 32), dtype=np.float32)).cuda()
d = torch.from_numpy(rng.standard_normal((3, 5, 5, 5)) * 2).cuda()
for _ in range(20): ed.deform_grid(X, d, order=3, mode="mirror")
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for _ in range(N): ed.deform_grid(X, d, order=3, mode="mirror")
torch.cuda.synchronize()
print("deform_grid 32^3: %.1f us per call" % ((time.perf_counter() - t0) / N * 1e6))
dY = torch.rand_like(X)
t0 = time.perf_counter()
for _ in range(N): ed.deform_grid_gradient(dY, d, order=3, mode="mirror")
torch.cuda.synchronize()
print("deform_grid_gradient 32^3: %.1f us per call" % ((time.perf_counter() - t0) / N * 1e6))
Xa = X.clone().requires_grad_()
t0 = time.perf_counter()
for _ in range(N):
    y = et.deform_grid(Xa, d, order=3, mode="mirror"); y.backward(dY); Xa.grad = None
torch.cuda.synchronize()
print("autograd fwd+bwd 32^3: %.1f us per call" % ((time.perf_counter() - t0) / N * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): ed.deform_grid(X, d, order=3, mode="mirror")
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:4500])

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import elasticdeform_amd as ed
rng = np.random.default_rng(2)
X = torch.from_numpy(rng.random((256, 256, 256), dtype=np.float32)).cuda()
disp = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * 5).cuda()
dY = torch.rand_like(X)
kw = dict(order=3, mode="mirror")
def step():
    ed.deform_grid(X, disp, **kw); ed.deform_grid_gradient(dY, disp, **kw)
for _ in range(5): step()
torch.cuda.synchronize()
N = 20
evs = [torch.cuda.Event(enable_timing=True) for _ in range(2 * N + 1)]
t0 = time.perf_counter()
evs[0].record()
for i in range(N):
    ed.deform_grid(X, disp, **kw); evs[2 * i + 1].record()
    ed.deform_grid_gradient(dY, disp, **kw); evs[2 * i + 2].record()
host_done = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
f = [evs[2 * i].elapsed_time(evs[2 * i + 1]) for i in range(N)]
g = [evs[2 * i + 1].elapsed_time(evs[2 * i + 2]) for i in range(N)]
print("wall/step %.4f ms  host enqueue/step %.4f ms  fwd median %.4f  grad median %.4f  (first fwd %.4f)" % (
    wall / N * 1e3, host_done / N * 1e3, sorted(f)[N // 2], sorted(g)[N // 2], f[0]))

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
import elasticdeform_amd as ed
rng = np.random.default_rng(0)
Xb = torch.from_numpy(rng.random((512, 512, 512), dtype=np.float32)).cuda()
d3 = torch.from_numpy(rng.standard_normal((3, 5, 5, 5)) * 5).cuda()
cropb = (slice(224, 288),) * 3
for _ in range(3): ed.deform_grid(Xb, d3, order=3, mode="constant", crop=cropb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): ed.deform_grid(Xb, d3, order=3, mode="constant", crop=cropb)
torch.cuda.synchronize()
print("wall per call ms", (time.perf_counter() - t0) * 100)
import importlib
dgm = importlib.import_module("elasticdeform_amd.deform_grid")
orig = dgm._crop_windows
def spy(*a, **k):
    w = orig(*a, **k); print("windows", w); return w
dgm._crop_windows = spy
ed.deform_grid(Xb, d3, order=3, mode="constant", crop=cropb)

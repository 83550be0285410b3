#!/usr/bin/env python3
"""dev tool (EXPERIMENTS build): where the waves of k1_fwd_kernel spend their time -- per-wave s_memtime sums of the
intervals of the tile loop on the bench workload.   python tools/k1_phases.py [sigma] [order]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
from elasticdeform_amd import _lib
dgm = importlib.import_module("elasticdeform_amd.deform_grid")
n = 256
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
order = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
X = torch.from_numpy(np.random.default_rng(2).random((n, n, n), dtype=np.float32)).to(dev)
disp = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * sigma).to(dev)
Xf = dgm._filter_axes(X, [0, 1, 2], order, False, dev) if order > 1 else X
df = dgm._filter_axes(disp, [1, 2, 3], 3, False, dev)
out = torch.empty_like(X)
stream = torch.cuda.current_stream(dev).cuda_stream
a_f = ([dgm._desc(Xf)], dgm._desc(df), None, [dgm._desc(out)], [(0, 1, 2)], [order], [3], [0.0], None, _lib.FLAG_AUTO, stream)
for _ in range(3):
    _lib.deform(False, *a_f)
torch.cuda.synchronize()
buf = torch.zeros((1 << 16, 8), dtype=torch.int64, device=dev)
os.environ["EDHIP_DEBUG_PTR"] = "%x" % buf.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); _lib.deform(False, *a_f); e1.record()
torch.cuda.synchronize()
del os.environ["EDHIP_DEBUG_PTR"]
b = buf.cpu().numpy().astype(np.float64)
b = b[b[:, 7] > 0]
names = ["prologue+boxes", "first coords", "stage issue", "coords(t+1)", "wait B2 (DMA)", "gather+store+B1", "general tiles"]
tot = b[:, :7].sum(axis=1)
print("sigma %g order %d: forward call %.1f us; %d waves reported; ticks per wave and strip: mean %.0f" % (sigma, order, e0.elapsed_time(e1) * 1e3, len(b), tot.mean()))
for k, nm in enumerate(names):
    print("  %-16s mean %8.0f ticks (%.1f %%)   per tile %7.0f   p10 %8.0f p90 %8.0f" %
          (nm, b[:, k].mean(), 100 * b[:, k].mean() / tot.mean(), (b[:, k] / b[:, 7]).mean(), np.percentile(b[:, k], 10), np.percentile(b[:, k], 90)))

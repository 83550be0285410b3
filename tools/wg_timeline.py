#!/usr/bin/env python3
"""dev tool (EXPERIMENTS build): per-workgroup timeline of the wave K1 kernel on the bench workload.
Every workgroup writes (s_memtime at entry, at exit, HW_ID, XCC_ID); this prints the duration
distribution, the idle gaps of the hardware wave slots and the number of resident waves over time.
  python tools/wg_timeline.py [grad]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
from elasticdeform_amd import _lib
dgm = importlib.import_module("elasticdeform_amd.deform_grid")
grad = len(sys.argv) > 1 and sys.argv[1] == "grad"
n, order, sigma = 256, 3, float(os.environ.get("SIGMA", "5"))
dev = torch.device("cuda", 0)
X = torch.from_numpy(np.random.default_rng(2).random((n, n, n), dtype=np.float32)).to(dev)
dY = torch.from_numpy(np.random.default_rng(7).random((n, n, n), dtype=np.float32)).to(dev)
disp = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * sigma).to(dev)
Xf = dgm._filter_axes(X, [0, 1, 2], order, False, dev)
df = dgm._filter_axes(disp, [1, 2, 3], 3, False, dev)
out = torch.empty_like(X); dxs = torch.zeros_like(X)
stream = torch.cuda.current_stream(dev).cuda_stream
a_f = ([dgm._desc(Xf)], dgm._desc(df), None, [dgm._desc(out)], [(0, 1, 2)], [order], [3], [0.0], None, _lib.FLAG_AUTO, stream)
a_g = ([dgm._desc(dxs)], dgm._desc(df), None, [dgm._desc(dY)], [(0, 1, 2)], [order], [3], [0.0], None, _lib.FLAG_AUTO, stream)
args = a_g if grad else a_f
for _ in range(3):
    _lib.deform(grad, *args)
torch.cuda.synchronize()
buf = torch.zeros((1 << 16, 4), dtype=torch.int64, device=dev)
os.environ["EDHIP_DEBUG_PTR"] = "%x" % buf.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); _lib.deform(grad, *args); e1.record()
torch.cuda.synchronize()
del os.environ["EDHIP_DEBUG_PTR"]
b = buf.cpu().numpy()
b = b[(b[:, 1] != 0) & (b[:, 0] != 0)]
t0, t1, hw, xcc = b[:, 0].astype(np.float64), b[:, 1].astype(np.float64), b[:, 2], b[:, 3] & 0xf
# the XCDs' counters are not synchronised: spans per XCD
spans = []
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    spans.append(t1[m].max() - t0[m].min())
span = float(np.mean(spans))
print("call %.1f us (events); %d workgroups; mean per-XCD tick span %.0f (min %.0f max %.0f) -> >= %.3f ticks/ns" %
      (e0.elapsed_time(e1) * 1e3, len(b), span, min(spans), max(spans), span / (e0.elapsed_time(e1) * 1e6)))
d = t1 - t0
print("workgroup duration ticks: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f  (sum/span = %.0f resident waves on average)" %
      (d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max(), d.sum() / span))
key = (xcc.astype(np.int64) << 32) | (hw & 0xffff)
slots = {}
for k, a_, c in zip(key, t0, t1):
    slots.setdefault(k, []).append((a_, c))
gaps = []
for k, lst in slots.items():
    lst.sort()
    gaps += [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
gaps = np.array(gaps)
print("%d distinct wave slots; workgroups per slot: mean %.2f" % (len(slots), len(b) / len(slots)))
print("gap between a slot's workgroups, ticks: mean %.0f median %.0f p90 %.0f max %.0f" % (gaps.mean(), np.median(gaps), np.percentile(gaps, 90), gaps.max()))
x0 = xcc == sorted(set(xcc.tolist()))[0]
edges = np.linspace(t0[x0].min(), t1[x0].max(), 21)
res = [(np.minimum(t1[x0], edges[i + 1]) - np.maximum(t0[x0], edges[i])).clip(0).sum() / (edges[i + 1] - edges[i]) for i in range(20)]
print("resident waves of XCD 0 (384 slots) per 5% of its span:", " ".join("%.0f" % r for r in res))
cus = len(set((int(x) << 32) | int(h & 0xff00) for x, h in zip(xcc, hw)))
print("CUs seen: %d; slots per CU: %.1f" % (cus, len(slots) / max(cus, 1)))

#!/usr/bin/env python3
"""dev tool (EXPERIMENTS build): phase clocks of k1z_geo_kernel (thread 0 of every workgroup).  python tools/geo_phases.py [sigma]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
n = 256
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
dev = torch.device("cuda", 0)
X = torch.from_numpy(np.random.default_rng(2).random((n, n, n), dtype=np.float32)).to(dev)
d = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * sigma).to(dev)
for _ in range(3):
    ed.deform_grid(X, d, order=3, mode="mirror", prefilter=False)
torch.cuda.synchronize()
buf = torch.zeros((1 << 12, 16), dtype=torch.int64, device=dev)
os.environ["EDHIP_DEBUG_PTR"] = "%x" % buf.data_ptr()
ed.deform_grid(X, d, order=3, mode="mirror", prefilter=False)
torch.cuda.synchronize()
del os.environ["EDHIP_DEBUG_PTR"]
b = buf.cpu().numpy().astype(np.float64)
b = b[b[:, 0] > 0]
t0 = b[:, 0].min()
names = ["grid -> LDS (+ prefilter)", "hint", "slack max", "axis entries, z table, sZ", "slack fold", "R", "boxes", "records", "summaries + lists"]
print("%d workgroups; first start -> last end %.0f ticks" % (len(b), b[:, 8].max() - t0))
print("start spread: %.0f ticks" % (b[:, 0].max() - t0))
for k in range(8):
    dt = b[:, k + 1] - b[:, k]
    print("  %-28s mean %8.0f  p10 %8.0f  p90 %8.0f  max %8.0f" % (names[k], dt.mean(), np.percentile(dt, 10), np.percentile(dt, 90), dt.max()))
print("  whole workgroup mean %.0f max %.0f" % ((b[:, 8] - b[:, 0]).mean(), (b[:, 8] - b[:, 0]).max()))

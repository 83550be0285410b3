#!/usr/bin/env python3
"""dev tool: BASELINE cfg4 forward / gradient calls in a loop (for rocprofv3 --kernel-trace --stats).
  WINDOW=auto|on|off python tools/cfg4_calls.py [iters]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases as C
import elasticdeform_amd as ed
dgm = importlib.import_module("elasticdeform_amd.deform_grid")
w = os.environ.get("WINDOW", "auto")
if w == "on":
    dgm.CROP_WINDOW_MIN_SAVING, dgm.CROP_WINDOW_MAX_FRACTION = 0, 1.0
elif w == "off":
    dgm.CROP_WINDOW_MIN_SAVING = 1 << 62
it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
(img, lab), disp, kw = C.cfg4_inputs(256)
dev = torch.device("cuda", 0)
Xs = [torch.from_numpy(img).to(dev), torch.from_numpy(lab).to(dev)]
dd = torch.from_numpy(disp).to(dev)
outs = ed.deform_grid(Xs, dd, **kw)
dYs = [torch.rand(outs[0].shape, device=dev), torch.ones_like(outs[1])]
xs = [tuple(img.shape), tuple(lab.shape)]
which = os.environ.get("WHICH", "both")
for _ in range(it):
    if which in ("both", "fwd"):
        ed.deform_grid(Xs, dd, **kw)
    if which in ("both", "grad"):
        ed.deform_grid_gradient(dYs, dd, X_shape=xs, **kw)
torch.cuda.synchronize()
print("done", w)

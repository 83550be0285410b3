#!/usr/bin/env python3
"""dev tool (experiments build): forward results of the level-1 configurations against each other, bit for bit."""
import os, subprocess, sys
import numpy as np
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import elasticdeform_amd as ed
    rng = np.random.default_rng(5)
    out = {}
    for name, shape, pts, sigma, kw in [("a", (96, 96, 96), (5, 5, 5), 9.0, dict(order=3, mode="mirror")),
                                       ("b", (48, 40, 72), (2, 3, 3), 9.0, dict(order=3, mode="wrap")),
                                       ("c", (64, 64, 64), (3, 3, 3), 4.0, dict(order=1, mode="constant")),
                                       ("d", (64, 64, 64), (3, 3, 3), 6.0, dict(order=2, mode="nearest")),
                                       ("e", (64, 56, 72), (3, 3, 3), 7.0, dict(order=4, mode="mirror")),
                                       ("f", (64, 56, 72), (3, 3, 3), 7.0, dict(order=5, mode="reflect")),
                                       ("g", (2, 64, 56, 72), (3, 3, 3), 7.0, dict(order=3, mode="constant", axis=(1, 2, 3),
                                                                                 affine=np.eye(3, 4) + 0.05 * np.arange(12).reshape(3, 4) / 12))]:
        X = torch.from_numpy(rng.random(shape, dtype=np.float32)).cuda()
        d = torch.from_numpy(rng.standard_normal((3,) + pts) * sigma).cuda()
        out[name] = ed.deform_grid(X, d, **kw).cpu().numpy()
    np.savez(sys.argv[1], **out)
    sys.exit(0)
cfgs = {"standard": {"EDHIP_NO_SPILL_HINT": "1"}, "large": {"EDHIP_NO_SPILL_HINT": "1", "EDHIP_HOT_FWD_KB": "52"},
        "general": {"EDHIP_NO_SPILL_HINT": "1", "EDHIP_NO_HOT": "1"}, "skip_l2": {"EDHIP_NO_SPILL_HINT": "1", "EDHIP_SKIP_L2": "1"},
        "wave": {"EDHIP_NO_SPILL_HINT": "1", "EDHIP_WAVE": "1"}, "feedback": {}}
res = {}
for k, env in cfgs.items():
    f = "/tmp/cmp_%s.npz" % k
    subprocess.check_call([sys.executable, __file__, f], env=dict(os.environ, **env))
    res[k] = np.load(f)
for k in cfgs:
    for c in res["standard"].files:
        a, b = res["standard"][c], res[k][c]
        print("%-9s case %s: %d of %d voxels differ, max |diff| %.3e" % (k, c, int((a != b).sum()), a.size, float(np.abs(a - b).max())))

#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; rm -rf $O; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_deformed or wide_control or grid_stamp or k1_" > $O/tests.txt 2>&1; tail -15 $O/tests.txt
timeout 300 python tools/time_4d.py 2>&1 | grep -v amdgpu | grep "grad" > $O/time_4d.txt; cat $O/time_4d.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/time_wide_f64.txt
import numpy as np, torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import elasticdeform_amd as ed
dev = torch.device("cuda", 0); rng = np.random.default_rng(1)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / iters
for dt in (np.float64, np.float32):
    for n, pts in ((256, 5), (256, 16), (256, 32), (128, 16)):
        X = torch.from_numpy(rng.random((n, n, n)).astype(dt)).to(dev)
        d = torch.from_numpy(rng.standard_normal((3, pts, pts, pts)) * (40.0 / pts)).to(dev)
        t = timeit(lambda: ed.deform_grid(X, d, order=3, mode="mirror", prefilter=False))
        tg = timeit(lambda: ed.deform_grid_gradient(X, d, order=3, mode="mirror", prefilter=False), 5)
        print("%d^3 %s, %2d^3 control points, order 3 (no prefilter): fwd %7.3f ms   grad %7.3f ms" % (n, np.dtype(dt).name, pts, t, tg), flush=True)
PY

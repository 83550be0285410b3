#!/usr/bin/env python3
"""dev tool: per-step durations (HIP events, no host sync between steps) of the benchmark step from a cold start:
does the step time settle with the GPU's clocks, with the allocator, or with the library's call-to-call state?
python tools/step_ramp.py [idle_ms_in_the_middle]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
idle_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
n = 256
dev = torch.device("cuda", 0)
X = torch.rand((n, n, n), device=dev)
dY = torch.rand((n, n, n), device=dev)
d = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * 5.0).to(dev)
torch.cuda.synchronize()
def run(N, tag):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2 * N + 1)]
    mem = []
    evs[0].record()
    for i in range(N):
        ed.deform_grid(X, d, order=3, mode="mirror")
        evs[2 * i + 1].record()
        ed.deform_grid_gradient(dY, d, order=3, mode="mirror")
        evs[2 * i + 2].record()
        mem.append(torch.cuda.memory_reserved() >> 20)
    torch.cuda.synchronize()
    f = [evs[2 * i].elapsed_time(evs[2 * i + 1]) for i in range(N)]
    g = [evs[2 * i + 1].elapsed_time(evs[2 * i + 2]) for i in range(N)]
    print(tag, "fwd :", " ".join("%.3f" % t for t in f[:14]), "...", " ".join("%.3f" % t for t in f[-5:]))
    print(tag, "grad:", " ".join("%.3f" % t for t in g[:14]), "...", " ".join("%.3f" % t for t in g[-5:]))
    print(tag, "reserved MiB:", mem[:8], mem[-1])
run(60, "A")
if idle_ms > 0:
    time.sleep(idle_ms * 1e-3)
    run(60, "B (after idle)")

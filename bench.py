#!/usr/bin/env python3
"""
bench.py -- BASELINE.json's metric on BASELINE.json's configuration, on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload = "cfg2"): one 256^3 float32 volume per GPU, 5x5x5 displacement grid
(sigma 5), order 3, mode 'mirror', the reference's default prefilter=True,
inputs already resident in HBM.  One step = one pass of the hot path over that volume:
``deform_grid`` (prefilter K3 -> forward K1) followed by ``deform_grid_gradient`` (scatter K2 ->
transposed prefilter K4) -- the "fwd+grad" of the metric.  Nothing is skipped or cached inside
the timed region.

value = Mvoxels/s = N * 256^3 * K / (max-over-ranks wall time of the K steps) / 1e6: every rank
owns an independent volume (different seed), no data-path collective (the path shards by volume,
SURVEY.md section 8e) => weak scaling.

Extra objects on the JSON line:
  roofline      dominant kernel = K1 forward (deform_tile3_fwd_kernel<float,3,true> plus its tiny
                tables / spill companions, i.e. one edhip_deform(gradient=0) call), HBM-bound.
                achieved = algorithmic bytes per launch (8 B/voxel: 4 read + 4 written,
                SURVEY.md 8d) / average launch duration, measured live with HIP events on the
                stream the kernel is launched on; peak 8 TB/s.
  cpu_baseline  the REAL reference C path (oracle/_ref, compiled from /root/reference) when that
                binary is present, else our C port (oracle/ed_oracle.c); one host core; bounded
                sample (128^3 forward + gradient with the same arguments); rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SIDE = 256
ALGO_BYTES_PER_VOXEL = 8      # K1 forward, float32: 4 B read + 4 B written per output voxel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--side", type=int, default=N_SIDE, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(side=128):
    """Reference C path on ONE host core: 128^3 forward + gradient, same arguments as the GPU
    workload (order 3, mirror, prefilter on, 5^3 grid scaled to the same relative strength)."""
    import numpy as np
    import scipy.ndimage
    from oracle import ref_loader
    n = side
    X = np.random.default_rng(3).random((n, n, n), dtype=np.float32)
    disp = np.random.default_rng(33).standard_normal((3, 5, 5, 5)) * (5.0 * n / 256)
    dY = np.random.default_rng(333).random((n, n, n), dtype=np.float32)
    ext = ref_loader.load_ref_ext()
    axis = [(0, 1, 2)]
    order = np.array([3], dtype="int64")
    mode = np.array([3], dtype="int64")      # mirror
    cval = np.array([0.0])
    if ext is not None:
        kind = "reference"
        # the reference's pipeline (deform_grid.py:155-174, :269-286) around its C entry points
        t0 = time.perf_counter()
        xf = np.zeros_like(X)
        src = X
        for d in range(3):
            scipy.ndimage.spline_filter1d(src, axis=d, order=3, output=xf)
            src = xf
        df = np.zeros_like(disp)
        src = disp
        for d in range(1, 4):
            scipy.ndimage.spline_filter1d(src, axis=d, order=3, output=df)
            src = df
        out = np.zeros_like(X)
        ext.deform_grid([xf], df, None, [out], axis, order, mode, cval, None)
        t1 = time.perf_counter()
        dX = np.zeros_like(X)
        ext.deform_grid_grad([dX], df, None, [dY], axis, order, mode, cval, None)
        g = np.zeros_like(dX)
        src = dX
        for d in range(3):
            ext.spline_filter1d_grad(src, g, d, 3)
            src = g
        t2 = time.perf_counter()
    else:
        kind = "port"
        from oracle import ed_oracle as orc
        t0 = time.perf_counter()
        orc.deform_grid(X, disp, order=3, mode="mirror")
        t1 = time.perf_counter()
        orc.deform_grid_gradient(dY, disp, order=3, mode="mirror")
        t2 = time.perf_counter()
    vox = float(n) ** 3
    return {
        "value": round(vox / (t2 - t0) / 1e6, 4), "unit": "Mvoxels/s", "cores": 1, "kind": kind,
        "sample": "%d^3 float32 forward+gradient, order 3, mirror, prefilter on, 5^3 grid "
                  "(same arguments as the GPU workload, 1/8 of its voxels)" % n,
        "fwd_s": round(t1 - t0, 3), "grad_s": round(t2 - t1, 3),
        "host_cores_available": os.cpu_count(),
    }


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # "nccl" is RCCL on ROCm; EDHIP_BENCH_BACKEND=gloo exists only to exercise this branch on a
        # single-GPU box
        dist.init_process_group(os.environ.get("EDHIP_BENCH_BACKEND", "nccl"), rank=rank,
                                world_size=world)

    import elasticdeform_amd as ed
    import importlib
    dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    from elasticdeform_amd import _lib

    n = args.side
    # every rank owns its own synthetic volume (shard = one volume; no cross-rank traffic)
    X = torch.from_numpy(np.random.default_rng(2 + 1000 * rank).random((n, n, n), dtype=np.float32)).to(dev)
    dY = torch.from_numpy(np.random.default_rng(7 + 1000 * rank).random((n, n, n), dtype=np.float32)).to(dev)
    disp_h = np.random.default_rng(22 + 1000 * rank).standard_normal((3, 5, 5, 5)) * (5.0 * n / 256)
    disp = torch.from_numpy(disp_h).to(dev)
    kw = dict(order=3, mode="mirror")

    def step():
        y = ed.deform_grid(X, disp, **kw)
        g = ed.deform_grid_gradient(dY, disp, **kw)
        return y, g

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-phase and dominant-kernel timing with HIP events on the launch stream ---------------
    def timed(fn, iters):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(iters)]
        for _ in range(2):      # untimed: first-use allocations of the torch caching allocator
            fn()
        torch.cuda.synchronize()
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return sum(ts) / len(ts), ts[len(ts) // 2]

    iters = max(10, args.steps)
    # (phases: medians -- a single allocator / first-use hiccup would dominate a 20-sample mean)
    _, fwd_ms = timed(lambda: ed.deform_grid(X, disp, **kw), iters)
    _, grad_ms = timed(lambda: ed.deform_grid_gradient(dY, disp, **kw), iters)

    # K1 alone: prefiltered inputs prepared once, then only edhip_deform(gradient=0) between events
    Xf = dgm._filter_axes(X, [0, 1, 2], 3, False, dev)
    df = dgm._filter_axes(disp, [1, 2, 3], 3, False, dev)
    out = torch.empty_like(X)
    dxs = torch.zeros_like(X)
    stream = torch.cuda.current_stream(dev).cuda_stream
    args_f = ([dgm._desc(Xf)], dgm._desc(df), None, [dgm._desc(out)], [(0, 1, 2)], [3], [3], [0.0],
              None, _lib.FLAG_AUTO, stream)
    args_g = ([dgm._desc(dxs)], dgm._desc(df), None, [dgm._desc(dY)], [(0, 1, 2)], [3], [3], [0.0],
              None, _lib.FLAG_AUTO, stream)
    for _ in range(3):
        _lib.deform(False, *args_f)
    k1_ms, k1_med = timed(lambda: _lib.deform(False, *args_f), max(50, iters))
    # the dominant kernel alone (without the per-call tables kernel and the two spill passes):
    # HIP events recorded by the library around that launch, on the launch stream
    L = _lib.load()
    L.edhip_profile_dominant(1)
    dom = []
    for _ in range(max(50, iters)):
        _lib.deform(False, *args_f)
        us = L.edhip_profile_last_us()
        if us > 0:
            dom.append(us)
    L.edhip_profile_dominant(0)
    if dom:
        dom_us = sum(dom) / len(dom)
        dom_med = sorted(dom)[len(dom) // 2]
    else:
        dom_us, dom_med = k1_ms * 1e3, k1_med * 1e3
    _, k2_ms = timed(lambda: _lib.deform(True, *args_g), iters)
    _, k3_ms = timed(lambda: dgm._filter_axes(X, [0, 1, 2], 3, False, dev), iters)
    _, k4_ms = timed(lambda: dgm._filter_axes(dxs, [0, 1, 2], 3, True, dev), iters)

    vox = float(n) ** 3
    algo_bytes = ALGO_BYTES_PER_VOXEL * vox
    achieved = algo_bytes / (dom_us * 1e-6) / 1e9         # GB/s
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                traffic = json.load(f).get("k1_forward_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        res = {
            "metric": "Mvoxels/s fwd+grad, 256^3 fp32 order=3",
            "value": round(world * vox * args.steps / elapsed / 1e6, 2),
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg2: 3D %dx%dx%d float32, 5x5x5 grid sigma 5, order 3, "
                                   "mode mirror, prefilter on, deform_grid + deform_grid_gradient "
                                   "per step, one volume per GPU" % (n, n, n),
                       "parallelism": "1 volume per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                         "kernel": "K1 forward deform: deform_tile3_fwd_kernel<float,3,true,0> (the strip launch of one edhip_deform gradient=0 call on the prefiltered input; whole_call adds the tables kernel and the two spill passes)",
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "avg_launch_us": round(dom_us, 2),
                         "median_launch_us": round(dom_med, 2),
                         "whole_call_avg_us": round(k1_ms * 1e3, 2),
                         "note": "HBM is the bounding roofline by definition (gather / interpolate, no "
                                 "MFMA); the kernel is VALU-issue-bound today: ~390 VALU instructions "
                                 "per voxel (fp64 coordinates + 64-tap separable gather), see DESIGN.md 5"},
            "phases_ms": {"deform_grid": round(fwd_ms, 4), "deform_grid_gradient": round(grad_ms, 4),
                          "K1_forward": round(k1_ms, 4), "K2_gradient": round(k2_ms, 4),
                          "K3_prefilter_3axes": round(k3_ms, 4),
                          "K4_prefilter_transpose_3axes": round(k4_ms, 4)},
            "fwd_only_mvox_s": round(vox / (fwd_ms * 1e-3) / 1e6, 1),
            "k1_only_mvox_s": round(vox / (k1_ms * 1e-3) / 1e6, 1),
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

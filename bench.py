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

`--workload cfg5` switches to BASELINE cfg5's per-GPU shard: 64 volumes of 128^3 float32 with one
5^3 grid each (512 volumes over 8 GPUs), forward + gradient through the single-launch batch kernels
(deform_grid_batch / deform_grid_gradient_batch); same metric, same JSON line.

`--workload cfg4` is BASELINE cfg4 (multi-input [3 x 256^3 float32 image order 3 mirror, 256^3 int32 labels
order 0 nearest], axis, crop 64^3, 3x4 affine = rotate 10 deg + zoom 1.1 about the crop centre): forward +
gradient per step, value = output voxels (4 x 64^3) per second; `crop_window` reports the forward / gradient
call with the crop-aware prefilter window as the library chooses, forced on and forced off.

`--collective` (with `--workload cfg5`, N > 1) is the other leg of SURVEY.md 8(e): the whole batch lives on
rank 0, the grids are broadcast, the volumes go out and the results come back point to point
(distributed.deform_batch_sharded(scatter_from=0, gather_to=0)); its own JSON line, `config.parallelism` says so.

`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`) and relays
rank 0's JSON line.

Extra objects on the JSON line:
  roofline      the DOMINANT kernel of the timed step = the kernel with the largest per-step GPU
                time.  The level-1 launches of K1 (forward gather) and K2 (gradient scatter) are
                both timed live with HIP events recorded by the library on the launch stream
                (edhip_profile_dominant); the larger one is reported (today K2), HBM-bound.
                achieved = algorithmic bytes per launch (8 B/voxel: 4 read + 4 written,
                SURVEY.md 8d) / average launch duration; peak 8 TB/s.  `traffic` = HBM bytes per
                launch from the PMC passes recorded in profiles/hbm_traffic.json (L2 <-> Infinity Cache /
                HBM requests by size: profiles/r06_traffic_calibration.txt), reported only when that file
                was measured on the kernel sources of this tree (stamped with its commit).
  north_star_kernel   the same numbers for K1, the kernel BASELINE.json's 50 % target is quoted on;
                `frac_read_only` prices it on the READ bytes alone (4 B/voxel), the literal
                "HBM-read roofline" of north_star.
  fresh_grid    the step as an augmentation loop runs it: a NEW displacement tensor every step (sigma 5 and
                sigma 10), so that whatever the library keeps from call to call (tile boxes, spill feedback,
                the repeat-call lane) is measured as that pattern uses it; cfg2, rank 0, N = 1 only.
  repeat_ms_per_step  the K-step region timed `--repeats` more times behind the headline region: median / min / max
                (the headline number stays the first region, as the contract says).  An idle MI355X runs the step
                ~7 % slower for its first ~25 repetitions (profiles/r06_step_ramp.txt: 0.71 -> 0.66 ms, again after
                0.5 s of idling; 60-200 ms of element-wise torch work in front does not shorten it), so the headline
                region -- steps 6..25 of the process -- sits on that ramp and the repeats show the settled rate.
  stress        SURVEY.md 8(d) "report both": the same step at sigma = 10 (the README example's
                aggressiveness: displacement gradient ~1.25, many tiles take the spill levels), a few
                timed steps after the headline region; cfg2, rank 0, N = 1 only.
  step_roofline       the whole step against HBM: 64 B/voxel algorithmic (K3 24 + K1 8 + K2 8 +
                K4 24) / ms_per_step.
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
    ap.add_argument("--side", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--workload", choices=("cfg2", "cfg4", "cfg5"), default="cfg2")
    ap.add_argument("--collective", action="store_true",
                    help="cfg5, N > 1: the batch lives on rank 0 (broadcast grids, point-to-point scatter / gather)")
    ap.add_argument("--dtype", choices=("f32", "bf16", "f16"), default="f32",
                    help="cfg2 with 16-bit float STORAGE (float32 arithmetic; the reduced-precision opt-in): its own line")
    ap.add_argument("--batch", type=int, default=64, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=6,
                    help="cfg2, N = 1: the K-step region timed this many more times behind the headline region "
                         "(median / min / max reported as `repeat_ms_per_step`; 0 = off)")
    ap.add_argument("--no-stress", action="store_true", help=argparse.SUPPRESS)      # profile collection: headline kernels only
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
        "note": ("the reference's own C path (oracle/_ref, built from /root/reference by `make -C oracle ref` / "
                 "__graft_entry__.build())" if kind == "reference" else
                 "oracle/_ref is not built in this tree (it is compiled from /root/reference, which a clean clone does "
                 "not have): the C restatement oracle/ed_oracle.c was timed instead"),
        "sample": "%d^3 float32 forward+gradient, order 3, mirror, prefilter on, 5^3 grid "
                  "(same arguments as the GPU workload, 1/8 of its voxels)" % n,
        "fwd_s": round(t1 - t0, 3), "grad_s": round(t2 - t1, 3),
        "host_cores_available": os.cpu_count(),
    }


def relaunch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks of one node (one process per GPU,
    127.0.0.1 rendezvous) with the same arguments and relay their output; returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# What K1 / K2 and their launch routing are built from: the PMC-derived numbers of profiles/hbm_traffic.json (HBM
# bytes per launch, VALUBusy, LDS busy) are reported only while the hash of these files matches the tree the bench
# runs in -- an edit to a kernel, to the strip length / routing in the launcher or to the API layer voids the stamp.
KERNEL_SOURCES = ("deform_k1z.hip", "ed_zwalk.h", "deform_k1.hip", "deform_hot.hip", "ed_hot.h", "ed_tile.h", "ed_device.h", "deform_tile.hip",
                  "edhip_api.hip", "ed_params.h")


def kernel_sources_sha16():
    import hashlib
    h = hashlib.sha256()
    for fn in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "elasticdeform_amd", "csrc", fn), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cfg4_main(args):
    """BASELINE cfg4 on one GPU: multi-input + axis + crop + affine, forward + gradient per step."""
    import importlib
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases as C
    import elasticdeform_amd as ed
    dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    n = args.side or N_SIDE
    (img, lab), disp, kw = C.cfg4_inputs(n)
    dev = torch.device("cuda", 0)
    Xs = [torch.from_numpy(img).to(dev), torch.from_numpy(lab).to(dev)]
    dd = torch.from_numpy(disp).to(dev)
    outs = ed.deform_grid(Xs, dd, **kw)
    dYs = [torch.rand(outs[0].shape, device=dev, dtype=torch.float32), torch.ones_like(outs[1])]
    xshape = [tuple(img.shape), tuple(lab.shape)]

    def fwd():
        return ed.deform_grid(Xs, dd, **kw)

    def bwd():
        return ed.deform_grid_gradient(dYs, dd, X_shape=xshape, **kw)

    def timed(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return ts[len(ts) // 2]

    for _ in range(args.warmup):
        fwd()
        bwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd()
        bwd()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    window = {"auto": {"deform_grid_ms": round(timed(fwd), 4), "deform_grid_gradient_ms": round(timed(bwd), 4)}}
    saved = (dgm.CROP_WINDOW_MIN_SAVING, dgm.CROP_WINDOW_MAX_FRACTION)
    try:
        dgm.CROP_WINDOW_MIN_SAVING, dgm.CROP_WINDOW_MAX_FRACTION = 0, 1.0
        window["forced_on"] = {"deform_grid_ms": round(timed(fwd), 4), "deform_grid_gradient_ms": round(timed(bwd), 4)}
        dgm.CROP_WINDOW_MIN_SAVING = 1 << 62
        window["forced_off"] = {"deform_grid_ms": round(timed(fwd), 4), "deform_grid_gradient_ms": round(timed(bwd), 4)}
    finally:
        dgm.CROP_WINDOW_MIN_SAVING, dgm.CROP_WINDOW_MAX_FRACTION = saved
    vox = float(sum(int(np.prod(o.shape)) for o in outs))
    ms = elapsed / args.steps * 1e3
    # Algorithmic bytes of the step, SURVEY.md 8(d): "with crop use bytes(output) + bytes(source bounding box actually
    # needed)".  The box is the filter window the library computed ON THE DEVICE for this very call
    # (edhip_source_window: the source box + the filter's decay margin); it is read back here, outside every timed
    # region, from the tensors _crop_windows hands to the filter passes.  An input without a window counts whole.
    seen = []
    real_cw = dgm._crop_windows

    def spy(plan, xs, *a, **k):
        wins = real_cw(plan, xs, *a, **k)
        seen.append([(tuple(int(d) for d in x.shape), w) for x, w in zip(xs, wins)])
        return wins
    dgm._crop_windows = spy
    try:
        fwd()
        bwd()
        torch.cuda.synchronize()
    finally:
        dgm._crop_windows = real_cw

    def window_voxels(shape, w):
        if w is None:
            return float(np.prod(shape))
        wv = w.cpu().numpy().reshape(-1, 2)
        return float(np.prod([max(int(b) - int(a), 0) for a, b in wv]))
    # per call: [image (3 channels, order 3), labels (order 0: no filter, no window)]
    win_f = window_voxels(*seen[0][0]) if seen else float(img.size)
    win_g = window_voxels(*seen[1][0]) if len(seen) > 1 else float(img.size)
    out_img, out_lab = float(np.prod(outs[0].shape)), float(np.prod(outs[1].shape))
    fwd_bytes = 24.0 * win_f + 4.0 * win_f + 4.0 * out_img + 8.0 * out_lab      # K3 passes, K1 source box, K1 stores, labels
    grad_bytes = 4.0 * out_img + 4.0 * win_g + 24.0 * win_g + 8.0 * out_lab     # dY, dX box, K4 passes, labels
    step_bytes = fwd_bytes + grad_bytes
    whole_bytes = 2.0 * (24.0 + 4.0) * float(img.size) + 8.0 * out_img + 16.0 * out_lab
    res = {"metric": "Mvoxels/s fwd+grad, cfg4 multi-input crop 64^3", "value": round(vox * args.steps / elapsed / 1e6, 2),
           "unit": "Mvoxels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "cfg4: [3x%d^3 float32 image order 3 mirror, %d^3 int32 labels order 0 nearest], "
                                  "axis [(1,2,3),(0,1,2)], crop %d^3, affine rotate 10 deg + zoom 1.1, 5x5x5 grid sigma 5, "
                                  "prefilter on, deform_grid + deform_grid_gradient per step" % (n, n, n // 4),
                      "parallelism": "1 GPU"},
           "roofline": {"bound": "hbm", "achieved": round(step_bytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(step_bytes / (ms * 1e-3) / 1e9 / 8000.0, 4), "traffic": None,
                        "kernel": "whole step against the bytes of the source window actually filtered and read "
                                  "(SURVEY 8d: outputs + source box): per voxel of the window 24 B (K3 / K4 passes) + 4 B "
                                  "(K1 read / K2 dX), per output voxel 4 B, labels 8 B; the window is the device-side one "
                                  "of this call (edhip_source_window), read back outside the timed region",
                        "algorithmic_bytes_per_step": int(step_bytes),
                        "window_voxels": {"forward": int(win_f), "gradient": int(win_g), "whole_input": int(img.size)},
                        "if_filtered_whole": {"algorithmic_bytes_per_step": int(whole_bytes),
                                              "note": "the reference's pipeline (deform_grid.py:155-164, 277-286) moves "
                                                      "these bytes; NOT what this step is priced on"}},
           "crop_window": window,
           "output_voxels_per_step": int(vox)}
    print(json.dumps(res), flush=True)


def storage16_main(args):
    """cfg2 with the volume, the result, dY and dX stored as 16-bit floats (opt-in extension; the reference rejects
    float16, deform.c:742-747).  Arithmetic stays float32: the first prefilter pass widens, K1's store narrows, K2's
    load of dY widens, the last transposed prefilter pass narrows.  Timed next to the cast route (widen / narrow with
    torch casts around the float32 pipeline), which is what the opt-in did until round 4."""
    import importlib
    import numpy as np
    import torch
    import elasticdeform_amd as ed
    dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    n = args.side or N_SIDE
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    X = torch.from_numpy(rng.random((n, n, n), dtype=np.float32)).to(dev).to(tdt)
    dY = torch.from_numpy(rng.random((n, n, n), dtype=np.float32)).to(dev).to(tdt)
    disp = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * (5.0 * n / 256)).to(dev)
    kw = dict(order=3, mode="mirror")
    ed.set_reduced_precision(True)

    def step():
        ed.deform_grid(X, disp, **kw)
        ed.deform_grid_gradient(dY, disp, **kw)

    def run():
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    ms = run()
    real = dgm._direct16
    try:
        dgm._direct16 = lambda *a, **k: False
        ms_cast = run()
    finally:
        dgm._direct16 = real
    vox = float(n) ** 3
    # bytes the step moves by design, per voxel: K3 (2 + 4) + 8 + 8, K1 4 + 2, K2 2 + 4, K4 8 + 8 + (4 + 2)
    direct_b, cast_b = 56.0, 64.0 + 4 * 6.0
    res = {"metric": "Mvoxels/s fwd+grad, 256^3 order=3, 16-bit storage", "value": round(vox / (ms * 1e-3) / 1e6, 2),
           "unit": "Mvoxels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 arithmetic, %s storage" % args.dtype, "data": "synthetic",
           "config": {"workload": "cfg2 with %s volumes: 3D %dx%dx%d, 5x5x5 grid sigma 5, order 3, mode mirror, prefilter on, "
                                  "deform_grid + deform_grid_gradient per step (set_reduced_precision(True))"
                                  % (args.dtype, n, n, n), "parallelism": "1 GPU"},
           "roofline": {"bound": "hbm", "achieved": round(direct_b * vox / (ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(direct_b * vox / (ms * 1e-3) / 1e9 / 8000.0, 4), "traffic": None,
                        "kernel": "whole step against the bytes it moves by design: 56 B/voxel (float32 intermediates "
                                  "between the filter passes; the cast route moves 88)",
                        "algorithmic_bytes_per_step": direct_b * vox},
           "cast_route": {"ms_per_step": round(ms_cast, 4), "bytes_per_voxel": cast_b,
                          "note": "the same step with torch casts around the float32 pipeline"},
           "cpu_baseline": None}
    print(json.dumps(res))


def collective_main(args, rank, world, dev, distributed):
    """SURVEY.md 8(e), second leg: the whole cfg5 batch lives on rank 0; per step the grids are broadcast, the
    volumes scattered and the results gathered point to point (forward, then the same for the gradient)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from elasticdeform_amd import distributed as edd
    n = args.side or 128
    B = args.batch * world                 # the batch of all ranks, resident on rank 0
    sigma = 5.0 * n / 256
    X = dY = disp = None
    if rank == 0:
        gen = torch.Generator(device=dev)
        gen.manual_seed(2)
        X = torch.rand((B, n, n, n), device=dev, dtype=torch.float32, generator=gen)
        dY = torch.rand((B, n, n, n), device=dev, dtype=torch.float32, generator=gen)
        disp = torch.from_numpy(np.random.default_rng(22).standard_normal((B, 3, 5, 5, 5)) * sigma).to(dev)
    kw = dict(order=3, mode="mirror")

    def step():
        edd.deform_batch_sharded(X, disp, scatter_from=0, gather_to=0, device=dev, **kw)
        edd.deform_batch_sharded(dY, disp, gradient=True, scatter_from=0, gather_to=0, device=dev, **kw)

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
    if rank == 0:
        vox = float(B) * float(n) ** 3
        moved = 4.0 * vox * 4.0 * (world - 1) / max(world, 1)       # X out, Y back, dY out, dX back: float32, off-rank shards only
        print(json.dumps({
            "metric": "Mvoxels/s fwd+grad, batch of 128^3 fp32 order=3, batch resident on rank 0",
            "value": round(vox * args.steps / elapsed / 1e6, 2), "unit": "Mvoxels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg5, collective leg: %d volumes of %d^3 float32 on rank 0, one 5x5x5 grid each, order 3, "
                                   "mirror, prefilter on, forward + gradient per step" % (B, n),
                       "parallelism": "grids broadcast; volumes scattered from and results gathered to rank 0 point to point "
                                      "(backend %s); %d volumes computed per rank" % (dist.get_backend() if distributed else "none",
                                                                                      args.batch)},
            "bytes_moved_per_step": int(moved),
            "link_GBps": round(moved / (elapsed / args.steps) / 1e9, 2)}), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch(args))
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

    if args.dtype != "f32":
        if args.workload != "cfg2" or args.gpus != 1:
            raise SystemExit("--dtype bf16 / f16: the cfg2 workload on one GPU")
        return storage16_main(args)
    if args.workload == "cfg4":
        if rank == 0:
            cfg4_main(args)
        if distributed:
            dist.destroy_process_group()
        return
    if args.collective:
        collective_main(args, rank, world, dev, distributed)
        if distributed:
            dist.destroy_process_group()
        return
    cfg5 = args.workload == "cfg5"
    n = args.side or (128 if cfg5 else N_SIDE)
    B = args.batch if cfg5 else 1
    # every rank owns its own synthetic data (shard = whole volumes; no cross-rank traffic)
    shape = (B, n, n, n) if cfg5 else (n, n, n)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2 + 1000 * rank)
    X = torch.rand(shape, device=dev, dtype=torch.float32, generator=gen)
    dY = torch.rand(shape, device=dev, dtype=torch.float32, generator=gen)
    sigma = 5.0 * n / 256          # same relative strength at every size (cfg5: 2.5 at 128^3)
    dshape = (B, 3, 5, 5, 5) if cfg5 else (3, 5, 5, 5)
    disp = torch.from_numpy(np.random.default_rng(22 + 1000 * rank).standard_normal(dshape) * sigma).to(dev)
    kw = dict(order=3, mode="mirror")
    fwd = ed.deform_grid_batch if cfg5 else ed.deform_grid
    bwd = ed.deform_grid_gradient_batch if cfg5 else ed.deform_grid_gradient

    def step():
        y = fwd(X, disp, **kw)
        g = bwd(dY, disp, **kw)
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

    # ---- per-phase and per-kernel timing with HIP events on the launch stream ---------------------
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
    _, fwd_ms = timed(lambda: fwd(X, disp, **kw), iters)
    _, grad_ms = timed(lambda: bwd(dY, disp, **kw), iters)

    # K1 / K2 alone: prefiltered inputs prepared once, then only the deform call between events
    vol_axes = [1, 2, 3] if cfg5 else [0, 1, 2]
    Xf = dgm._filter_axes(X, vol_axes, 3, False, dev)
    df = dgm._filter_axes(disp, [a + 1 for a in vol_axes], 3, False, dev)
    out = torch.empty_like(X)
    dxs = torch.zeros_like(X)
    stream = torch.cuda.current_stream(dev).cuda_stream
    if cfg5:
        (xd, xs), (dd0, ds), (od, os_) = dgm._desc_sample0(Xf), dgm._desc_sample0(df), dgm._desc_sample0(out)
        (gxd, gxs), (gyd, gys) = dgm._desc_sample0(dxs), dgm._desc_sample0(dY)

        def k1():
            _lib.deform_batch_strided(False, B, xd, xs, dd0, ds, None, od, os_, (0, 1, 2), 3, 3, 0.0, None,
                                      _lib.FLAG_AUTO, stream)

        def k2():
            _lib.deform_batch_strided(True, B, gxd, gxs, dd0, ds, None, gyd, gys, (0, 1, 2), 3, 3, 0.0, None,
                                      _lib.FLAG_AUTO, stream)
    else:
        # as in the step: the forward call leaves its tiles' bounding boxes for the gradient call
        # (EDHIP_FLAG_KEEP_BOXES / USE_BOXES; deform_grid / deform_grid_gradient set them for a
        # displacement tensor that is handed to both unchanged)
        args_f = ([dgm._desc(Xf)], dgm._desc(df), None, [dgm._desc(out)], [(0, 1, 2)], [3], [3], [0.0],
                  None, _lib.FLAG_AUTO | _lib.FLAG_KEEP_BOXES, stream)
        args_g = ([dgm._desc(dxs)], dgm._desc(df), None, [dgm._desc(dY)], [(0, 1, 2)], [3], [3], [0.0],
                  None, _lib.FLAG_AUTO | _lib.FLAG_USE_BOXES, stream)

        def k1():
            _lib.deform(False, *args_f)

        def k2():
            _lib.deform(True, *args_g)
    for _ in range(3):
        k1()
        k2()
    k1_ms, _ = timed(k1, max(30, iters))
    k2_ms, _ = timed(k2, max(30, iters))
    # the level-1 launch of each (without the per-call tables kernel and the two spill passes):
    # HIP events recorded by the library around that launch, on the launch stream
    L = _lib.load()

    def level1(fn, n_it):
        L.edhip_profile_dominant(1)
        ts = []
        for _ in range(n_it):
            fn()
            us = L.edhip_profile_last_us()
            if us > 0:
                ts.append(us)
        L.edhip_profile_dominant(0)
        if not ts:
            return None, None
        return sum(ts) / len(ts), sorted(ts)[len(ts) // 2]

    k1_us, k1_med = level1(k1, max(30, iters))
    k2_us, k2_med = level1(k2, max(30, iters))
    if k1_us is None:
        k1_us, k1_med = k1_ms * 1e3, k1_ms * 1e3
    if k2_us is None:
        k2_us, k2_med = k2_ms * 1e3, k2_ms * 1e3
    _, k3_ms = timed(lambda: dgm._filter_axes(X, vol_axes, 3, False, dev), iters)
    _, k4_ms = timed(lambda: dgm._filter_axes(dxs, vol_axes, 3, True, dev), iters)

    vox = float(B) * float(n) ** 3
    algo_bytes = ALGO_BYTES_PER_VOXEL * vox

    def commit_id():
        try:
            import subprocess
            return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"],
                                           stderr=subprocess.DEVNULL).decode().strip()
        except Exception:
            return None

    # HBM bytes per launch from the PMC passes (profiles/hbm_traffic.json): reported only while the kernels'
    # sources are the ones the passes were made on (hash of the files, written by tools/hbm_traffic.py)
    traffic_db = {}
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                traffic_db = json.load(f)
            if traffic_db.get("kernel_sources_sha16") != kernel_sources_sha16():
                traffic_db = {}
        except Exception:
            traffic_db = {}

    def kernel_obj(tag, name, us, med, call_ms):
        achieved = algo_bytes / (us * 1e-6) / 1e9          # GB/s
        t = traffic_db.get("kernels", {}).get(tag) if not cfg5 and n == N_SIDE else None
        return {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4),
                "traffic": t.get("bytes_per_launch") if t else None,
                # (bytes between the L2s and the Infinity Cache / HBM, read + written, from the request-size counters:
                # profiles/r06_traffic_calibration.txt; Infinity-Cache hits are included)
                "traffic_read": t.get("read_bytes") if t else None,
                "traffic_written": t.get("write_bytes") if t else None,
                "l2_hit_rate": t.get("l2_hit_rate") if t else None,
                "traffic_measured_at": t.get("commit") if t else None,
                # (PMC passes of the same stamp: SQ_ACTIVE_INST_VALU x 4 / SIMDs / busy cycles, SQ_LDS_IDX_ACTIVE / CUs / busy cycles)
                "valu_busy": t.get("valu_busy") if t else None,
                "lds_busy": t.get("lds_busy") if t else None,
                "kernel": name,
                "algorithmic_bytes_per_launch": int(algo_bytes),
                "avg_launch_us": round(us, 2), "median_launch_us": round(med, 2),
                "whole_call_avg_us": round(call_ms * 1e3, 2)}

    def commit_built():
        try:
            from elasticdeform_amd import _build_info
            return _build_info.COMMIT
        except Exception:
            return None

    k1_obj = kernel_obj("K1", "K1 forward deform: k1z_tile_kernel<3,false> (deform_k1z.hip, the z-walk kernel of "
                        "round 6; the tile launch of one edhip_deform gradient=0 call on the prefiltered input; "
                        "whole_call adds the geometry kernel in front of it and the fix-up kernel behind it)",
                        k1_us, k1_med, k1_ms)
    k2_obj = kernel_obj("K2", "K2 gradient scatter-add: hot_grad_kernel<3,false> (deform_hot.hip; the strip "
                        "launch of one edhip_deform gradient=1 call; whole_call adds the tables kernel and "
                        "the two spill passes)", k2_us, k2_med, k2_ms)
    k1_obj["frac_read_only"] = round(4.0 * vox / (k1_us * 1e-6) / 1e9 / 8000.0, 4)
    dominant = k2_obj if k2_us >= k1_us else k1_obj
    dominant = dict(dominant)
    dominant["note"] = ("dominant = largest per-step GPU time among K1, K2 and the six prefilter passes "
                        "(K3/K4: %.1f / %.1f us per axis pass); HBM is the bounding roofline by definition "
                        "(gather / scatter, no MFMA); both tile kernels are LDS- and issue-bound today, "
                        "see DESIGN.md 5" % (k3_ms * 1e3 / 3, k4_ms * 1e3 / 3))
    ms_per_step = elapsed / args.steps * 1e3
    # box-to-box and run-to-run noise is a few per cent: the same K-step region REPEATS more times (outside the headline
    # number, which stays "exactly K steps after W warm-up steps"); median, minimum and maximum are reported beside it
    repeats = []
    if rank == 0 and world == 1 and args.repeats > 0:
        for _ in range(args.repeats):
            fence()
            tr = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            repeats.append((time.perf_counter() - tr) / args.steps * 1e3)
        repeats.sort()
    step_bytes = 64.0 * vox
    step_obj = {"bound": "hbm", "algorithmic_bytes_per_step": int(step_bytes),
                "achieved": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / 8000.0, 4),
                "note": "64 B/voxel = prefilter 24 + forward 8 + gradient 8 + transposed prefilter 24"}

    stress = None
    if rank == 0 and world == 1 and not cfg5 and not args.no_stress:
        disp10 = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * (10.0 * n / 256)).to(dev)

        def step10():
            fwd(X, disp10, **kw)
            bwd(dY, disp10, **kw)
        for _ in range(5):            # (the level-1 spill feedback settles within two or three calls)
            step10()
        torch.cuda.synchronize()
        ns = max(5, args.steps // 2)
        t0s = time.perf_counter()
        for _ in range(ns):
            step10()
        torch.cuda.synchronize()
        ms10 = (time.perf_counter() - t0s) / ns * 1e3
        _, f10 = timed(lambda: fwd(X, disp10, **kw), 10)
        _, g10 = timed(lambda: bwd(dY, disp10, **kw), 10)
        stress = {"workload": "cfg2 at sigma 10 (stress): same volume, grid N(0,1)*10, order 3, mirror, prefilter on, "
                              "deform_grid + deform_grid_gradient per step",
                  "steps": ns, "ms_per_step": round(ms10, 4), "value": round(vox / (ms10 * 1e-3) / 1e6, 2),
                  "unit": "Mvoxels/s", "vs_headline_ms": round(ms10 / (elapsed / args.steps * 1e3), 3),
                  "deform_grid_ms": round(f10, 4), "deform_grid_gradient_ms": round(g10, 4)}

    fresh = None
    if rank == 0 and world == 1 and not cfg5 and not args.no_stress:
        # the step as an augmentation loop runs it: a new displacement tensor every step
        fresh = {}
        for sg in (5.0, 10.0):
            ns = max(10, args.steps)
            rng = np.random.default_rng(2200 + int(sg))
            grids = [torch.from_numpy(rng.standard_normal((3, 5, 5, 5)) * (sg * n / 256)).to(dev) for _ in range(ns + 3)]
            for g in grids[:3]:
                fwd(X, g, **kw)
                bwd(dY, g, **kw)
            torch.cuda.synchronize()
            t0f = time.perf_counter()
            for g in grids[3:]:
                fwd(X, g, **kw)
                bwd(dY, g, **kw)
            torch.cuda.synchronize()
            msf = (time.perf_counter() - t0f) / ns * 1e3
            fresh["sigma_%g" % sg] = {"steps": ns, "ms_per_step": round(msf, 4), "value": round(vox / (msf * 1e-3) / 1e6, 2),
                                      "unit": "Mvoxels/s"}
        fresh["workload"] = ("cfg2 with a NEW 5x5x5 displacement tensor every step (deform_grid + deform_grid_gradient on "
                             "the same new tensor), sigma 5 and sigma 10")
        fresh["vs_headline_ms"] = round(fresh["sigma_5"]["ms_per_step"] / ms_per_step, 3)

    if rank == 0:
        if cfg5:
            workload = ("cfg5 shard: %d volumes of %d^3 float32 per GPU, one 5x5x5 grid (sigma %.3g) each, "
                        "order 3, mode mirror, prefilter on, deform_grid_batch + "
                        "deform_grid_gradient_batch per step" % (B, n, sigma))
        else:
            workload = ("cfg2: 3D %dx%dx%d float32, 5x5x5 grid sigma 5, order 3, mode mirror, prefilter "
                        "on, deform_grid + deform_grid_gradient per step, one volume per GPU" % (n, n, n))
        res = {
            "metric": "Mvoxels/s fwd+grad, 256^3 fp32 order=3" if not cfg5 else
                      "Mvoxels/s fwd+grad, batch of 128^3 fp32 order=3",
            "value": round(world * vox * args.steps / elapsed / 1e6, 2),
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload,
                       "parallelism": "%s per GPU, no collective" % ("%d volumes" % B if cfg5 else "1 volume")},
            "roofline": dominant,
            "north_star_kernel": k1_obj,
            "step_roofline": step_obj,
            "commit": commit_id() or commit_built(),
            "phases_ms": {"deform_grid": round(fwd_ms, 4), "deform_grid_gradient": round(grad_ms, 4),
                          "K1_forward": round(k1_ms, 4), "K2_gradient": round(k2_ms, 4),
                          "K3_prefilter_3axes": round(k3_ms, 4),
                          "K4_prefilter_transpose_3axes": round(k4_ms, 4)},
            "fwd_only_mvox_s": round(vox / (fwd_ms * 1e-3) / 1e6, 1),
            "k1_only_mvox_s": round(vox / (k1_ms * 1e-3) / 1e6, 1),
        }
        if repeats:
            res["repeat_ms_per_step"] = {"repeats": len(repeats), "steps_each": args.steps,
                                         "median": round(repeats[len(repeats) // 2], 4),
                                         "min": round(repeats[0], 4), "max": round(repeats[-1], 4),
                                         "note": "the same K-step region timed again after the headline region "
                                                 "(value / ms_per_step above are the first region alone)"}
        if stress is not None:
            res["stress"] = stress
        if fresh is not None:
            res["fresh_grid"] = fresh
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

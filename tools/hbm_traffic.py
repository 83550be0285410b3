#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, one counter per pass) -> profiles/hbm_traffic.json.
usage: hbm_traffic.py <pmc dir> <commit>.  Raw counters are in KB per dispatch.  Calibration
(profiles/r01_traffic_calibration.txt): FETCH_SIZE counts 64-byte requests at face value and reports
half of the bytes of wide streams; K1 reads its volume with 16-byte lanes in 64-byte runs (face value:
rd16_runs), K2 reads dY with 4-byte lanes in 64-byte rows, which the counter reports at HALF
(profiles/r02_traffic_calibration.txt, rd4_k2rows): the missing half of the 64 MiB of dY is added
back for K2 (`fetch_correction_kb`).  WRITE_SIZE is exact for full streams and 32-byte runs."""
import collections
import csv
import glob
import json
import sys

import hashlib
import os

root, commit = sys.argv[1], sys.argv[2]


def kernel_sources_sha16():
    """Hash of the sources the two measured kernels are built from: bench.py reports `traffic` only while it
    matches the tree it runs in (a stamp that cannot go stale unnoticed)."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("deform_hot.hip", "ed_tile.h", "ed_device.h"):
        with open(os.path.join(here, "elasticdeform_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/time_k12.py (the bench workload: "
               "256^3 float32, order 3, mirror, sigma 5), mean per dispatch, KB -> bytes",
       "algorithmic_bytes_per_launch": 134217728, "kernel_sources_sha16": kernel_sources_sha16(), "kernels": {}}
for tag, key in (("K1", "hot_fwd_kernel<3, false, 0,"), ("K2", "hot_grad_kernel<3, false")):
    for name, c in vals.items():
        if key in name and c.get("FETCH_SIZE") and c.get("WRITE_SIZE"):
            fkb = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
            wkb = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            corr = 32768 if tag == "K2" else 0       # KB: the uncounted half of dY (256^3 float32)
            out["kernels"][tag] = {"kernel": name, "fetch_kb": round(fkb), "fetch_correction_kb": corr,
                                   "write_kb": round(wkb),
                                   "bytes_per_launch": int((fkb + corr + wkb) * 1024), "commit": commit}
print(json.dumps(out, indent=1))

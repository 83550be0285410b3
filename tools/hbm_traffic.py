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


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha16      # noqa: E402  (one list of files for the stamp and for its check)


vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
                                 "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
            vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/time_k12.py (the bench workload: "
               "256^3 float32, order 3, mirror, sigma 5), mean per dispatch, KB -> bytes",
       "algorithmic_bytes_per_launch": 134217728, "kernel_sources_sha16": kernel_sources_sha16(), "kernels": {}}
def mean(c, k):
    return sum(c[k]) / len(c[k]) if c.get(k) else None


for tag, key in (("K1", "k1z_tile_kernel<3, false"), ("K2", "hot_grad_kernel<3, false")):
    for name, c in vals.items():
        if key in name and c.get("FETCH_SIZE") and c.get("WRITE_SIZE"):
            fkb = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
            wkb = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            corr = 32768 if tag == "K2" else 0       # KB: the uncounted half of dY (256^3 float32)
            out["kernels"][tag] = {"kernel": name, "fetch_kb": round(fkb), "fetch_correction_kb": corr,
                                   "write_kb": round(wkb),
                                   "bytes_per_launch": int((fkb + corr + wkb) * 1024), "commit": commit}
            # busy fractions: GRBM_GUI_ACTIVE is summed over the 8 XCDs -> cycles of the launch = / 8;
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs, SQ_LDS_IDX_ACTIVE cycles over the 256 CUs
            gui = mean(c, "GRBM_GUI_ACTIVE")
            if gui:
                cyc = gui / 8.0
                k = out["kernels"][tag]
                k["launch_cycles"] = round(cyc)
                if mean(c, "SQ_ACTIVE_INST_VALU"):
                    k["valu_busy"] = round(mean(c, "SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / cyc, 3)
                if mean(c, "SQ_LDS_IDX_ACTIVE"):
                    k["lds_busy"] = round(mean(c, "SQ_LDS_IDX_ACTIVE") / 256.0 / cyc, 3)
                    if mean(c, "SQ_LDS_BANK_CONFLICT"):
                        k["lds_conflict_share"] = round(mean(c, "SQ_LDS_BANK_CONFLICT") / mean(c, "SQ_LDS_IDX_ACTIVE"), 3)
                for nm, cn in (("valu_per_64_voxels", "SQ_INSTS_VALU"), ("salu_per_64_voxels", "SQ_INSTS_SALU"),
                               ("lds_per_64_voxels", "SQ_INSTS_LDS")):
                    if mean(c, cn):
                        k[nm] = round(mean(c, cn) / 262144.0, 1)
print(json.dumps(out, indent=1))

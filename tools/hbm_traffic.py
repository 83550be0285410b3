#!/usr/bin/env python3
"""Fabric-side traffic of the L2s per launch (rocprofv3 --pmc, one counter group per pass: tools/pmc_k1.sh) -> profiles/hbm_traffic.json.
usage: hbm_traffic.py <pmc dir> <commit>.

    read bytes  = 128 x TCC_EA0_RDREQ_128B + 64 x TCC_EA0_RDREQ_64B + 32 x TCC_EA0_RDREQ_32B
    write bytes = 64 x TCC_EA0_WRREQ_64B + 32 x (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B)

calibrated on known byte counts in the kernels' own access patterns (profiles/r06_traffic_calibration.txt, tools/ubench_traffic.hip):
every read pattern of K1z / K2 issues 128-byte requests, which FETCH_SIZE (= requests x 64 B) reports at HALF -- the guide's gfx950
correction -- and the rounds 1-5 stamps took "64-byte runs" at face value, i.e. they under-reported K1's reads by 2x.  These are
bytes between the L2s and the Infinity Cache / HBM: re-reads that hit the 256 MiB Infinity Cache are included.  The L2 hit rate
(TCC_HIT / (TCC_HIT + TCC_MISS)) is reported next to them."""
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
        if r["Counter_Name"] in ("FETCH_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum",
                                 "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_HIT_sum", "TCC_MISS_sum",
                                 "SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
                                 "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
            vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc TCC_EA0_RDREQ / WRREQ by request size (separate passes) on tools/time_k12.py (the bench workload: "
               "256^3 float32, order 3, mirror, sigma 5), mean per dispatch; bytes between the L2s and the Infinity Cache / HBM "
               "(profiles/r06_traffic_calibration.txt)",
       "algorithmic_bytes_per_launch": 134217728, "kernel_sources_sha16": kernel_sources_sha16(), "kernels": {}}
def mean(c, k):
    return sum(c[k]) / len(c[k]) if c.get(k) else None


for tag, key in (("K1", "k1z_tile_kernel<3, false"), ("K2", "hot_grad_kernel<3, false")):
    for name, c in vals.items():
        if key in name and c.get("TCC_EA0_RDREQ_sum") and c.get("TCC_EA0_WRREQ_sum"):
            rd = 128.0 * mean(c, "TCC_EA0_RDREQ_128B_sum") + 64.0 * mean(c, "TCC_EA0_RDREQ_64B_sum") + 32.0 * mean(c, "TCC_EA0_RDREQ_32B_sum")
            wr = 64.0 * mean(c, "TCC_EA0_WRREQ_64B_sum") + 32.0 * (mean(c, "TCC_EA0_WRREQ_sum") - mean(c, "TCC_EA0_WRREQ_64B_sum"))
            out["kernels"][tag] = {"kernel": name, "read_bytes": int(rd), "write_bytes": int(wr),
                                   "bytes_per_launch": int(rd + wr), "commit": commit}
            if mean(c, "FETCH_SIZE"):
                out["kernels"][tag]["fetch_size_kb_raw"] = round(mean(c, "FETCH_SIZE"))
            if mean(c, "TCC_HIT_sum") is not None and mean(c, "TCC_MISS_sum") is not None:
                out["kernels"][tag]["l2_hit_rate"] = round(mean(c, "TCC_HIT_sum") / (mean(c, "TCC_HIT_sum") + mean(c, "TCC_MISS_sum")), 3)
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

#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc csv output (counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections
def main(root):
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k, cs in agg.items():
            if not any(t in k for t in ("deform_", "prefilter", "hot_", "wave_", "k1_", "k2_", "k1z_")): continue
            print("  ", k)
            for c, v in cs.items():
                print("      %-24s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
if __name__ == "__main__":
    main(sys.argv[1])

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
            if "deform_" not in k and "prefilter" not in k and "hot_" not in k and "wave_" not in k: continue
            print("  ", k)
            for c, v in cs.items():
                print("      %-24s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
if __name__ == "__main__":
    main(sys.argv[1])

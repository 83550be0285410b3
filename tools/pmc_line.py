import sys, csv, glob, collections
# usage: pmc_line.py <dir> <kernel substring>  -> one line of mean counters
root, key = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(root, key, " ".join("%s=%.3g" % (k, sum(v) / len(v)) for k, v in sorted(agg.items())))

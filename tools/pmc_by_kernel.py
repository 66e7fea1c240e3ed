"""Sums a rocprofv3 counter_collection.csv by kernel and counter: average per launch.
Usage: python tools/pmc_by_kernel.py <dir with *counter_collection.csv> [more dirs]"""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        tot = collections.defaultdict(float); calls = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
        print("#", f)
        for k in sorted(calls):
            print("%-72s launches %3d " % (k, len(calls[k])) + "  ".join("%s=%.4g" % (c, v / len(calls[k])) for (kk, c), v in sorted(tot.items()) if kk == k))

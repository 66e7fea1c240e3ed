"""Prints per-dispatch counter values of a tools/prof_pmc.sh pass, rt:: kernels in launch order.
python tools/pmc_table.py <tag> [max rows]"""
import collections, csv, glob, sys
tag = sys.argv[1]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = glob.glob("gpurun_out/pmc_%s/vm/*_counter_collection.csv" % tag)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "rt::" not in r["Kernel_Name"]:
        continue
    key = int(r["Dispatch_Id"])
    name = r["Kernel_Name"].split("(")[0].replace("void rt::", "")
    rows.setdefault(key, {"name": name})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in rows.values() for k in v if k != "name"})
print("%-32s " % "kernel" + " ".join("%14s" % n[:14] for n in names))
for i, (k, v) in enumerate(sorted(rows.items())):
    if i >= limit:
        break
    print("%-32s " % v["name"] + " ".join("%14.4g" % v.get(n, float("nan")) for n in names))

"""Summarises the CSV files a tools/prof_bench.sh run left in gpurun_out/prof_<tag>/ (kernel stats and
per-kernel averages of every counter).   python tools/prof_summary.py <tag> [out.txt]"""
import collections, csv, glob, sys

tag = sys.argv[1]
base = "gpurun_out/prof_" + tag
out = ["# rocprofv3 summary '%s' (bench.py --steps 2 --warmup 1: 48 batches of 4 spp at 1080p, cubes)\n" % tag]
rows = list(csv.DictReader(open(glob.glob(base + "/trace/vm/*_kernel_stats.csv")[0])))
out.append("\n## kernel stats (--kernel-trace --stats)\n%-44s %6s %14s %12s %7s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for r in rows:
    if "rt::" in r["Name"]:
        short = r["Name"].split("(")[0].replace("void ", "")
        out.append("%-44s %6s %14s %12.0f %7s\n" % (short, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
out.append("\n## PMC passes (separate runs; average per launch)\n")
for d in sorted(glob.glob(base + "/pmc_*")):
    f = glob.glob(d + "/vm/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rt::" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in sorted(agg.items()):
        out.append("%-44s " % k + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(dd.items())) + "\n")
text = "".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)

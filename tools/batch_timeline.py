"""Per-launch timeline of the last batch in a rocprofv3 kernel trace (tools/prof_bench.sh): start, duration, gap.
Usage: python tools/batch_timeline.py <tag>"""
import csv, glob, sys
f = glob.glob("gpurun_out/prof_%s/trace/vm/*_kernel_trace.csv" % sys.argv[1])[0]
rows = [r for r in csv.DictReader(open(f)) if "rt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "k_first" in r["Kernel_Name"]][-1]
t0 = int(rows[first]["Start_Timestamp"])
prev = None
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", "")
    print("%-28s start %9.1f us  dur %8.1f us  gap %6.1f us" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
    if "k_resolve" in name:
        break

"""Per-dispatch timeline of the last batch in a tools/prof_trace.sh run.  python tools/trace_batch.py <tag>"""
import csv, glob, sys
f = glob.glob("gpurun_out/trace_%s/vm/*_kernel_trace.csv" % sys.argv[1])[0]
rows = [r for r in csv.DictReader(open(f)) if "rt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("void rt::", "").replace("rt::", ""),
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows]
starts = [i for i, (n, d) in enumerate(seq) if n.startswith("k_extend<true")]
i0 = starts[-1]
tot = 0.0
for n, d in seq[i0:]:
    print("%-36s %8.3f ms" % (n, d))
    tot += d
print("total %.3f ms" % tot)

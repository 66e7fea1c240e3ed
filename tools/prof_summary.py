"""Summarises the CSV files a tools/prof_bench.sh run left in gpurun_out/prof_<tag>/: kernel stats, per-kernel
averages of every counter, and (third argument) a JSON with the per-launch counters of each kernel for bench.py,
stamped with the fingerprint of the kernel sources they were measured on.
    python tools/prof_summary.py <tag> [summary.txt [counters.json]]"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_fingerprint  # noqa: E402

tag = sys.argv[1]
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
try:
    command = open(os.path.join(base, "command.txt")).read().strip()
except OSError:
    command = "python bench.py --steps 2 --warmup 1 --spp-per-step 512 --no-cpu-baseline --no-bruteforce-pass"
out = ["# rocprofv3 summary '%s': %s\n# (3 passes of 512 spp each at 1920x1080 unless the command says otherwise: 3 launches of "
       "the depth-0 kernel)\n" % (tag, command)]
def newest(pattern):
    """gpurun merges every run's files into the same directories: take the latest."""
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


rows = list(csv.DictReader(open(newest(base + "/trace/*/*_kernel_stats.csv"))))
out.append("\n## kernel stats (--kernel-trace --stats)\n%-60s %6s %14s %12s %7s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for r in rows:
    if "rt::" in r["Name"]:
        short = r["Name"].split("(")[0].replace("void ", "")
        out.append("%-60s %6s %14s %12.0f %7s\n" % (short, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
out.append("\n## PMC passes (separate runs; average per launch; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 prints them:\n"
           "## on gfx950 FETCH_SIZE counts 64 bytes for every 128-byte read, so wide reads are to be doubled)\n")
per_kernel = collections.defaultdict(dict)
for d in sorted(glob.glob(base + "/pmc_*")):
    f = [newest(d + "/*/*_counter_collection.csv")]
    if not f[0]:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rt::" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in sorted(agg.items()):
        out.append("%-60s " % k + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(dd.items())) + "\n")
        for c, v in dd.items():
            per_kernel[k][c] = sum(v) / len(v)
text = "".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
if len(sys.argv) > 3:
    js = {"_meta": {"command": command, "kernel_source_fingerprint": kernel_source_fingerprint(), "tag": tag}}
    for k, c in per_kernel.items():
        e = {}
        if "FETCH_SIZE" in c:
            e["FETCH_SIZE_KB"] = c["FETCH_SIZE"]
        if "WRITE_SIZE" in c:
            e["WRITE_SIZE_KB"] = c["WRITE_SIZE"]
        for name in ("TCC_HIT_sum", "TCC_MISS_sum", "TA_BUSY_avr", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"):
            if name in c:
                e[name] = c[name]
        if c.get("GRBM_GUI_ACTIVE", 0) > 0:
            e["gpu_cycles"] = c["GRBM_GUI_ACTIVE"] / 8.0         # the counter sums the 8 XCDs
        if "SQ_INSTS_VALU" in c:
            e["valu_wave_insts"] = c["SQ_INSTS_VALU"]
        js[k.replace("rt::", "")] = e
    json.dump(js, open(sys.argv[3], "w"), indent=1, sort_keys=True)

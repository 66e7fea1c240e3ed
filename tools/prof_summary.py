"""Summarises the CSV files a tools/prof_bench.sh run left in gpurun_out/prof_<tag>/: kernel stats, per-kernel
averages of every counter, and (third argument) a JSON with the per-launch HBM traffic and vector-unit occupancy of
each kernel for bench.py.     python tools/prof_summary.py <tag> [summary.txt [traffic.json]]"""
import collections, csv, glob, json, sys

tag = sys.argv[1]
base = "gpurun_out/prof_" + tag
out = ["# rocprofv3 summary '%s': python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bruteforce-pass\n"
       "# (cubes, 1920x1080, RECURSION 6; 3 x 64 spp, one pass of 64 spp each: 3 launches of the depth-0 kernel)\n" % tag]
rows = list(csv.DictReader(open(glob.glob(base + "/trace/vm/*_kernel_stats.csv")[0])))
out.append("\n## kernel stats (--kernel-trace --stats)\n%-44s %6s %14s %12s %7s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for r in rows:
    if "rt::" in r["Name"]:
        short = r["Name"].split("(")[0].replace("void ", "")
        out.append("%-44s %6s %14s %12.0f %7s\n" % (short, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
out.append("\n## PMC passes (separate runs; average per launch)\n")
per_kernel = collections.defaultdict(dict)
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
        for c, v in dd.items():
            per_kernel[k][c] = sum(v) / len(v)
text = "".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
if len(sys.argv) > 3:
    js = {}
    for k, c in per_kernel.items():
        e = {}
        if "FETCH_SIZE" in c:
            e["FETCH_SIZE_KB"] = c["FETCH_SIZE"]
        if "WRITE_SIZE" in c:
            e["WRITE_SIZE_KB"] = c["WRITE_SIZE"]
        if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            # SQ_ACTIVE_INST_VALU counts quad-cycles (4 clocks) summed over waves; a wave64 vector instruction
            # without packed math occupies its SIMD for one of them (tools/microbench_valu.hip: 4.2-4.7 clocks
            # per instruction and SIMD sustained).  GRBM_GUI_ACTIVE sums the 8 XCDs; the chip has 1024 SIMDs.
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0
            e["gpu_cycles"] = cycles
            e["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * 1024.0)
            if "SQ_INSTS_VALU" in c:
                e["valu_wave_insts"] = c["SQ_INSTS_VALU"]
        js[k.replace("rt::", "")] = e
    json.dump(js, open(sys.argv[3], "w"), indent=1, sort_keys=True)

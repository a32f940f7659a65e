#!/usr/bin/env python3
"""Per-kernel averages of a few SQ counters over `tools/kernel_bench.py <args>` (one rocprofv3 --pmc pass with --kernel-trace only).
Usage (GPU box; this process never touches the GPU):
  python3 tools/sq_counters.py SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY -- 65536 128 128 --only mfma
SQ_WAIT_ANY = wave parked on s_waitcnt / barrier; SQ_WAIT_INST_ANY = issue stall (pipe busy, dependency); SQ_ACTIVE_INST_ANY =
issuing; they are disjoint and sum to ~SQ_WAVE_CYCLES (quad-cycle units, MI355X_MICROARCH.md)."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
i = sys.argv.index("--")
counters = sys.argv[1].split(",")
d = os.path.join(ROOT, "gpurun_out", "sq_pmc")
subprocess.run(["rm", "-rf", d]); os.makedirs(d)
cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "s", "--",
       "python3", os.path.join(ROOT, "tools", "kernel_bench.py")] + sys.argv[i + 1:]
with open(os.path.join(d, "run.log"), "w") as log:
    subprocess.run(cmd, check=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=log, stderr=subprocess.STDOUT)
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
acc = {}
for row in csv.DictReader(open(cc)):
    name = row["Kernel_Name"]
    if "cmtfpls::" not in name:
        continue
    short = name.split("cmtfpls::")[1].split("(")[0]
    acc.setdefault(short, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for k, v in acc.items():
    avg = {c: sum(x) / len(x) for c, x in v.items()}
    wc = avg.get("SQ_WAVE_CYCLES")
    frac = {c: round(a / wc, 3) for c, a in avg.items() if wc and c != "SQ_WAVE_CYCLES"}
    print(json.dumps({"kernel": k, "dispatches": len(next(iter(v.values()))), "avg": {c: round(a) for c, a in avg.items()}, "fraction_of_wave_cycles": frac}))

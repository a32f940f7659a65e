#!/usr/bin/env python3
"""Matrix-pipe utilisation of the two f64 MFMA kernels from the SQ counters -> profiles/<name>.json.

One rocprofv3 pass (--pmc with --kernel-trace only, as the MI355X guide prescribes) over
`tools/kernel_bench.py --only mfma`: SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipe is busy, summed over the
SIMDs), SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE.  utilisation = MFMA busy cycles / (1024 SIMDs x effective clock x kernel
time); effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time (MI355X_MICROARCH.md, DVFS: the chip throttles under
f64 MFMA load).  Run on the GPU box:  python3 tools/mfma_utilisation.py [out.json]   (this process never touches the
GPU);  python3 tools/mfma_utilisation.py --parse [out.json] re-reads CSVs collected earlier."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]


def main():
    args = [a for a in sys.argv[1:] if a != "--parse"]
    out = args[0] if args else os.path.join(ROOT, "profiles", "mfma_utilisation.json")
    d = os.path.join(ROOT, "gpurun_out", "mfma_pmc")
    os.makedirs(d, exist_ok=True)
    if "--parse" not in sys.argv:
        cmd = ["rocprofv3", "--pmc"] + COUNTERS + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "m", "--",
                                                   "python3", os.path.join(ROOT, "tools", "kernel_bench.py"), "--only", "mfma"]
        with open(os.path.join(d, "run.log"), "w") as log:
            subprocess.run(cmd, check=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=log, stderr=subprocess.STDOUT)
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    dur = {}
    for row in csv.DictReader(open(kt)):
        dur[row["Dispatch_Id"]] = (row["Kernel_Name"], float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    acc = {}
    for row in csv.DictReader(open(cc)):
        name = row["Kernel_Name"]
        for key in ("xcov_kernel<", "mttkrp_kernel<", "mttkrp_jk_kernel<", "xcov_mixed_kernel<", "mttkrp_mixed_kernel<"):
            if "cmtfpls::" + key in name:
                rec = acc.setdefault(key[:-1], {}).setdefault(row["Dispatch_Id"], {})
                rec[row["Counter_Name"]] = float(row["Counter_Value"])
    res = {}
    for kern, disp in acc.items():
        rows = [(dur[i][1], c) for i, c in disp.items() if i in dur and all(k in c for k in COUNTERS)]
        if not rows:
            continue
        t_ns = sum(r[0] for r in rows) / len(rows)
        busy = sum(r[1]["SQ_VALU_MFMA_BUSY_CYCLES"] for r in rows) / len(rows)
        gui = sum(r[1]["GRBM_GUI_ACTIVE"] for r in rows) / len(rows)
        clock_ghz = gui / 8.0 / t_ns                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; cycles per ns
        res[kern] = {"dispatches": len(rows), "duration_us": t_ns / 1e3, "mfma_busy_cycles": busy, "grbm_gui_active": gui,
                     "effective_clock_GHz": clock_ghz,
                     "mfma_utilisation_at_effective_clock": busy / (1024 * clock_ghz * t_ns),
                     "mfma_utilisation_at_2.4GHz": busy / (1024 * 2.4 * t_ns)}
    json.dump({"method": __doc__.split("\n\n")[1].replace("\n", " "), "workload": "65536x128x128 f32, M=16, R=10 (tools/kernel_bench.py --only mfma)",
               "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

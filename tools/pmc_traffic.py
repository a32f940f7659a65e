#!/usr/bin/env python3
"""HBM traffic per launch of the X sweeps from the PMC counters -> profiles/pmc_traffic.json.

Two separate rocprofv3 passes (FETCH_SIZE, then WRITE_SIZE; each with --kernel-trace only, as the
MI355X guide prescribes) over `bench.py --steps 5 --warmup 2 --no-cpu --graphs 0`; the counter unit is
KB (x1024); FETCH_SIZE is doubled (gfx950 reports half of the bytes of a wide coalesced streaming read,
MI355X_MICROARCH.md); WRITE_SIZE is taken as is; per launch = mean over the kernel's dispatches.
Run on the GPU box:  python3 tools/pmc_traffic.py [out.json]
(this process never touches the GPU; rocprofv3 is started with python3 directly after `--`)."""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"contract_vec_kernel": 1, "score_kernel": 1, "deflate_kernel": 2, "deflate_rows_kernel": 2,
           "deflate_contract_kernel": 2, "deflate_contract_rows_kernel": 2, "center_kernel": 2, "center_rows_kernel": 2,
           "score_deflate_kernel": 2, "xcov_kernel": 1, "mttkrp_kernel": 1, "score_contract_rows_kernel": 1}
ROUND = "r04"
# bench.py refuses the profile once any of these changed (the kernels it describes are compiled from them)
SOURCES = ["cmtf_pls_amd/csrc/sweeps.hip", "cmtf_pls_amd/csrc/common.hpp"]
XBYTES = 65536 * 128 * 128 * 4


def one_pass(counter, outdir):
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "p", "--",
           "python3", os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu", "--graphs", "0", "--no-ceilings", "--repeats", "1", "--no-north-star"]
    with open(os.path.join(outdir, "run.log"), "w") as log:
        subprocess.run(cmd, check=True, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT)
    return parse(counter, outdir)


def parse(counter, outdir):
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        for key in KERNELS:
            name = row["Kernel_Name"]
            if f"cmtfpls::{key}<float" in name and not (key == "contract_vec_kernel" and "<float, 2" in name):
                acc.setdefault(key, []).append(float(row["Counter_Value"]) * 1024.0)
    return acc


def split_rows(out):
    """--split-rows: the same two passes over tools/split_row_time.py 32768 256 256 (the north-star row, one read: the row split
    over four workgroups) -> HBM bytes per launch of score_contract_split_kernel against X's 8.59 GB."""
    scratch = os.path.join(ROOT, "gpurun_out", "pmc_split")
    xbytes = 32768 * 256 * 256 * 4
    vals = {}
    for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        outdir = os.path.join(scratch, sub)
        os.makedirs(outdir, exist_ok=True)
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "p", "--",
               "python3", os.path.join(ROOT, "tools", "split_row_time.py"), "32768", "256", "256"]
        with open(os.path.join(outdir, "run.log"), "w") as log:
            subprocess.run(cmd, check=True, cwd=os.path.join(ROOT, "tools"), env=dict(os.environ, TMPDIR="/tmp"), stdout=log, stderr=subprocess.STDOUT)
        files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
        acc = [float(r["Counter_Value"]) * 1024.0 for r in csv.DictReader(open(files[0]))
               if r["Counter_Name"] == counter and "score_contract_split_kernel<float" in r["Kernel_Name"]]
        vals[counter] = (sum(acc) / len(acc), len(acc))
    f, w = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    doc = {"round": ROUND, "workload": "score_contract on 32768x256x256 f32 (tools/split_row_time.py), 1 GPU", "kernel": "score_contract_split_kernel<float, 4, true, 1>",
           "fetch_raw_bytes": f, "fetch_corrected_bytes": 2 * f, "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w, "dispatches": vals["FETCH_SIZE"][1],
           "algorithmic_bytes": xbytes, "ratio": (2 * f + w) / xbytes,
           "method": "two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only); unit KB; FETCH_SIZE doubled (gfx950 note of MI355X_MICROARCH.md)"}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({"score_contract_split_kernel": round(doc["ratio"], 4)}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--split-rows":
        return split_rows(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "pmc_traffic_split_rows.json"))
    scratch = os.path.join(ROOT, "gpurun_out", "pmc")
    if len(sys.argv) > 1 and sys.argv[1] == "--parse":          # re-parse CSVs collected earlier (no GPU needed)
        out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "pmc_traffic.json")
        fetch = parse("FETCH_SIZE", os.path.join(scratch, "fetch"))
        write = parse("WRITE_SIZE", os.path.join(scratch, "write"))
    else:
        out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "pmc_traffic.json")
        fetch = one_pass("FETCH_SIZE", os.path.join(scratch, "fetch"))
        write = one_pass("WRITE_SIZE", os.path.join(scratch, "write"))
    kernels = {}
    for key, passes in KERNELS.items():
        if key not in fetch or key not in write:
            continue
        f = sum(fetch[key]) / len(fetch[key])
        w = sum(write[key]) / len(write[key])
        rec = {"fetch_raw_bytes": f, "fetch_corrected_bytes": 2 * f, "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w,
               "dispatches": len(fetch[key])}
        if passes:
            rec["algorithmic_bytes"] = passes * XBYTES
            rec["ratio"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes"]
        kernels[key] = rec
    doc = {"round": ROUND, "method": __doc__.split("\n\n")[1].replace("\n", " "), "workload": "cfg2 65536x128x128 f32, 1 GPU",
           "command": "python3 tools/pmc_traffic.py", "kernels": kernels,
           "source_sha256": {rel: hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest() for rel in SOURCES}}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: round(v.get("ratio", 0), 4) for k, v in kernels.items()}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turns a rocprofv3 --pmc FETCH_SIZE pass (rocpd sqlite) into profiles/pmc_traffic.json, stamped with the hash of the
dominant kernel's sources (bench.kernel_code_hash): bench.py quotes `roofline.traffic` from that file only while the hash
matches the code it runs, and reports null otherwise.

usage (on the GPU box, counters in their OWN pass, per MI355X_MICROARCH.md "HBM"):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python bench.py --steps 4 --warmup 1 --repeats 1 \
        --no-cpu-baseline --no-parity-check --no-context --no-prefill
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/*/*_results.db profiles/r02_pmc_fetch_size.md
FETCH_SIZE is in KiB and on gfx950 reports exactly 1/2 of the bytes of a wide coalesced streaming read: x 1024 x 2."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KERNEL = "k_gateup_q"  # the dominant kernel of the default bench line
ALGO_BYTES = 66191360   # 2 x 14336 x 4096 / 32 x 18 + 4 x 4096 + 4 x 2 x 14336


def main():
    dbs = [p for a in sys.argv[1:] if a.endswith(".db") and not a.startswith("--") for p in glob.glob(a)]
    md_out = next((a for a in sys.argv[1:] if a.endswith(".md")), None)
    if not dbs:
        sys.exit("no rocpd database given")
    table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_pmc.py"), dbs[0]], capture_output=True, text=True).stdout
    kib = None
    for line in table.splitlines():
        cells = [c.strip() for c in line.split("|")]
        if len(cells) > 4 and KERNEL in cells[1] and cells[2] == "FETCH_SIZE":
            kib = float(cells[4])
    if kib is None:
        sys.exit(f"{KERNEL} not found in the PMC table")
    hbm = int(round(kib * 1024 * 2))
    # optional second database (--trace=...db): the rocprofv3 --kernel-trace of the same command -> the dominant kernel's average
    # duration, quoted by bench.py as roofline.frac_rocprof beside its own event-timed fraction
    trace = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--trace=")), None)
    trace_md = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--trace-md=")), None)
    avg_us = None
    if trace:
        import sqlite3

        c = sqlite3.connect(trace)
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
        d = [e - s0 for n, s0, e in c.execute(f"select {name_col}, start, end from kernels") if KERNEL + "<" in n or KERNEL + "(" in n]
        if d:
            avg_us = round(sum(d) / len(d) / 1e3, 3)
    # the tracked copy of the table: gpu_profile.sh writes gpurun_out/<tag>_pmc_fetch_size.md, which is committed as profiles/<same name>
    tracked = "profiles/" + os.path.basename(md_out) if md_out else os.path.basename(dbs[0])
    out = {"source": tracked + " (rocprofv3 --pmc FETCH_SIZE, separate pass; FETCH_SIZE KiB x 1024 x 2 gfx950 correction)",
           "dominant_kernel": KERNEL + "<Q4_0>", "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": ALGO_BYTES,
           "ratio": round(hbm / ALGO_BYTES, 4), "kernel_code_hash": bench.kernel_code_hash()}
    if avg_us:
        out["rocprof_avg_launch_us"] = avg_us
        out["rocprof_source"] = ("profiles/" + os.path.basename(trace_md) if trace_md else os.path.basename(trace)) + " (rocprofv3 --kernel-trace --stats of the bench command)"
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    if md_out:
        with open(md_out, "w") as f:
            f.write(f"# PMC pass: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py (kernel code hash {out['kernel_code_hash']})\n\n"
                    f"{KERNEL}: {kib:.1f} KiB -> {hbm / 1e6:.2f} MB per launch (algorithmic {ALGO_BYTES / 1e6:.2f} MB, ratio {out['ratio']}).\n\n" + table)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

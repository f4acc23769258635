#!/usr/bin/env python3
"""Turn one `tools/profile_round.sh` session (gpurun_out/prof_round/) into the committed evidence:
profiles/<round>/*.csv (kernel stats, per-dispatch counters, calibration) and profiles/traffic.json
(HBM bytes per launch of the tile kernel, read by bench.py for `roofline.traffic`).

Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: counters come in KB
(x1024); FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled - and the factor is re-measured in
the same session on tools/kbench's float4 copy of a known byte count.

    python tools/derive_traffic.py [gpurun_out/prof_round] [r01]
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_round")
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
COPY_BYTES = 2_684_354_560            # tools/kbench copy: 160 Mi float4 elements


def counters(path, kernel_substr):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel_substr in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return vals


def mean(v):
    return sum(v) / len(v) if v else 0.0


out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_ATOMIC_sum"):
    p = os.path.join(src, f"pmc_{c}", "pmc_counter_collection.csv")
    shutil.copy(p, os.path.join(dst, f"pmc_{c}.csv"))
    out[c] = {"tile": counters(p, "seg_tile_kernel"), "fixup": counters(p, "seg_fixup_kernel")}
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(src, f"cal_{c}", "cal_counter_collection.csv")
    shutil.copy(p, os.path.join(dst, f"cal_{c}.csv"))
    cal[c] = mean(counters(p, "copy4_kernel"))
for name in ("bench_kernel_stats.csv", "bench_domain_stats.csv"):
    shutil.copy(os.path.join(src, "kt", name), os.path.join(dst, name))
for name in ("bench_under_kernel_trace.json", "bench_unprofiled.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, name))

fetch_factor = COPY_BYTES / (cal["FETCH_SIZE"] * 1024)      # ~2.0 on gfx950
write_factor = COPY_BYTES / (cal["WRITE_SIZE"] * 1024)      # ~1.0
fetch = mean(out["FETCH_SIZE"]["tile"]) * 1024 * 2
write = mean(out["WRITE_SIZE"]["tile"]) * 1024
alg = 10_000_000 * (4 * 64 + 8) + 1_000_000 * 4 * 64
kernel = None
with open(os.path.join(src, "kt", "bench_kernel_stats.csv")) as f:
    for r in csv.DictReader(f):
        if "seg_tile_kernel" in r["Name"]:
            kernel = {"name": r["Name"].split("::")[-2].split("(")[0] if "::" in r["Name"] else r["Name"],
                      "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                      "max_ns": float(r["MaxNs"])}
        if "seg_fixup_kernel" in r["Name"]:
            fix_ns = float(r["AverageNs"])
res = {
    "kernel": "seg_tile_kernel<float, 4, false, 0, false, 3, 3, 16>",
    "workload": "BASELINE.json configs[1]: index_scatter sorted sum, power-law 10M edges -> 1M nodes, feat=64",
    "hbm_bytes_per_launch": int(fetch + write),
    "fetch_bytes_per_launch": int(fetch),
    "write_bytes_per_launch": int(write),
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": (fetch + write) / alg,
    "raw": {"FETCH_SIZE_KB_mean": mean(out["FETCH_SIZE"]["tile"]), "WRITE_SIZE_KB_mean": mean(out["WRITE_SIZE"]["tile"]),
            "dispatches": len(out["FETCH_SIZE"]["tile"])},
    "calibration": {"copy_bytes": COPY_BYTES, "FETCH_SIZE_KB": cal["FETCH_SIZE"], "WRITE_SIZE_KB": cal["WRITE_SIZE"],
                    "bytes_per_FETCH_SIZE_KB": fetch_factor * 1024, "bytes_per_WRITE_SIZE_KB": write_factor * 1024},
    "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_ATOMIC_sum in separate passes of `python bench.py "
              "--steps 5 --warmup 2` (tools/profile_round.sh; profiles/%s/pmc_*.csv); units KB (x1024); FETCH_SIZE "
              "doubled per MI355X_MICROARCH.md 'HBM' (gfx950 tallies 128-B requests at 64 B), factor re-measured in the "
              "same session on tools/kbench's float4 copy (profiles/%s/cal_*.csv)" % (rnd, rnd),
    "atomics": {"TCC_EA0_ATOMIC_sum_tile": mean(out["TCC_EA0_ATOMIC_sum"]["tile"]),
                "TCC_EA0_ATOMIC_sum_fixup": mean(out["TCC_EA0_ATOMIC_sum"]["fixup"])},
    "fixup_kernel": {"fetch_bytes_per_launch": int(mean(out["FETCH_SIZE"]["fixup"]) * 2048),
                     "write_bytes_per_launch": int(mean(out["WRITE_SIZE"]["fixup"]) * 1024),
                     "average_ns_kernel_trace": fix_ns},
    "kernel_trace": kernel,
    "achieved_GBps_kernel_trace": alg / kernel["average_ns"],
}
json.dump(res, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))

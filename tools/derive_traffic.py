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
rnd = sys.argv[2] if len(sys.argv) > 2 else "r06"
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


HEADLINE = "seg_tile_kernel<float, 4, false, 0, false, 3, 3, 16>"
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_ATOMIC_sum"):
    p = os.path.join(src, f"pmc_{c}", "pmc_counter_collection.csv")
    shutil.copy(p, os.path.join(dst, f"pmc_{c}.csv"))
    out[c] = {"tile": counters(p, HEADLINE), "fixup": counters(p, "seg_fixup_kernel")}
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
        if HEADLINE in r["Name"]:
            kernel = {"name": r["Name"].split("::")[-2].split("(")[0] if "::" in r["Name"] else r["Name"],
                      "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                      "max_ns": float(r["MaxNs"])}
        if "seg_fixup_kernel" in r["Name"]:
            fix_ns = float(r["AverageNs"])
res = {
    "kernel": HEADLINE,
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


# ---- the kernels of bench.py's `secondary` workloads, one profiling pass set per workload (tools/profile_round.sh):
#      kernel-trace time, fabric-side bytes (FETCH_SIZE x2 - 16-B-per-lane row reads, same shape as the calibration copy -
#      + WRITE_SIZE), L2 hit rate (TCC_HIT / (TCC_HIT + TCC_MISS)), against SURVEY 8(d)'s compulsory bytes.  Keys of
#      gather_kernels.json["kernels"]: the workload's own name = the kernel bench.py times for it (what bench.py's
#      `roofline.traffic` of that entry quotes), "<workload>/<comparator>" = the other kernels seen in the same passes.
import re


def per_kernel(path):
    d = {}
    if not os.path.exists(path):
        return d
    with open(path) as f:
        for r in csv.DictReader(f):
            d.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: mean(v) for k, v in d.items()}


def kernel_times(path):
    kt = {}
    if os.path.exists(path):
        with open(path) as f:
            for r in csv.DictReader(f):
                kt[r["Name"]] = {"calls": int(r["Calls"]), "average_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6}
    return kt


T16 = r"(__bf16|bf16_t|hip_bfloat16|__hip_bfloat16|_Float16|__half|half_t)"
WORKLOADS = {
    "gws_cfg3": [("", r"seg_tile_kernel<float, 4, true, 1,"), ("/rocsparse csr_nnz_split", r"csrmmnt_nnz_split_main_kernel"),
                 ("/rocsparse csr_merge_path", r"csrmmnt_merge_path_main_kernel"), ("/rocsparse csr_row_split", r"csrmmnt_row_split")],
    "gws_cfg3_local": [("", r"seg_tile_kernel<float, 4, true, 1,"), ("/rocsparse csr_nnz_split", r"csrmmnt_nnz_split_main_kernel"),
                       ("/rocsparse csr_merge_path", r"csrmmnt_merge_path_main_kernel")],
    "gws_cfg3_powerlaw_src": [("", r"seg_tile_kernel<float, 4, true, 1,"), ("/rocsparse csr_nnz_split", r"csrmmnt_nnz_split_main_kernel"),
                              ("/rocsparse csr_merge_path", r"csrmmnt_merge_path_main_kernel")],
    "gws_cfg3_blockmodel_asis": [("", r"seg_tile_kernel<float, 4, true, 1,")],
    "gws_cfg3_blockmodel_renum": [("", r"seg_tile_kernel<float, 4, true, 1,"), ("/gather_rows (x in, y out)", r"gather_rows_kernel<float")],
    "mh_spmm_cfg4_powerlaw_src": [("", r"seg_slab_kernel<float, 2, true,"), ("/per-edge [nnz,H]", r"seg_tile_kernel<float, 4, true, 2,")],
    "mh_spmm_cfg4": [("", r"seg_slab_kernel<float, 2, true,"), ("/per-edge [nnz,H]", r"seg_tile_kernel<float, 4, true, 2,"),
                     ("/per-edge [H,nnz]", r"seg_tile_kernel<float, 4, true, 3,"), ("/phase A edge keys", r"plan_edge_keys_kernel"),
                     ("/phase A edge out", r"plan_edge_out_kernel")],
    # (rocprofv3 leaves the __bf16 instantiations mangled: DF16b)
    "gather_scatter_cfg5": [("", r"seg_tile_kernel<float, 4, true, 0,")],
    "gws_cfg3_bf16": [("", r"seg_tile_kernel(<" + T16 + r", 8, true, 1,|IDF16bLi8ELb1ELi1E)")],
    # (round 6: the matrix-core SpMM serves this workload; its gated vector-ALU twin is launched behind it and returns at once)
    "mh_spmm_cfg4_bf16": [("", r"seg_slab_spmm_mfma_kernel"), ("/gated twin (returns at once)", r"seg_slab_wrow_kernel(<" + T16 + r", 2, 4|IDF16bLi2ELi4E)"),
                          ("/finite-table check", r"slab_nonfinite_kernel"),
                          ("/per-edge [nnz,H]", r"seg_tile_kernel(<" + T16 + r", 8, true, 2,|IDF16bLi8ELb1ELi2E)")],
}
gather = {}
# (gpurun MERGES a session's files into gpurun_out/: what an earlier round's session left there under other workload names must not be
#  taken for this session's - everything older than the session's first file by more than a minute is ignored)
session_t0 = os.path.getmtime(os.path.join(src, "bench_unprofiled.json")) - 60
for name in ("bench_driver_cmd.json", "bench_driver_detail.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
for w, wanted in WORKLOADS.items():
    ktp = os.path.join(src, f"kt_{w}", "bench_kernel_stats.csv")
    if os.path.exists(ktp) and os.path.getmtime(ktp) < session_t0:
        print(f"# {w}: stale files of an earlier session ignored", file=sys.stderr)
        continue
    kt = kernel_times(ktp)
    if not kt:
        continue
    shutil.copy(os.path.join(src, f"kt_{w}", "bench_kernel_stats.csv"), os.path.join(dst, f"kernel_stats__{w}.csv"))
    pm = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_ATOMIC_sum"):
        pth = os.path.join(src, f"pmc_{c}__{w}", "pmc_counter_collection.csv")
        pm[c] = per_kernel(pth)
        if os.path.exists(pth) and c != "TCC_EA0_ATOMIC_sum":      # (the per-kernel atomic count is in gather_kernels.json)
            shutil.copy(pth, os.path.join(dst, f"pmc_{c}__{w}.csv"))
    try:
        sec = json.load(open(os.path.join(src, f"kt_{w}.json"))).get("secondary", {}).get(w, {})
    except Exception:
        sec = {}
    for suffix, pat in wanted:
        rx = re.compile(pat)
        name = next((k for k in kt if rx.search(k)), None)
        if name is None:
            continue
        g = {"kernel": name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0], **kt[name]}
        pick = lambda c: next((v for k, v in pm.get(c, {}).items() if rx.search(k)), None)  # noqa: E731
        fetch, write, hit, miss, atom = (pick(c) for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_ATOMIC_sum"))
        if fetch is not None and write is not None:
            g["fabric_bytes_per_launch"] = int(fetch * 2048 + write * 1024)
            g["FETCH_SIZE_KB_raw"], g["WRITE_SIZE_KB_raw"] = fetch, write
        if hit is not None and miss is not None and hit + miss > 0:
            g["l2_hit_rate"] = hit / (hit + miss)
        if atom is not None:
            g["TCC_EA0_ATOMIC_sum"] = atom
        comp = sec.get("compulsory_bytes") or sec.get("compulsory_bytes_rank0")
        if comp and not suffix.startswith("/phase A"):
            g["compulsory_bytes"] = comp
            g["frac_of_8TBps_on_compulsory_bytes"] = comp / (g["average_ms"] * 1e-3) / 8e12
            if "fabric_bytes_per_launch" in g:
                g["traffic_over_compulsory"] = g["fabric_bytes_per_launch"] / comp
        gather[w + suffix] = g
# configs[0] (launch-bound): the tile kernel's own duration from the kernel trace of `bench.py --only-secondary cfg1`
kt1 = kernel_times(os.path.join(src, "kt_cfg1", "bench_kernel_stats.csv"))
if kt1:
    shutil.copy(os.path.join(src, "kt_cfg1", "bench_kernel_stats.csv"), os.path.join(dst, "kernel_stats__cfg1.csv"))
    try:
        kname = json.load(open(os.path.join(src, "kt_cfg1.json")))["secondary"]["cfg1"]["kernel"]      # what the launcher picked (geot_last_kernel)
    except Exception:
        kname = "seg_tile_kernel<float, 4, false, 0, false, 3, 3, 8>"
    tile = next((v for k, v in kt1.items() if kname in k), None)
    fix = next((v for k, v in kt1.items() if "seg_fixup_kernel<float" in k), None)
    if tile:
        gather["cfg1"] = {"kernel": kname, **tile, "second_launch_us": fix["average_ms"] * 1e3 if fix else None}
if "gws_cfg3_blockmodel_asis" in gather:                # bench.py's entry `gws_cfg3_blockmodel` quotes the as-shipped kernel
    gather["gws_cfg3_blockmodel"] = dict(gather["gws_cfg3_blockmodel_asis"])
json.dump({"method": "one rocprofv3 pass set per workload (tools/profile_round.sh: bench.py --only-secondary <workload>): --kernel-trace "
                     "--stats for the times, one --pmc counter per pass for the bytes; FETCH_SIZE doubled (gfx950, 16-B-per-lane reads); "
                     "counters in KB",
           "kernels": gather}, open(os.path.join(dst, "gather_kernels.json"), "w"), indent=1)
print(json.dumps(gather, indent=1))

#!/usr/bin/env python3
"""summary.txt of one tools/cfg5_box.sh record: the box, its ceilings, the kernel on it, the translation / fabric counters of that kernel."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
lines = []
box = open(os.path.join(out, "box.txt"), errors="replace").read()
for key in ("Unique ID", "mclk", "fclk", "sclk", "VRAM Total Memory"):
    for ln in box.splitlines():
        if key in ln:
            lines.append("box: " + " ".join(ln.split()))
            break
d = json.load(open(os.path.join(out, "bench_cfg5.json")))
leg = d["secondary"]["gather_scatter_cfg5"]
r = leg["roofline"]
keys = ("box_read_ceiling_gbps", "box_sclk_mhz", "box_random_row_gbps_default_policy", "box_random_row_gbps_nt", "box_random_row_gbps_with_write_mix",
        "write_mix_run", "row_gather_gbps", "row_gather_frac_of_box_random_row", "row_gather_frac_of_box_row_mix", "moved_gbps")
lines.append(f"kernel {leg['kernel']}: {leg['kernel_ms_rank0']:.3f} ms (HIP events, 3 launches), {leg['ms_per_step']:.3f} ms per step as dispatched")
lines.append(" ".join(f"{k}={r.get(k):.4g}" if isinstance(r.get(k), float) else f"{k}={r.get(k)}" for k in keys))
for dd in sorted(glob.glob(out + "/pmc_*")):
    if not os.path.isdir(dd):
        continue
    f = glob.glob(dd + "/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    vals = [float(x["Counter_Value"]) for x in csv.DictReader(open(f[0])) if "seg_tile_kernel<float, 4, true, 0" in x["Kernel_Name"]]
    lines.append(f"{os.path.basename(dd)[4:]:45s} dispatches {len(vals):2d}  mean per launch {sum(vals) / max(len(vals), 1):.6g}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

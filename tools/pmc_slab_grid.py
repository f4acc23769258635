#!/usr/bin/env python3
"""Table of tools/pmc_slab_grid.sh: per (slab MiB, window) the second launch's duration, L2 hit rate and fabric fetch bytes."""
import csv
import glob
import os
import sys

out, seq = sys.argv[1], sys.argv[2].split(",")


def rows(pattern, value):
    f = glob.glob(os.path.join(out, pattern), recursive=True)
    if not f:
        return []
    r = [x for x in csv.DictReader(open(f[0])) if "seg_slab_kernel" in x["Kernel_Name"]]
    key = "Dispatch_Id" if r and "Dispatch_Id" in r[0] else "Start_Timestamp"
    r.sort(key=lambda x: int(x[key]))
    return [value(x) for x in r]


dur = rows("kt/**/*kernel_trace.csv", lambda x: (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6)
hit = rows("pmc_TCC_HIT_sum/**/*counter_collection.csv", lambda x: float(x["Counter_Value"]))
miss = rows("pmc_TCC_MISS_sum/**/*counter_collection.csv", lambda x: float(x["Counter_Value"]))
fetch = rows("pmc_FETCH_SIZE/**/*counter_collection.csv", lambda x: float(x["Counter_Value"]))
print(f"launches seen: trace {len(dur)}, hit {len(hit)}, miss {len(miss)}, fetch {len(fetch)} (2 per setting, {len(seq)} settings)")
print("slab_MiB window  ms(2nd launch, under kernel trace)  L2 hit rate  fabric fetch GB (FETCH_SIZE KB x 1024 x 2: gfx950 correction)")
for i, s in enumerate(seq):
    j = 2 * i + 1
    mib, k = s.split(":")
    ms = dur[j] if j < len(dur) else float("nan")
    hr = hit[j] / (hit[j] + miss[j]) if j < len(hit) and j < len(miss) else float("nan")
    gb = fetch[j] * 1024 * 2 / 1e9 if j < len(fetch) else float("nan")
    print(f"{mib:>8s} {k:>6s}  {ms:10.3f}  {hr:10.3f}  {gb:10.2f}")

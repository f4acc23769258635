#!/bin/bash
# One profiling session for profiles/rNN (run on the MI355X box: `gpurun -- bash tools/profile_round.sh`).
# Separate rocprofv3 passes, as the micro-arch guide prescribes: kernel trace + stats, then one --pmc counter per pass
# (never combined with a trace domain), plus the copy-kernel calibration of the counters.  One set of passes PER WORKLOAD
# (the headline and each entry of bench.py's `secondary`): two workloads that run the same kernel instantiation - configs[2]
# with uniform-random and with local sources - must not share a per-kernel average.
# Everything lands in gpurun_out/prof_round/; tools/derive_traffic.py turns it into profiles/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py --full --secondary all --steps 100 --warmup 10 > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"
# ... and the driver's own command, as the driver sees it (the compact last line)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out "$OUT/bench_driver_detail.json" > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"
# ---- headline (BASELINE.json configs[1])
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o bench -- python3 bench.py --full --steps 100 --warmup 10 --no-cpu-baseline --no-secondary \
    > "$OUT/bench_under_kernel_trace.json" 2> "$OUT/kt.err"
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- python3 bench.py --full --steps 5 --warmup 2 --no-cpu-baseline --no-secondary \
      > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err"
done
# ---- the secondary workloads, one at a time
# (gather_scatter_cfg5 = BASELINE.json configs[4]'s one-GPU shard: bench.py selects it with `--only-secondary cfg5`)
for w in ${WORKLOADS:-gws_cfg3 gws_cfg3_local mh_spmm_cfg4 gws_cfg3_bf16 mh_spmm_cfg4_bf16 gather_scatter_cfg5}; do
  sel=$w; [ $w = gather_scatter_cfg5 ] && sel=cfg5
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$w" -o bench -- python3 bench.py --full --steps 2 --warmup 1 --no-cpu-baseline --only-secondary $sel \
      > "$OUT/kt_$w.json" 2> "$OUT/kt_$w.err"
  for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${c}__$w" -o pmc -- python3 bench.py --full --steps 2 --warmup 1 --no-cpu-baseline --only-secondary $sel \
        > "$OUT/pmc_${c}__$w.json" 2> "$OUT/pmc_${c}__$w.err"
  done
done
# configs[0] is launch-bound: its kernel's own duration can only come from a trace (HIP events bracket the launch gap too)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_cfg1" -o bench -- python3 bench.py --full --steps 2 --warmup 1 --no-cpu-baseline --only-secondary cfg1 \
    > "$OUT/kt_cfg1.json" 2> "$OUT/kt_cfg1.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/cal_$c" -o cal -- ./tools/kbench copy > "$OUT/cal_$c.txt" 2> "$OUT/cal_$c.err"
done
# keep the CSV summaries only (the rocpd databases are tens of MB)
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.txt" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +4M -delete
du -sh "$OUT"; find "$OUT" -name "*.csv" | wc -l
for f in "$OUT"/*.err; do tail -n 2 "$f"; done | grep -v "^$" | head -40

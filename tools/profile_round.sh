#!/bin/bash
# One profiling session for profiles/rNN (run on the MI355X box: `gpurun -- bash tools/profile_round.sh`).
# Separate rocprofv3 passes, as the micro-arch guide prescribes: kernel trace + stats, then one --pmc
# counter per pass (never combined with a trace domain), plus the copy-kernel calibration of the counters.
# bench.py's default run covers the headline kernel AND the `secondary` workloads (gws cfg3, mh_spmm cfg4,
# rocSPARSE beside them), so every pass sees the gather-mode kernels too.
# Everything lands in gpurun_out/prof_round/; tools/derive_traffic.py turns it into profiles/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py --steps 100 --warmup 10 > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o bench -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline \
    > "$OUT/bench_under_kernel_trace.json" 2> "$OUT/kt.err"
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline \
      > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/cal_$c" -o cal -- ./tools/kbench copy > "$OUT/cal_$c.txt" 2> "$OUT/cal_$c.err"
done
# keep the CSV summaries only (the rocpd databases are tens of MB)
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.txt" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +4M -delete
du -sh "$OUT"; find "$OUT" -name "*.csv" | head -40
tail -3 "$OUT"/*.err | head -60

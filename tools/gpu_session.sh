#!/bin/bash
# GPU session driver (logs under gpurun_out/r06): `gpurun --timeout 3300 -- bash tools/gpu_session.sh [steps...]`
# Each step writes its log under gpurun_out/r02/ and is bounded by its own timeout.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
steps="${@:-tests bench kexp unsorted soak}"
for s in $steps; do
  echo "=== $s $(date +%T)"
  case $s in
    r6a)      timeout 1200 python3 -m pytest tests/test_gpu_graph_handle.py tests/test_gpu_multirank.py -m gpu -x -q > $O/pytest_r6a.log 2>&1; echo "rc=$?"; tail -8 $O/pytest_r6a.log
              ( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_driver_detail.json > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time; echo "rc=$?"
              cat $O/bench_driver_cmd.time; wc -c $O/bench_driver_cmd.json; cat $O/bench_driver_cmd.json; tail -3 $O/bench_driver_cmd.err ;;
    kexp4)    timeout 600 ./tools/kexp4 > $O/kexp4_random_rows.txt 2>&1; echo "rc=$?"; cat $O/kexp4_random_rows.txt
              ;;
    devtests) GEOT_HIP_LIB=dev timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu_devlib.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_gpu_devlib.log ;;
    mfma)     timeout 1200 python3 -m pytest tests/test_gpu_round6.py -m gpu -x -q > $O/pytest_r6_mfma.log 2>&1; echo "rc=$?"; tail -25 $O/pytest_r6_mfma.log
              for o in slab_spmm_mfma=1 slab_spmm_mfma=0 slab_spmm_mfma=1 slab_spmm_mfma=0; do
                timeout 600 python3 tools/bench_slab_cases.py --only mh --dtypes bf16 --options $o 2>&1 | grep -v amdgpu.ids
              done > $O/slab_cases_mfma_ab.txt 2>&1; cat $O/slab_cases_mfma_ab.txt ;;
    dbg)      timeout 600 python3 tools/_ab/dbg_gate2.py 2>&1 | grep -v amdgpu.ids | tail -60 ;;
    mfmaprobe) for o in slab_spmm_mfma=1 slab_spmm_mfma=1,slab_probe=1 slab_spmm_mfma=1,slab_probe=1,slab_window=-1 slab_spmm_mfma=1,slab_window=-1 slab_spmm_mfma=1,slab_blocks=4 slab_spmm_mfma=0,slab_probe=1; do
                timeout 600 python3 tools/bench_slab_cases.py --only mh --dtypes bf16 --options $o 2>&1 | grep -v amdgpu.ids | grep -v "^# done"
              done > $O/slab_cases_mfma_probe.txt 2>&1; cat $O/slab_cases_mfma_probe.txt ;;
    knock)    for o in slab_probe=0 slab_probe=1 slab_probe=3 slab_probe=5 slab_probe=7 slab_probe=9 slab_probe=2 slab_probe=4 slab_probe=8; do
                timeout 600 python3 tools/bench_slab_cases.py --only mh --dtypes bf16 --options slab_window=-1,$o 2>&1 | grep "PLAN order\|options"
              done > $O/slab_cases_mfma_knockout.txt 2>&1; cat $O/slab_cases_mfma_knockout.txt ;;
    kexp5)    timeout 600 ./tools/kexp5 > $O/kexp5_row_gather_instruction_cost.txt 2>&1; echo "rc=$?"; cat $O/kexp5_row_gather_instruction_cost.txt ;;
    slabmatrix) timeout 1500 python3 tools/bench_slab_cases.py 2>&1 | grep -v amdgpu.ids > $O/slab_cases_matrix.txt; cat $O/slab_cases_matrix.txt ;;
    rows256)  timeout 900 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -x -q 2>&1 | tail -4
              { for o in slab_spmm_mfma=1,slab_sddmm_mfma=1 slab_spmm_mfma=0,slab_sddmm_mfma=0; do
                  timeout 600 python3 tools/bench_slab_cases.py --only mh,sddmm --dtypes bf16 --mh-shape 2,64 --options $o 2>&1 | grep -v amdgpu.ids
                  timeout 600 python3 tools/bench_slab_cases.py --only mh,sddmm --dtypes bf16 --mh-shape 4,32 --options $o 2>&1 | grep -v amdgpu.ids
                  timeout 600 python3 tools/bench_slab_cases.py --only gws,sddmm --dtypes bf16 --gws-wave-cut --options $o 2>&1 | grep -v amdgpu.ids
                done; timeout 600 python3 tools/bench_slab_cases.py --only gws,sddmm --dtypes bf16 2>&1 | grep -v amdgpu.ids; } > $O/slab_cases_rows256.txt 2>&1; cat $O/slab_cases_rows256.txt ;;
    mfmaslab) for mib in 0.5 1 1.5 2 3; do for w in 0 1 2 3; do
                timeout 300 python3 tools/bench_slab_cases.py --only mh,sddmm --dtypes bf16 --slab-mib $mib --options slab_window=$w 2>&1 | grep "PLAN order\|left in plan\|^# lib" | sed "s/^# lib.*options=/# /"
              done; done > $O/slab_cases_mfma_slab_sweep.txt 2>&1; cat $O/slab_cases_mfma_slab_sweep.txt ;;
    stagemh)  timeout 900 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_round4.py tests/test_gpu_slab.py -m gpu -x -q 2>&1 | tail -4
              timeout 900 python3 tools/bench_slab_cases.py --only mh,sddmm 2>&1 | grep -v amdgpu.ids > $O/slab_cases_mh_group_local_staging.txt; cat $O/slab_cases_mh_group_local_staging.txt ;;
    stagetrace) export TMPDIR=/tmp; rm -rf $O/st; rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o st -- python3 tools/bench_slab_cases.py --only mh,sddmm --dtypes bf16 > $O/st.log 2>&1; tail -3 $O/st.log
              python3 - <<PY > $O/stage_unstage_kernel_trace.txt
import csv, glob
f = glob.glob('$O/st/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('stage', 'mfma', 'fill', 'nonfinite', 'combine')):
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e6:8.3f} ms  min {float(r['MinNs'])/1e6:8.3f}  max {float(r['MaxNs'])/1e6:8.3f}")
PY
              cat $O/stage_unstage_kernel_trace.txt; rm -rf $O/st ;;
    stageab)  timeout 900 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_round4.py tests/test_gpu_slab.py -m gpu -x -q 2>&1 | tail -3
              for m in direct tile direct tile; do echo "# GEOT_PERMUTE=$m"; GEOT_PERMUTE=$m timeout 900 python3 tools/bench_slab_cases.py --only mh,sddmm 2>&1 | grep "as the ABI\|unstage"; done > $O/slab_cases_permute_ab.txt; cat $O/slab_cases_permute_ab.txt ;;
    rows512)  { timeout 600 python3 tools/bench_slab_cases.py --only gws,sddmm --dtypes bf16 --gws-wave-cut 2>&1 | grep -v amdgpu.ids; timeout 600 python3 tools/bench_slab_cases.py --only gws,sddmm --dtypes bf16 2>&1 | grep -v amdgpu.ids; } | grep "rows 512\|F=256\|^# lib" > $O/slab_cases_rows512_single_head.txt; cat $O/slab_cases_rows512_single_head.txt ;;
    soak16)   for seed in 311 312 313 314; do
                timeout 900 python3 tools/soak_fuzz.py --iters 200 --seed $seed --ops gws,gs,gws,gs,mh,is > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; grep -c "dtype=" $O/soak_seed$seed.log; tail -1 $O/soak_seed$seed.log | cut -c1-400
              done ;;
    rows1k)   { for sh in 8,64 4,128 2,256; do timeout 600 python3 tools/bench_slab_cases.py --only mh,sddmm --dtypes bf16 --mh-shape $sh 2>&1 | grep -v amdgpu.ids; done
                echo "# the same with the matrix-core kernels off when the plans are built: the vector-ALU kernels on the plans their LDS allows (4-5 rows a group)"
                for sh in 8,64 4,128; do timeout 600 python3 tools/bench_slab_cases.py --only mh,sddmm --dtypes bf16 --mh-shape $sh --options slab_spmm_mfma=0,slab_sddmm_mfma=0 2>&1 | grep -v amdgpu.ids; done
                echo "# seg_slab_twin1k_kernel, the stand-in on a 16-row plan (a source table with an Inf / NaN in it; here: forced by the option)"
                timeout 900 python3 tools/bench_slab_cases.py --only mh --dtypes bf16 --mh-shape 8,64 --rows-per-group 16 --options slab_spmm_mfma=0 --iters 2 2>&1 | grep -v amdgpu.ids; } > $O/slab_cases_rows1k_16bit.txt 2>&1; cat $O/slab_cases_rows1k_16bit.txt ;;
    hunt1k)   export HUNT_DTYPE=bf16 HUNT_HEADS=8
              timeout 900 python3 tools/hang_hunt.py --scenario lockstep --runs 10 --slab-turn 1 --T 90 > $O/hunt1k_lockstep.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt1k_lockstep.txt
              timeout 900 python3 tools/hang_hunt.py --scenario lockstep --runs 6 --slab-turn 0 --T 90 > $O/hunt1k_lockstep_turn0.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt1k_lockstep_turn0.txt
              timeout 900 python3 tools/hang_hunt.py --scenario procs --runs 8 --T 90 > $O/hunt1k_procs.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt1k_procs.txt
              timeout 900 python3 tools/hang_hunt.py --scenario graphs --runs 8 --T 90 > $O/hunt1k_graphs.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt1k_graphs.txt
              unset HUNT_DTYPE HUNT_HEADS ;;
    sddmm16)  timeout 900 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -x -q -k "sddmm or attention or matrix_core" 2>&1 | tail -4
              for o in slab_sddmm_mfma=1 slab_sddmm_mfma=0 slab_sddmm_mfma=1,slab_probe=1; do
                timeout 600 python3 tools/bench_slab_cases.py --only mh,sddmm --dtypes bf16 --options $o 2>&1 | grep "mh_sddmm\|options"
              done > $O/slab_cases_sddmm_mfma_b128.txt 2>&1; cat $O/slab_cases_sddmm_mfma_b128.txt ;;
    compile)  timeout 1500 python3 -m pytest tests/test_compile_models.py tests/test_match_replace.py tests/test_gpu_graph_handle.py -m gpu -x -q 2>&1 | tail -25 ;;
    pmcmfma)  GEOT_HIP_LIB=product bash tools/pmc_spmm_mfma.sh $O/pmc_spmm_mfma 2>&1 | tail -40 ;;
    mfmaprod) for o in slab_spmm_mfma=1 slab_spmm_mfma=0 slab_spmm_mfma=1 slab_spmm_mfma=0; do
                GEOT_HIP_LIB=product timeout 600 python3 tools/bench_slab_cases.py --only mh --dtypes bf16 --options $o 2>&1 | grep -v amdgpu.ids
              done > $O/slab_cases_mfma_ab_product_lib.txt 2>&1; cat $O/slab_cases_mfma_ab_product_lib.txt ;;
    benchall) timeout 1500 python3 bench.py --steps 20 --warmup 5 --secondary all --detail-out $O/bench_all_detail.json > $O/bench_all.json 2> $O/bench_all.err; echo "rc=$?"; cat $O/bench_all.json; tail -2 $O/bench_all.err
              python3 - <<PY
import json
d=json.load(open("$O/bench_all_detail.json"))
for k,v in d["secondary"].items():
    keep={a:b for a,b in v.items() if a in ("kernel_ms","kernel","error","kernel_ms_without_content_guard","kernel_ms_with_graph_handle","rocsparse_best_ms","speedup_vs_rocsparse_best","us_per_call_as_dispatched","kernel_us","ms_per_step","attention_forward_ms","as_dispatched","with_graph_handle","without_content_guard","per_edge_kernels")}
    print(k, json.dumps(keep)[:700])
PY
              ;;
    mfmaR)    for R in 8 11 12 13 14 15 16; do
                timeout 600 python3 tools/bench_slab_cases.py --only mh,sddmm --dtypes bf16 --rows-per-group $R 2>&1 | grep "mh_\|options" | cut -c1-170
              done > $O/slab_cases_mfma_rows_per_group.txt 2>&1; cat $O/slab_cases_mfma_rows_per_group.txt ;;
    mfmastage) for o in slab_stage=1 slab_stage=2; do
                timeout 600 python3 tools/bench_slab_cases.py --only mh --dtypes bf16,fp32 --options $o 2>&1 | grep "mh_spmm\|options" | cut -c1-170
              done > $O/slab_cases_mh_weight_prepass.txt 2>&1; cat $O/slab_cases_mh_weight_prepass.txt ;;
    cfg5box)  bash tools/cfg5_box.sh ${CFG5_TAG:-box_$(date +%H%M%S)} 2>&1 | tail -14 ;;
    soak6)    for seed in 301 302 303 304 305 306; do
                timeout 900 python3 tools/soak_fuzz.py --iters 200 --seed $seed --ops mh,mh,mh,gws,gs,is > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -1 $O/soak_seed$seed.log | cut -c1-200
              done
              timeout 900 python3 tools/hang_hunt.py --scenario threads --runs 20 --slab-turn 1 --T 60 > $O/hunt_threads20.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt_threads20.txt
              timeout 900 python3 tools/hang_hunt.py --scenario procs --runs 10 --T 90 > $O/hunt_procs10.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt_procs10.txt
              timeout 900 python3 tools/hang_hunt.py --scenario graphs --runs 10 --T 90 > $O/hunt_graphs10.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt_graphs10.txt ;;
    kexp6)    timeout 300 ./tools/kexp6 > $O/kexp6_memset_node_under_replay.txt 2>&1; echo "rc=$?"; cat $O/kexp6_memset_node_under_replay.txt ;;
    tests)    timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_gpu.log ;;
    w8)       timeout 1500 python3 -m pytest tests/test_gpu_world8.py tests/test_gpu_multirank.py -m gpu -q --durations=12 > $O/pytest_w8.log 2>&1; echo "rc=$?"; tail -25 $O/pytest_w8.log ;;
    hunt)     timeout 1500 python3 tools/hang_hunt.py --scenario lockstep --runs 3 --slab-turn 0 --T 90 > $O/hunt_lockstep_turn0.txt 2>&1; echo "rc=$?"; tail -5 $O/hunt_lockstep_turn0.txt
              timeout 600 python3 tools/hang_hunt.py --scenario lockstep --runs 2 --slab-turn 1 --T 90 > $O/hunt_lockstep_turn1.txt 2>&1; echo "rc=$?"; tail -4 $O/hunt_lockstep_turn1.txt
              timeout 600 python3 tools/hang_hunt.py --scenario procs --runs 3 --T 90 > $O/hunt_procs.txt 2>&1; echo "rc=$?"; tail -5 $O/hunt_procs.txt
              timeout 600 python3 tools/hang_hunt.py --scenario graphs --runs 2 --T 90 > $O/hunt_graphs.txt 2>&1; echo "rc=$?"; tail -4 $O/hunt_graphs.txt
              timeout 2400 python3 tools/hang_hunt.py --scenario threads --runs 40 --slab-turn 0 --T 60 > $O/hunt_threads_turn0.txt 2>&1; echo "rc=$?"; tail -6 $O/hunt_threads_turn0.txt ;;
    alltests) timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest_gpu_all.log 2>&1; echo "rc=$?"; tail -15 $O/pytest_gpu_all.log ;;
    sweepslab) timeout 900 python3 tools/sweep_slab.py > $O/sweep_slab_mh_uniform.txt 2>&1; echo "rc=$?"; cat $O/sweep_slab_mh_uniform.txt
              timeout 900 python3 tools/sweep_slab.py --sources powerlaw > $O/sweep_slab_mh_powerlaw.txt 2>&1; echo "rc=$?"; cat $O/sweep_slab_mh_powerlaw.txt
              timeout 900 python3 tools/sweep_slab.py --case gws --quick > $O/sweep_slab_gws_uniform.txt 2>&1; echo "rc=$?"; cat $O/sweep_slab_gws_uniform.txt ;;
    slabab)   timeout 600 python3 tools/sweep_slab.py --ab slab_nt=0,1 > $O/slab_ab_nt.txt 2>&1; echo "rc=$?"; cat $O/slab_ab_nt.txt
              timeout 600 python3 tools/sweep_slab.py --ab slab_far=2,6,12,1000000 > $O/slab_ab_far.txt 2>&1; echo "rc=$?"; cat $O/slab_ab_far.txt
              timeout 600 python3 tools/sweep_slab.py --case gws --ab slab_nt=0,1 > $O/slab_ab_nt_gws.txt 2>&1; echo "rc=$?"; cat $O/slab_ab_nt_gws.txt ;;
    locality) timeout 1500 python3 tools/bench_slab_locality.py > $O/slab_vs_per_edge_by_locality.txt 2>&1; echo "rc=$?"; cat $O/slab_vs_per_edge_by_locality.txt ;;
    newtests) timeout 1500 python3 -m pytest tests/test_gpu_slab.py tests/test_reorder.py tests/test_gpu_guard.py tests/test_gpu_round3.py -m gpu -q > $O/pytest_new.log 2>&1; echo "rc=$?"; tail -12 $O/pytest_new.log ;;
    hunt2)    timeout 600 python3 tools/hang_hunt.py --scenario graphs --runs 3 --T 90 > $O/hunt_graphs2.txt 2>&1; echo "rc=$?"; tail -4 $O/hunt_graphs2.txt
              timeout 2400 python3 tools/hang_hunt.py --scenario threads --runs 200 --slab-turn 0 --T 60 > $O/hunt_threads200_turn0.txt 2>&1; echo "rc=$?"; tail -3 $O/hunt_threads200_turn0.txt ;;
    r4tests)  timeout 1500 python3 -m pytest tests/test_gpu_round4.py tests/test_reorder.py tests/test_gpu_slab.py -m gpu -q > $O/pytest_r4.log 2>&1; echo "rc=$?"; tail -12 $O/pytest_r4.log ;;
    stressho) timeout 2400 python3 tools/stress_handoff.py --matrix --calls 1000 > $O/stress_handoff_matrix.txt 2>&1; echo "rc=$?"; cat $O/stress_handoff_matrix.txt ;;
    hunt3)    timeout 600 python3 tools/hang_hunt.py --scenario graphs --runs 3 --T 90 > $O/hunt_graphs3.txt 2>&1; echo "rc=$?"; tail -4 $O/hunt_graphs3.txt; grep -h RES $O/hunt/graphs_turn1_guard1_always_00*.log ;;
    coalesced) timeout 600 python3 tools/sweep_slab.py --coalesced --ab slab_nt=0,0 > $O/slab_coalesced_mh.txt 2>&1; echo "rc=$?"; cat $O/slab_coalesced_mh.txt
              timeout 600 python3 tools/sweep_slab.py --ab slab_nt=0,0 > $O/slab_uncoalesced_mh.txt 2>&1; echo "rc=$?"; cat $O/slab_uncoalesced_mh.txt ;;
    soak4)    for seed in 61 62 63; do
                timeout 900 python3 tools/soak_fuzz.py --iters 300 --seed $seed > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -3 $O/soak_seed$seed.log
              done ;;
    trainstep) timeout 900 python3 tools/bench_train_step.py > $O/bench_train_step.txt 2>&1; echo "rc=$?"; cat $O/bench_train_step.txt ;;
    longrun)  for seed in 64 65 66 67; do
                timeout 900 python3 tools/soak_fuzz.py --iters 300 --seed $seed > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -1 $O/soak_seed$seed.log | cut -c1-200
              done
              timeout 900 python3 tools/hang_hunt.py --scenario lockstep --runs 30 --slab-turn 0 --T 90 > $O/hunt_lockstep30_turn0.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt_lockstep30_turn0.txt
              timeout 900 python3 tools/hang_hunt.py --scenario procs --runs 20 --T 90 > $O/hunt_procs20.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt_procs20.txt
              timeout 900 python3 tools/hang_hunt.py --scenario graphs --runs 20 --T 90 > $O/hunt_graphs20.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt_graphs20.txt ;;
    shardover) timeout 600 python3 tools/bench_sharded_overhead.py > $O/bench_sharded_overhead.txt 2>&1; echo "rc=$?"; cat $O/bench_sharded_overhead.txt ;;
    graphs20) timeout 900 python3 tools/hang_hunt.py --scenario graphs --runs 20 --T 90 > $O/hunt_graphs20b.txt 2>&1; echo "rc=$?"; grep -c "rc=0" $O/hunt_graphs20b.txt; tail -2 $O/hunt_graphs20b.txt
              timeout 900 python3 -m pytest tests/test_gpu_slab.py tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -q -k "graph or captur" > $O/pytest_graphs.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_graphs.log ;;
    sweep16)  timeout 900 python3 tools/sweep_slab.py --dtype bfloat16 > $O/sweep_slab_mh_bf16.txt 2>&1; echo "rc=$?"; cat $O/sweep_slab_mh_bf16.txt
              timeout 900 python3 tools/sweep_slab.py --case gws > $O/sweep_slab_gws_f32_full.txt 2>&1; echo "rc=$?"; cat $O/sweep_slab_gws_f32_full.txt ;;
    sweepw)   for c in gs64 gs128 gws256; do timeout 600 python3 tools/sweep_slab.py --case $c --ab slab_window=1,2,3 > $O/slab_window_$c.txt 2>&1; echo "rc=$?"; cat $O/slab_window_$c.txt; done
              timeout 600 python3 tools/sweep_slab.py --case gws --dtype bfloat16 --ab slab_window=1,2,3 > $O/slab_window_gws_bf16.txt 2>&1; cat $O/slab_window_gws_bf16.txt ;;
    driverbench) ( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time; echo "rc=$?"; cat $O/bench_driver_cmd.time; head -c 600 $O/bench_driver_cmd.json; echo
              timeout 900 python3 -m pytest tests/test_gpu_multirank.py -m gpu -q > $O/pytest_multirank.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_multirank.log ;;
    evidence) for seed in 68 69 70 71 72 73; do
                timeout 900 python3 tools/soak_fuzz.py --iters 300 --seed $seed > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -1 $O/soak_seed$seed.log | cut -c1-120
              done
              timeout 2400 python3 tools/hang_hunt.py --scenario threads --runs 150 --slab-turn 1 --T 60 > $O/hunt_threads150_turn1.txt 2>&1; echo "rc=$?"; tail -2 $O/hunt_threads150_turn1.txt
              timeout 1200 python3 tools/stress_handoff.py --calls 3000 > $O/stress_handoff_3000.txt 2>&1; echo "rc=$?"; tail -2 $O/stress_handoff_3000.txt ;;
    soak72)   timeout 900 python3 tools/soak_fuzz.py --iters 300 --seed 72 > $O/soak_seed72b.log 2>&1; echo "rc=$?"; tail -4 $O/soak_seed72b.log | cut -c1-900 ;;
    oddfeat)  timeout 900 python3 tools/bench_odd_feat.py > $O/bench_odd_feat.txt 2>&1; echo "rc=$?"; cat $O/bench_odd_feat.txt ;;
    kexp3)    timeout 600 ./tools/kexp3 > $O/kexp3_misaligned_vectors.txt 2>&1; echo "rc=$?"; cat $O/kexp3_misaligned_vectors.txt ;;
    absplit)  timeout 600 python3 tools/ab_libs.py geot_amd/libgeot_hip.so tools/_ab/libgeot_before_split.so > $O/ab_libs_split.txt 2>&1; echo "rc=$?"; cat $O/ab_libs_split.txt ;;
    abrag)    timeout 600 python3 tools/ab_libs.py geot_amd/libgeot_hip.so tools/_ab/libgeot_before_rag.so > $O/ab_libs_rag.txt 2>&1; echo "rc=$?"; cat $O/ab_libs_rag.txt ;;
    pmcslab)  bash tools/pmc_slab_grid.sh float32 $O/pmc_slab_float32 2:0,2:1,2:2,2:3,1:1,1:2,1:4,4:0,4:1 2>&1 | tail -14
              bash tools/pmc_slab_grid.sh bfloat16 $O/pmc_slab_bfloat16 2:1,2:2,2:3,1:2,1:4,4:0,4:1 2>&1 | tail -12
              bash tools/pmc_slab_grid.sh float32 $O/pmc_slab_float32_coalesced 2:1,2:2 --coalesced 2>&1 | tail -6
              bash tools/pmc_slab_grid.sh bfloat16 $O/pmc_slab_bfloat16_coalesced 2:1,2:2 --coalesced 2>&1 | tail -6 ;;
    slabcases) for spec in "mh float32" "mh bfloat16" "gws float32" "gws bfloat16" "gs128 float32" "gs64 float32" "gws256 float32" "mh float16"; do
                set -- $spec
                timeout 600 python3 tools/sweep_slab.py --case $1 --dtype $2 --ab slab_window=-2,-2 2>&1 | grep -v amdgpu.ids | tail -4
              done > $O/slab_cases.txt 2>&1; cat $O/slab_cases.txt ;;
    slabsweep) timeout 900 python3 tools/sweep_slab.py --case mh 2>&1 | grep -v amdgpu.ids > $O/sweep_slab_mh_f32_v2.txt; cat $O/sweep_slab_mh_f32_v2.txt
               timeout 900 python3 tools/sweep_slab.py --case mh --dtype bfloat16 2>&1 | grep -v amdgpu.ids > $O/sweep_slab_mh_bf16_v2.txt; cat $O/sweep_slab_mh_bf16_v2.txt ;;
    slabsweep2) for spec in "gws float32" "gws bfloat16" "gs128 float32" "gs64 float32" "gws256 float32" "gs128 bfloat16"; do
                 set -- $spec
                 timeout 900 python3 tools/sweep_slab.py --case $1 --dtype $2 2>&1 | grep -v amdgpu.ids > $O/sweep_slab_$1_$2_v2.txt; cat $O/sweep_slab_$1_$2_v2.txt | head -9
               done ;;
    slabsweep3) for spec in "sddmm256 float32" "sddmm128 float32" "sddmm256 bfloat16"; do
                 set -- $spec
                 timeout 900 python3 tools/sweep_slab.py --case $1 --dtype $2 2>&1 | grep -v amdgpu.ids > $O/sweep_slab_$1_$2_v2.txt; cat $O/sweep_slab_$1_$2_v2.txt | head -9
               done ;;
    benchmh)  timeout 900 python3 bench.py --full --steps 20 --warmup 5 --no-cpu-baseline --only-secondary mh_spmm_cfg4,mh_spmm_cfg4_powerlaw_src,mh_spmm_cfg4_coalesced,mh_spmm_cfg4_bf16 > $O/bench_mh.json 2> $O/bench_mh.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_mh.json'))
for k,v in d['secondary'].items(): print(k, {a:b for a,b in v.items() if a in ('kernel_ms','kernel','error','kernel_ms_without_content_guard')})" ;;
    mhrow)    timeout 1200 python3 -m pytest tests/test_gpu_slab.py -x -q -m gpu 2>&1 | tail -6
              for spec in "mh bfloat16" "mh float16" "mh float32"; do set -- $spec; timeout 600 python3 tools/sweep_slab.py --case $1 --dtype $2 --ab slab_window=-2,-2 2>&1 | grep -v amdgpu.ids | tail -2; done ;;
    slabtests) timeout 1200 python3 -m pytest tests/test_gpu_slab.py tests/test_gpu_round4.py tests/test_gpu_guard.py -x -q -m gpu 2>&1 | tail -5 ;;
    sddmmorder) for c in sddmm128 sddmm256; do timeout 600 python3 tools/sweep_slab.py --case $c --ab staged=0,1 2>&1 | grep -v amdgpu.ids | tail -4; done ;;
    pmcsqbench) bash tools/pmc_sq_bench.sh $O/pmc_sq_bench 2>&1 | tail -12 ;;
    pmcsddmm) bash tools/pmc_sddmm.sh $O/pmc_sddmm 2>&1 | tail -4 ;;
    pmcsq)    bash tools/pmc_slab_sq.sh $O/pmc_slab_sq 2>&1 | tail -30 ;;
    soaklast) for seed in 131 132 133 134 135 136; do
                timeout 900 python3 tools/soak_fuzz.py --iters 300 --seed $seed > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -1 $O/soak_seed$seed.log | cut -c1-120
              done
              for seed in 211 212 213 214; do
                timeout 900 python3 tools/soak_fuzz.py --iters 200 --seed $seed --ops mh,mh,mh,gws,gs > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -1 $O/soak_seed$seed.log | cut -c1-120
              done ;;
    soakmh)   for seed in 201 202 203 204; do
                timeout 900 python3 tools/soak_fuzz.py --iters 200 --seed $seed --ops mh,mh,gws,gs,is > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -2 $O/soak_seed$seed.log | cut -c1-300
              done ;;
    soakfinal2) for seed in 111 112 113 114 115 116 117 118 119 120 121 122; do
                timeout 900 python3 tools/soak_fuzz.py --iters 300 --seed $seed > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -1 $O/soak_seed$seed.log | cut -c1-160
              done ;;
    soakfinal) for seed in 101 102 103 104 105 106 107 108; do
                timeout 900 python3 tools/soak_fuzz.py --iters 300 --seed $seed > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -1 $O/soak_seed$seed.log | cut -c1-160
              done ;;
    soakrag)  for seed in 81 82 83; do
                timeout 900 python3 tools/soak_fuzz.py --iters 300 --seed $seed > $O/soak_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -1 $O/soak_seed$seed.log | cut -c1-160
              done ;;
    warmstate) timeout 600 python3 tools/bench_warm_state.py > $O/bench_warm_state.txt 2>&1; echo "rc=$?"; cat $O/bench_warm_state.txt ;;
    epoch) timeout 600 python3 tools/repro_epoch.py > $O/repro_epoch.txt 2>&1; echo "rc=$?"; tail -8 $O/repro_epoch.txt ;;
    renumber) timeout 1200 python3 tools/exp_renumber.py > $O/exp_renumber.txt 2>&1; echo "rc=$?"; cat $O/exp_renumber.txt ;;
    guardcost) timeout 900 python3 tools/bench_guard.py > $O/bench_content_guard.txt 2>&1; echo "rc=$?"; cat $O/bench_content_guard.txt ;;
    r3)       timeout 1200 python3 -m pytest tests/test_gpu_round3.py tests/test_plugin_registration.py -m gpu -q --durations=12 > $O/pytest_r3.log 2>&1; echo "rc=$?"; tail -25 $O/pytest_r3.log ;;
    ab)       timeout 600 python3 tools/ab_libs.py geot_amd/libgeot_hip.so tools/_ab/libgeot_r02.so > $O/ab_libs.txt 2>&1; echo "rc=$?"; cat $O/ab_libs.txt ;;
    tests2)   timeout 1500 python3 -m pytest tests/test_gpu_round2.py -m gpu -x -q > $O/pytest_round2.log 2>&1; echo "rc=$?"; tail -15 $O/pytest_round2.log ;;
    bench)    timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cat $O/bench.json; tail -3 $O/bench.err ;;
    kexp)     hipcc -O3 --offload-arch=gfx950 tools/kexp2.hip -o tools/kexp2 2>&1 | tail -3
              timeout 600 ./tools/kexp2 slab > $O/kexp2_slab.txt 2>&1; echo "rc=$?"; cat $O/kexp2_slab.txt
              timeout 300 ./tools/kexp2 slab 2449029 512 123718280 > $O/kexp2_slab_cfg3.txt 2>&1; echo "rc=$?"; cat $O/kexp2_slab_cfg3.txt
              timeout 300 ./tools/kexp2 mfma > $O/kexp2_mfma.txt 2>&1; echo "rc=$?"; cat $O/kexp2_mfma.txt ;;
    unsorted) timeout 600 python3 tools/bench_unsorted.py > $O/bench_unsorted.csv 2>&1; echo "rc=$?"; cat $O/bench_unsorted.csv ;;
    soak)     for seed in 31 32 33; do
                timeout 600 python3 tools/soak_fuzz.py --iters 150 --seed $seed --ops gws,gs,gws,gs,is > $O/soak_final_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -3 $O/soak_final_seed$seed.log
              done ;;
    profile)  bash tools/profile_round.sh > $O/profile_round.log 2>&1; echo "rc=$?"; tail -30 $O/profile_round.log ;;
    slab)     timeout 900 python3 -m pytest tests/test_gpu_slab.py -m gpu -x -q > $O/pytest_slab.log 2>&1; echo "rc=$?"; tail -15 $O/pytest_slab.log
              timeout 900 python3 tools/bench_slab.py > $O/bench_slab.txt 2>&1; echo "rc=$?"; cat $O/bench_slab.txt ;;
    small)    timeout 300 python3 tools/bench_small.py > $O/bench_small.txt 2>&1; echo "rc=$?"; cat $O/bench_small.txt ;;
    *)        echo "unknown step $s" ;;
  esac
done

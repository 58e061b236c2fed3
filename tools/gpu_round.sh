#!/bin/bash
# One GPU-box session: tests, bench, rocprof.  Every step is bounded by `timeout`; nothing reads stdin.
# usage: tools/gpu_round.sh <tag> [tests|bench|prof|pmc ...]
tag=$1; shift
steps="$*"; [ -z "$steps" ] && steps="tests bench prof"
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for s in $steps; do
  case $s in
    quick) JSLP_FORCE_PATH=resident timeout 180 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense_synthetic and resident and 200" > $out/quick.log 2>&1 < /dev/null; echo "quick rc=$?"; tail -5 $out/quick.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 < /dev/null; echo "smoke rc=$?" ;;
    xl) timeout 600 python -m pytest tests/test_xcd_local.py -m gpu -q > $out/xl.log 2>&1 < /dev/null; echo "xl rc=$?"; tail -25 $out/xl.log ;;
    xlt) timeout 600 python tools/xl_times.py $out/xl_times.md > $out/xl_times.log 2>&1 < /dev/null; echo "xlt rc=$?"; tail -20 $out/xl_times.log ;;
    pins) timeout 1500 python -m pytest tests/test_resident_pins.py -m gpu -q > $out/pins.log 2>&1 < /dev/null; echo "pins rc=$?"; tail -25 $out/pins.log ;;
    tests) timeout ${TESTS_TIMEOUT:-1800} python -m pytest tests -m gpu -q --timeout 400 --timeout-method thread ${TESTS_ARGS:-} > $out/pytest_gpu.log 2>&1 < /dev/null; echo "tests rc=$?"; tail -4 $out/pytest_gpu.log; grep -a "^FAILED\|^ERROR" $out/pytest_gpu.log | head -20 ;;
    bench) timeout 600 python bench.py --steps 3 --warmup 1 > $out/bench.log 2>&1 < /dev/null; echo "bench rc=$?"; grep '^{' $out/bench.log | cut -c1-1500 ;;
    benchfast) timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench.log 2>&1 < /dev/null; echo "bench rc=$?"; grep '^{' $out/bench.log | cut -c1-1200 ;;
    prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof.log 2>&1 < /dev/null); echo "prof rc=$?"; timeout 120 python tools/rocpd_stats.py $(ls $out/prof/*.db | head -1) $out/kernel_stats.md < /dev/null | tail -12 ;;
    pmc) for k in pivots relax; do for c in fetch write; do
           C=FETCH_SIZE; [ $c = write ] && C=WRITE_SIZE
           (cd /tmp && timeout 600 rocprofv3 --pmc $C -d $GRAFT_REPO_ROOT/$out/pmc_${k}_$c -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py $k > $GRAFT_REPO_ROOT/$out/pmc_${k}_$c.log 2>&1 < /dev/null); echo "pmc $k $c rc=$?"
         done; timeout 120 python tools/pmc_latest.py $out $k "gpurun_out/$tag (tools/gpu_round.sh pmc)" < /dev/null | head -40; done
         cp profiles/pmc_latest.json $out/pmc_latest.json ;;
    profw) for w in ${PROFW:-3a 3a_check 3b tall_4001x2001 wide_2001x4001 big_3001x3001}; do
             mkdir -p $out/profw/$w
             (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/profw/$w/kt -o k -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py $w > $GRAFT_REPO_ROOT/$out/profw/$w/kt.log 2>&1 < /dev/null)
             (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$out/profw/$w/fetch -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py $w > $GRAFT_REPO_ROOT/$out/profw/$w/fetch.log 2>&1 < /dev/null)
             (cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$out/profw/$w/write -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py $w > $GRAFT_REPO_ROOT/$out/profw/$w/write.log 2>&1 < /dev/null)
             echo "profw $w rc=$?"
           done
           timeout 120 python tools/workload_rows.py $out/profw $out/workload_rows.md $out/workload_rows.json < /dev/null ;;
    shim) for f in Monster_Problem Monster_II LargeFarmMIP Knapsack_1 Vendor_Selection; do echo "== $f" >> $out/shim_profile.log; timeout 300 node tools/shim_profile.js $f >> $out/shim_profile.log 2>&1 < /dev/null; done; echo "shim rc=$?"; cat $out/shim_profile.log | cut -c1-400 ;;
    config) timeout 900 python tools/config_times.py $out/config_times.md > $out/config_times.log 2>&1 < /dev/null; echo "config rc=$?"; cat $out/config_times.md ;;
    wgt) (JSLP_HIP_LIBRARY=build/libjslp_hip_dbg.so timeout 120 python tools/wglds_timing.py single; JSLP_HIP_LIBRARY=build/libjslp_hip_dbg.so timeout 120 python tools/wglds_timing.py batch
          timeout 120 python tools/wglds_timing.py rate; JSLP_GROUP_MAX=768 timeout 120 python tools/wglds_timing.py rate; timeout 120 python tools/wglds_timing.py single) > $out/wglds_timing.log 2>&1 < /dev/null; echo "wgt rc=$?"; grep -v "^{" $out/wglds_timing.log ;;
    ab5) # A/B of builds of the batch node kernels: AB_LIBS="build/libX.so build/libY.so" (variant libraries built on the CPU box; `shipped` is always run too).
         # Per library: outcome digests of the 2416-node batch (must be identical), compact rate x3, full read-back rate, node latency.  Round 5 used it for
         # the register diet / the prefetch / flat work items / the in-flight loads (profiles/r05_batch_kernel_register_diet_ab.md, r05_batch_kernel_experiments.md)
         (for l in shipped ${AB_LIBS:-}; do
            echo "== library: $l"; L="JSLP_HIP_LIBRARY=$l"; [ $l = shipped ] && L="JSLP_AB_NONE=1"
            env $L timeout 120 python tools/queue_check.py 2>/dev/null | head -3
            for rep in 1 2 3; do env $L WATCHED=1 timeout 120 python tools/wglds_timing.py rate | tail -1; done
            env $L timeout 120 python tools/wglds_timing.py rate | tail -1
            env $L timeout 200 python tools/node_latency.py | tail -12
          done) > $out/ab5.log 2>&1 < /dev/null; echo "ab5 rc=$?"; cat $out/ab5.log ;;
    sqp) (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $GRAFT_REPO_ROOT/$out/pmc_sq_pivots -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py pivots > $GRAFT_REPO_ROOT/$out/pmc_sq_pivots.log 2>&1 < /dev/null); echo "sqp rc=$?"
         timeout 120 python tools/pmc_sq.py $out pivots "gpurun_out/$tag (tools/gpu_round.sh sqp)" $out/headline_sq_counters.md < /dev/null
         cp profiles/pmc_latest.json $out/pmc_latest.json ;;
    sqr) (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $GRAFT_REPO_ROOT/$out/pmc_sq_relax -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py relax > $GRAFT_REPO_ROOT/$out/pmc_sq_relax.log 2>&1 < /dev/null); echo "sqr rc=$?"
         (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d $GRAFT_REPO_ROOT/$out/pmc_sq_relax2 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py relax > $GRAFT_REPO_ROOT/$out/pmc_sq_relax2.log 2>&1 < /dev/null); echo "sqr2 rc=$?"
         timeout 120 python tools/pmc_sq.py $out relax "gpurun_out/$tag (tools/gpu_round.sh sqr)" $out/node_kernel_sq_counters.md < /dev/null
         cp profiles/pmc_latest.json $out/pmc_latest.json ;;
    sq) (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $GRAFT_REPO_ROOT/$out/pmc_sq -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py relax > $GRAFT_REPO_ROOT/$out/pmc_sq.log 2>&1 < /dev/null); echo "sq rc=$?"
        (cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d $GRAFT_REPO_ROOT/$out/pmc_sq2 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py relax > $GRAFT_REPO_ROOT/$out/pmc_sq2.log 2>&1 < /dev/null); echo "sq2 rc=$?"
        timeout 120 python tools/rocpd_pmc.py $out $out/pmc_sq_summary.json < /dev/null | grep -A40 "k_node_queue" | head -120 ;;
    wgm) (JSLP_HIP_LIBRARY=build/libjslp_hip_dbg.so timeout 120 python tools/wglds_timing.py single; JSLP_HIP_LIBRARY=build/libjslp_hip_dbg.so timeout 120 python tools/wglds_timing.py batch) 2>&1 | grep -i "micro\|total\|single\|batch" > $out/wglds_micro.log; cat $out/wglds_micro.log ;;
    stress) for i in 1 2 3; do timeout 120 python tools/wglds_timing.py rate; done > $out/stress.log 2>&1 < /dev/null; echo "stress rc=$?"; grep -c relaxations $out/stress.log; tail -3 $out/stress.log ;;
    sweep5) JSLP_SWEEP_SIZES="${SWEEP_SIZES:-[[60,45],[100,75],[140,105],[200,150],[300,225],[450,340]]}" timeout 420 node tools/policy_sweep.js > $out/policy_sweep.md 2> $out/policy_sweep.err < /dev/null; echo "sweep5 rc=$?"; cat $out/policy_sweep.md ;;
    litmus) timeout 900 python -m pytest tests/test_pool_and_extras.py -m gpu -q -k "litmus or torn or host_requested or hand_over" > $out/litmus.log 2>&1 < /dev/null; echo "litmus rc=$?"; tail -15 $out/litmus.log ;;
    stream) timeout 900 python -m pytest tests/test_resident_pins.py -m gpu -q -k "beyond_the_register_file" > $out/stream.log 2>&1 < /dev/null; echo "stream rc=$?"; tail -15 $out/stream.log ;;
    bench20) timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench20.log 2>&1 < /dev/null; echo "bench20 rc=$?"; grep '^{' $out/bench20.log | cut -c1-800 ;;
    ab) (for l in base noabort; do echo "== dense LPs, build/libjslp_dev_$l.so"; JSLP_HIP_LIBRARY=build/libjslp_dev_$l.so timeout 200 python tools/dense_lp_times.py; done
         for l in base nopf; do echo "== node latency, build/libjslp_dev_$l.so"; JSLP_HIP_LIBRARY=build/libjslp_dev_$l.so timeout 200 python tools/node_latency.py; done
         echo "== shipped library"; timeout 200 python tools/dense_lp_times.py; timeout 200 python tools/node_latency.py $out/node_latency.md) > $out/ab.log 2>&1 < /dev/null; echo "ab rc=$?"; cat $out/ab.log ;;
    cpufull) timeout 400 node --max-old-space-size=8192 oracle/ref_pivot_rate.js 2000 1000000 > $out/cpu_full_run.json 2> $out/cpu_full_run.err < /dev/null; echo "cpufull rc=$?"; cat $out/cpu_full_run.json ;;
    nodelat) timeout 200 python tools/node_latency.py $out/node_latency.md > $out/node_latency.log 2>&1 < /dev/null; echo "nodelat rc=$?"; cat $out/node_latency.md ;;
    retest) timeout 1200 python -m pytest tests -m gpu -q -k "xl or xcd or virtual_shards or chaos or beyond_the_register or host_requested or rccl" > $out/retest.log 2>&1 < /dev/null; echo "retest rc=$?"; tail -12 $out/retest.log ;;
    sweep) timeout 600 node tools/mincells_sweep.js > $out/mincells_sweep.md 2> $out/mincells_sweep.err < /dev/null; echo "sweep rc=$?"; cat $out/mincells_sweep.md ;;
    dense) (timeout 300 python tools/dense_lp_times.py; JSLP_HIP_LIBRARY=build/libjslp_hip_nodefer.so timeout 300 python tools/dense_lp_times.py) > $out/dense_lp_times.log 2>&1 < /dev/null; echo "dense rc=$?"; cat $out/dense_lp_times.log ;;
    zc) (ROUNDS=3 timeout 200 python tools/batch_modes.py | tail -8
         JSLP_HIP_LIBRARY=build/libjslp_hip_dbg.so timeout 120 python tools/wglds_timing.py batch) > $out/zero_copy.log 2>&1 < /dev/null; echo "zc rc=$?"; grep -v micro $out/zero_copy.log ;;
    pmcrelax) for cow in 0 1; do for c in fetch write; do
           C=FETCH_SIZE; [ $c = write ] && C=WRITE_SIZE
           (cd /tmp && JSLP_NODE_COW=$cow timeout 600 rocprofv3 --pmc $C -d $GRAFT_REPO_ROOT/$out/pmc_relax_$c -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py relax > $GRAFT_REPO_ROOT/$out/pmc_relax_$c.log 2>&1 < /dev/null); echo "pmc relax cow=$cow $c rc=$?"
         done; timeout 120 python tools/pmc_latest.py $out relax "gpurun_out/$tag (tools/gpu_round.sh pmcrelax, JSLP_NODE_COW=$cow)" < /dev/null | grep "bytes_per_unit\|traffic_over"; rm -rf $out/pmc_relax_fetch $out/pmc_relax_write; done ;;
    qcheck) (for cfg in "JSLP_NODE_QUEUE=0 JSLP_SNAPSHOT_TRANSPOSE=0" "JSLP_NODE_COW=0" "JSLP_NODE_COW=1" "JSLP_NODE_COW=1 JSLP_SNAPSHOT_TRANSPOSE=0" "JSLP_NODE_QUEUE=1" "JSLP_NODE_QUEUE=2" "JSLP_ZERO_COPY=0" "JSLP_GROUP_MAX=100"; do echo "== $cfg"; env $cfg timeout 120 python tools/queue_check.py; done) > $out/queue_check.log 2>&1 < /dev/null; echo "qcheck rc=$?"; cat $out/queue_check.log ;;
    qprof) (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/qprof -o q -- python $GRAFT_REPO_ROOT/tools/wglds_timing.py rate > $GRAFT_REPO_ROOT/$out/qprof.log 2>&1 < /dev/null); echo "qprof rc=$?"; tail -2 $out/qprof.log
          timeout 60 python tools/rocpd_stats.py $(ls $out/qprof/*.db | head -1) 2>/dev/null | head -12 ;;
    batchtests) timeout 900 python -m pytest tests -m gpu -q -x -k "batch or pool or relax" > $out/batchtests.log 2>&1 < /dev/null; echo "batchtests rc=$?"; tail -5 $out/batchtests.log ;;
    tree) # round 6: N consecutive Monster_II trees, every C-ABI call timed; TREE_N (default 500), TREE_ENV="A=1 B=2" extra environment
          env ${TREE_ENV:-JSLP_NOOP=1} timeout 900 python tools/tree_latency.py ${TREE_N:-500} $out/tree_latency${TREE_TAG:-}.md > $out/tree_latency${TREE_TAG:-}.log 2>&1 < /dev/null; echo "tree rc=$?"; grep -v Warning $out/tree_latency${TREE_TAG:-}.log | tail -25 ;;
    devab) # round 6: A/B of development builds of the headline kernel (DEV_LIBS="build/libA.so build/libB.so"; `shipped` always runs first): config 3a / 3b rates + digests,
           # and the per-section cycle table for every library whose name ends in _dbg.so (-DJSLP_DEBUG_RESIDENT)
           (for l in shipped ${DEV_LIBS:-}; do
              echo "== library: $l"; L="JSLP_HIP_LIBRARY=$l"; [ $l = shipped ] && L="JSLP_AB_NONE=1"
              case $l in
                *_dbg.so) env $L timeout 200 python tools/resident_phase_timing.py 2000 2>&1 | grep -v "^micro" ;;
                *) env $L ${DEV_ENV:-JSLP_NOOP=1} timeout 300 python tools/dense_lp_times.py 2>&1 | tail -4 ;;
              esac
            done) > $out/devab.log 2>&1 < /dev/null; echo "devab rc=$?"; cat $out/devab.log ;;
    cpt4) (JSLP_RES_CPT=4 timeout 300 python tools/dense_lp_times.py) > $out/cpt4.log 2>&1 < /dev/null; echo "cpt4 rc=$?"; cat $out/cpt4.log ;;
    wide) timeout 900 python -m pytest tests/test_wide_goldens.py -m gpu -q > $out/wide.log 2>&1 < /dev/null; echo "wide rc=$?"; tail -15 $out/wide.log ;;
  esac
done
# keep the merge-back small (gpurun copies gpurun_out/ back only when it stays under 64 MiB -- two sessions of round 5 lost their files to a
# few 20-30 MB rocprofv3 databases): every database is summarised by the step that made it, so none travels
find $out -name '*.db' -size +2M -delete
du -sh $out | tail -1

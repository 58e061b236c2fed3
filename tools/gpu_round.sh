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
    tests) timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1 < /dev/null; echo "tests rc=$?"; tail -4 $out/pytest_gpu.log ;;
    bench) timeout 600 python bench.py --steps 3 --warmup 1 > $out/bench.log 2>&1 < /dev/null; echo "bench rc=$?"; grep '^{' $out/bench.log | cut -c1-1500 ;;
    benchfast) timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench.log 2>&1 < /dev/null; echo "bench rc=$?"; grep '^{' $out/bench.log | cut -c1-1200 ;;
    prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof.log 2>&1 < /dev/null); echo "prof rc=$?"; timeout 120 python tools/rocpd_stats.py $(ls $out/prof/*.db | head -1) $out/kernel_stats.md < /dev/null | tail -12 ;;
    pmc) for k in pivots relax; do for c in fetch write; do
           C=FETCH_SIZE; [ $c = write ] && C=WRITE_SIZE
           (cd /tmp && timeout 600 rocprofv3 --pmc $C -d $GRAFT_REPO_ROOT/$out/pmc_${k}_$c -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py $k > $GRAFT_REPO_ROOT/$out/pmc_${k}_$c.log 2>&1 < /dev/null); echo "pmc $k $c rc=$?"
         done; timeout 120 python tools/pmc_latest.py $out $k "gpurun_out/$tag (tools/gpu_round.sh pmc), round 2" < /dev/null | head -40; done
         cp profiles/pmc_latest.json $out/pmc_latest.json ;;
    shim) for f in Monster_Problem Monster_II LargeFarmMIP Knapsack_1 Vendor_Selection; do echo "== $f" >> $out/shim_profile.log; timeout 300 node tools/shim_profile.js $f >> $out/shim_profile.log 2>&1 < /dev/null; done; echo "shim rc=$?"; cat $out/shim_profile.log | cut -c1-400 ;;
    config) timeout 900 python tools/config_times.py $out/config_times.md > $out/config_times.log 2>&1 < /dev/null; echo "config rc=$?"; cat $out/config_times.md ;;
  esac
done
# keep the merge-back small: databases can be large
find $out -name '*.db' -size +40M -delete

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
    pmc) (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$out/pmc_fetch -o pf -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-relaxations --lp-size 2000 > $GRAFT_REPO_ROOT/$out/pmc_fetch.log 2>&1 < /dev/null); echo "pmc fetch rc=$?"
         (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$out/pmc_write -o pw -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-relaxations --lp-size 2000 > $GRAFT_REPO_ROOT/$out/pmc_write.log 2>&1 < /dev/null); echo "pmc write rc=$?"
         timeout 120 python tools/rocpd_pmc.py $out $out/pmc_summary.json < /dev/null | head -60 ;;
  esac
done
# keep the merge-back small: databases can be large
find $out -name '*.db' -size +40M -delete

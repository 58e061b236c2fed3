#!/bin/bash
# round 3, call R: full GPU suite, bench line, config wall times, service times
out=gpurun_out/r03_r; mkdir -p $out
export TMPDIR=/tmp
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -5 $out/pytest_gpu.log | cut -c1-300
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $out/bench.log | tail -1 > $out/bench_line.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_r/bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['avg_unit_us'], d['cycle_check_on']['value'], d['phase1_instance']['pivots_per_s'])
print(d['dropin_js']['configs'])
r=d['relaxations']; print(r['value'], r['compact_read_back']['value'], r['roofline']['frac'], r['roofline_kernel_bound_leg']['frac'], r['tree']['ms_per_solve'], r['tree']['eval_ms'], r.get('pool_virtual4',{}).get('value'))
print(d.get('speedup_vs_cpu_baseline'), r.get('speedup_vs_cpu_1_thread'))
PY
echo "== config times"; timeout 900 python tools/config_times.py $out/config_times.md > $out/config_times.log 2>&1; echo "config rc=$?"; cat $out/config_times.md

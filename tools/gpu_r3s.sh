#!/bin/bash
# round 3, call S: optional objectives in the LDS one-workgroup kernels, device-resident exchange, fuzz of the resident kernels
out=gpurun_out/r03_s; mkdir -p $out
export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests/test_wide_goldens.py tests/test_gpu_parity.py tests/test_fuzz_services.py tests/test_pool_and_extras.py tests/test_edge_cases.py tests/test_sharded_gloo.py tests/test_node_dropin.py -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -8 $out/pytest_gpu.log | cut -c1-400
echo "== fuzz"; timeout 600 python tools/fuzz_resident.py check > $out/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -4 $out/fuzz.log
echo "== batch rate"; timeout 300 python bench.py --steps 10 --warmup 3 --no-dropin > $out/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $out/bench.log | tail -1 > $out/bench_line.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_s/bench_line.json'))
r=d['relaxations']; print(d['value'], r['value'], r['compact_read_back']['value'], r['tree']['ms_per_solve'])
PY

#!/bin/bash
# The round's closing GPU session (one gpurun call): the whole GPU suite, the rocprofv3 passes bench.py's roofline reads (kernel trace, FETCH_SIZE, WRITE_SIZE, SQ
# counters), the per-workload rows, config / node-latency tables, the resident stress run, the section timing of the debug build, the fp32 sweep and the driver's
# own bench command -- everything under gpurun_out/<tag>/, from where the builder copies the summaries into profiles/.   usage: tools/final_round.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
bash tools/gpu_round.sh $tag tests 2>&1 | tail -6
bash tools/gpu_round.sh $tag prof pmc sqp sqr 2>&1 | grep -v "^\[" | tail -30
bash tools/gpu_round.sh $tag profw config nodelat 2>&1 | tail -40
PROFW="stream_5001x2001 stream_5001x3001 stream_3001x5001" bash tools/gpu_round.sh ${tag}_stream profw 2>&1 | tail -6
bash tools/stress_round.sh $out/resident_stress.txt 2>&1 | tail -1
[ -f build/libjslp_dev_final_dbg.so ] && (JSLP_HIP_LIBRARY=build/libjslp_dev_final_dbg.so timeout 200 python tools/resident_phase_timing.py 2000; python tools/resident_stamps.py) > $out/phase_timing.txt 2>&1
timeout 600 python tools/fp32_sweep.py $out/fp32_sweep.md > /dev/null 2>&1
timeout 300 python tools/pool_handoff.py 4 $out/pool_handoff.md > /dev/null 2>&1
timeout 900 python bench.py > $out/bench.log 2>&1; echo "bench rc=$?"
grep '^{' $out/bench.log > $out/bench_line.json; cut -c1-600 $out/bench_line.json
du -sh $out

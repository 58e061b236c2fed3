#!/bin/bash
# round 3, call T: where Solve(Monster LP) through the reference host spends its time (distribution, addon share, GC trace)
out=gpurun_out/r03_t; mkdir -p $out
export TMPDIR=/tmp
echo "== sharded tests"; timeout 600 python -m pytest tests/test_sharded_gloo.py -m gpu -q -x > $out/pytest_sharded.log 2>&1; echo "rc=$?"; tail -3 $out/pytest_sharded.log | cut -c1-300
echo "== shim Monster_Problem"; SHIM_DEFAULTS=1 SHIM_RUNS=30 timeout 300 node tools/shim_profile.js Monster_Problem > $out/shim_monster_lp.log 2>&1; tail -22 $out/shim_monster_lp.log | cut -c1-400
echo "== gc trace"; SHIM_DEFAULTS=1 SHIM_RUNS=30 timeout 300 node --trace-gc tools/shim_profile.js Monster_Problem > $out/shim_monster_lp_gc.log 2>&1; grep -c Scavenge $out/shim_monster_lp_gc.log; grep -c "Mark-Compact" $out/shim_monster_lp_gc.log; tail -30 $out/shim_monster_lp_gc.log | cut -c1-200
echo "== cpu reference distribution"; timeout 300 node -e '
const fs=require("fs"),path=require("path"),zlib=require("zlib");const root=process.cwd();
const solver=require(path.join(root,"oracle/_ref/src/solver.js")).default;
const g=JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(root,"tests/golden/fixtures/Monster_Problem.json.gz"))).toString());
const a=[];for(let i=0;i<42;i++){const m=JSON.parse(JSON.stringify(g.model));const t0=process.hrtime.bigint();solver.Solve(m);a.push(Number(process.hrtime.bigint()-t0)/1e6);}
console.log(a.slice(12).map(x=>x.toFixed(2)).join(" "));' 2>&1 | tail -2
echo "== shim Monster_II"; SHIM_DEFAULTS=1 SHIM_RUNS=16 timeout 300 node tools/shim_profile.js Monster_II > $out/shim_monster_ii.log 2>&1; tail -6 $out/shim_monster_ii.log | cut -c1-500

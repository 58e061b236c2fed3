// Golden vectors from random small MILPs (TEST INFRASTRUCTURE; build container only):
//   python oracle/build_ref.py && node tests/golden/gen_golden_fuzz.js | gzip -9 > tests/golden/fuzz_services.jsonl.gz
// The reference's own generators (src/test-utils/problem-generator.ts) with seeds 100..129, every model under seven
// service policies (default, incremental, enhanced variants, MIR cuts); one JSON line per run of oracle/_ref (the
// type-erased reference itself): the model, the pivot count and FNV-1a digest, the relaxation count, whether the
// reference's presolve fixed variables (such cases cannot be replayed by a host without presolve) and the result.
const path=require('path');const root=path.join(__dirname,'..','..','oracle','_ref','src');
const solver=require(path.join(root,'solver.js')).default;
const Tableau=require(path.join(root,'tableau/tableau.js')).default;
const gen=require(path.join(root,'test-utils/problem-generator.js'));
let rec=null;const P=Tableau.prototype;const op=P.pivot;
P.pivot=function(r,c){if(rec){rec.n++;rec.h=Math.imul(rec.h^r,16777619);rec.h=Math.imul(rec.h^c,16777619);}return op.call(this,r,c);};
const num=x=>Number.isFinite(x)?(Object.is(x,-0)?"-0":x):String(x);
const pols=[{},{useIncremental:true},{nodeSelection:'depth-first'},{branching:'strong'},{useMIRCuts:true},{useIncremental:true,useMIRCuts:true},{nodeSelection:'best-first',branching:'most-fractional'}];
const gens=[['generateRandomMIP',s=>({seed:s,numVariables:12+s%9,numConstraints:8+s%7,density:0.6,integerFraction:0.6})],
 ['generateKnapsack',s=>({seed:s,numVariables:15+s%20})],['generateSetCover',s=>({seed:s,numVariables:14+s%10,numConstraints:10+s%8})],
 ['generateTransportation',s=>({seed:s,numVariables:4+s%4,numConstraints:3+s%4})]];
for(let seed=100;seed<130;seed++)for(const [g,o] of gens){const model=gen[g](o(seed));
 if(!model.ints&&!model.binaries)continue;
 for(const pol of pols){const m=JSON.parse(JSON.stringify(model));m.options=Object.assign({},m.options||{},pol);
  rec={n:0,h:2166136261|0};let sol;try{sol=solver.Solve(JSON.parse(JSON.stringify(m)),undefined,true);}catch(e){rec=null;continue;}
  const r=rec;rec=null;const pre=solver.lastSolvedModel&&solver.lastSolvedModel.presolveResult;
  const res=solver.buildSimplifiedResult(sol);
  console.log(JSON.stringify({gen:g,seed,model:m,nPivots:r.n,digest:(r.h>>>0).toString(16),iter:sol._tableau.branchAndCutIterations,
   fixed:pre&&pre.fixedVariables?pre.fixedVariables.size:0,infeasPre:!!(pre&&pre.isInfeasible),keys:Object.keys(res),result:JSON.parse(JSON.stringify(res,(k,v)=>typeof v==='number'?num(v):v))}));
 }}

// Golden vectors from random small LP / MILP models with soft constraints (weight / priority), equalities, ranges and
// unrestricted variables (TEST INFRASTRUCTURE; build container only):
//   python oracle/build_ref.py && node tests/golden/gen_golden_fuzz_soft.js | gzip -9 > tests/golden/fuzz_soft.jsonl.gz
// One JSON line per run of oracle/_ref (the type-erased reference itself); every model runs in its own child process
// under a time limit because the reference does not terminate on a few of them (those seeds are left out).
const path=require('path');const root=path.join(__dirname,'..','..','oracle','_ref','src');
const solver=require(path.join(root,'solver.js')).default;
const Tableau=require(path.join(root,'tableau/tableau.js')).default;
let rec=null;const P=Tableau.prototype;const op=P.pivot;
P.pivot=function(r,c){if(rec){rec.n++;rec.h=Math.imul(rec.h^r,16777619);rec.h=Math.imul(rec.h^c,16777619);}return op.call(this,r,c);};
const num=x=>Number.isFinite(x)?(Object.is(x,-0)?"-0":x):String(x);
function rng(seed){let a=seed>>>0;return()=>{a=(a+0x6D2B79F5)>>>0;let t=a;t=Math.imul(t^(t>>>15),t|1);t^=t+Math.imul(t^(t>>>7),t|61);return((t^(t>>>14))>>>0)/4294967296;};}
const prios=["weak","medium","strong","required",1,2,3,undefined];
const child=require('child_process');
if(process.argv[2]!=='--seed'){for(let seed=1;seed<=250;seed++){const r=child.spawnSync(process.execPath,['--max-old-space-size=512',__filename,'--seed',String(seed)],{timeout:8000,encoding:'utf8',maxBuffer:1<<26});if(r.status===0&&r.stdout)process.stdout.write(r.stdout);else process.stderr.write('seed '+seed+' skipped\n');}process.exit(0);}
for(let seed=Number(process.argv[3]);seed<=Number(process.argv[3]);seed++){const r=rng(seed*7919);const ri=(a,b)=>a+Math.floor(r()*(b-a+1));
 const nv=ri(2,7),nc=ri(2,6);const model={optimize:"obj",opType:r()<0.5?"max":"min",constraints:{},variables:{}};
 for(let c=0;c<nc;c++){const k="c"+c;const b={};const kind=r();
  if(kind<0.4)b.max=ri(5,60);else if(kind<0.7)b.min=ri(1,20);else if(kind<0.85)b.equal=ri(5,40);else{b.min=ri(1,10);b.max=b.min+ri(5,40);}
  if(r()<0.45){if(r()<0.8)b.weight=ri(1,4);const p=prios[ri(0,prios.length-1)];if(p!==undefined)b.priority=p;}
  model.constraints[k]=b;}
 for(let v=0;v<nv;v++){const k="x"+v;const o={obj:ri(-5,12)};for(let c=0;c<nc;c++)if(r()<0.7)o["c"+c]=ri(-3,9);model.variables[k]=o;}
 if(r()<0.4){model.ints={};for(let v=0;v<nv;v++)if(r()<0.5)model.ints["x"+v]=1;}
 if(r()<0.3){model.unrestricted={};for(let v=0;v<nv;v++)if(r()<0.3)model.unrestricted["x"+v]=1;}
 const pols=[{},{useIncremental:true},{nodeSelection:'depth-first',branching:'most-fractional'}];
 for(const pol of (model.ints?pols:[{}])){const m=JSON.parse(JSON.stringify(model));if(Object.keys(pol).length)m.options=pol;
  rec={n:0,h:2166136261|0};let sol;try{sol=solver.Solve(JSON.parse(JSON.stringify(m)),undefined,true);}catch(e){rec=null;continue;}
  const q=rec;rec=null;const pre=solver.lastSolvedModel&&solver.lastSolvedModel.presolveResult;const res=solver.buildSimplifiedResult(sol);
  console.log(JSON.stringify({gen:'soft',seed,model:m,nPivots:q.n,digest:(q.h>>>0).toString(16),iter:sol._tableau.branchAndCutIterations===undefined?null:sol._tableau.branchAndCutIterations,
   fixed:pre&&pre.fixedVariables?pre.fixedVariables.size:0,infeasPre:!!(pre&&pre.isInfeasible),keys:Object.keys(res),result:JSON.parse(JSON.stringify(res,(k,v)=>typeof v==='number'?num(v):v))}));
 }}

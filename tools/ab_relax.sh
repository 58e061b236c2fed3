# variability of bench.py's relaxation figure across fresh processes / knobs
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['relaxations']; print(round(r['value']), r['per_call_us'])"; }
for i in 1 2 3 4 5; do run A=1; done

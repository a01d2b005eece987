run() { python bench.py $1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fp64_valu']['kernel_ms'], d['value'])"; }
echo -n "cfg2: "; run "--config 2"

run() { python bench.py $1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fp64_valu']['kernel_ms']['k_singlet'])"; }
echo -n "cfg2 class: "; run "--config 2"
echo -n "cfg2 class 20k cells: "; run "--config 2 --cells 20000"
echo -n "cfg5: "; run "--config 5"

run() { python bench.py $1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fp64_valu']['kernel_ms'], d['value'])"; }
for cw in 1 2 4; do echo -n "cfg2 classes CW=$cw: "; DMX_K1_CW=$cw run "--config 2"; done
for cw in 1 2 4; do echo -n "cfg4x2000 classes CW=$cw: "; DMX_K1_CW=$cw run "--config 4 --cells 2000"; done

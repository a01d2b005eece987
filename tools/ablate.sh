run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"; }
for cw in 1 2; do echo -n "full CW=$cw: "; DMX_K1_CW=$cw run; done

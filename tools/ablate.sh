run() { python bench.py $1 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fp64_valu']['kernel_ms']['k_singlet'])"; }
for v in 16 24 32 48; do
echo -n "V=$v wide: "; DMX_K1_WIDE_V=1 run "--config 2 --cells 4000 --samples $v"
echo -n "V=$v narrow: "; DMX_K1_WIDE_V=1000 run "--config 2 --cells 4000 --samples $v"
done

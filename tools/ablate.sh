run() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"; }
for cw in 1 2; do echo -n "full CW=$cw: "; DMX_K1_CW=$cw run; done
for ab in 6 14; do echo -n "ablate=$ab: "; DMX_LIB=$PWD/demuxlet_amd/libdmx_ab$ab.so run; done

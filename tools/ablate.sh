run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"; }
echo -n "default: "; run
echo -n "tables global: "; DMX_LIB=$PWD/demuxlet_amd/libdmx_tg.so run

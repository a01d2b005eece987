# k_singlet_can with four barcodes per wavefront (the form launch_singlet picks from 131 072 barcodes) against two, at the sizes that select it
export DMX_EXPERIMENTS=1
for B in 65536 131072 262144; do
  for cw in 2 4; do
    DMX_K1_CW=$cw python bench.py --config 2 --cells $B --only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> /tmp/e.err
    echo "B=$B CW=$cw: $(python tools/bench_brief.py)"
  done
done

bash scripts/ab.sh "--workload risingbubble --steps 20 --warmup 5 --no-parity --no-secondary" default minw3 default minw3 2>&1 | tee gpurun_out/r4z/ab_minw3_rb.txt

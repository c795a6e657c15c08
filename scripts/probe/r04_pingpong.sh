mkdir -p gpurun_out/r4z
bash scripts/ab_env.sh "--steps 20 --warmup 5 --no-parity --no-secondary" CMDG_DBG_PINGPONG=0 CMDG_DBG_PINGPONG=1 CMDG_DBG_PINGPONG=0 CMDG_DBG_PINGPONG=1 2>&1 | tee gpurun_out/r4z/ab_pingpong_hs.txt
bash scripts/ab_env.sh "--workload bomex --steps 10 --warmup 3 --no-parity --no-secondary" CMDG_DBG_PINGPONG=0 CMDG_DBG_PINGPONG=1 2>&1 | tee gpurun_out/r4z/ab_pingpong_bomex.txt
bash scripts/ab_env.sh "--nhorz 11 --steps 50 --warmup 10 --no-parity --no-secondary" CMDG_DBG_PINGPONG=0 CMDG_DBG_PINGPONG=1 2>&1 | tee gpurun_out/r4z/ab_pingpong_hs_n11.txt

mkdir -p gpurun_out/r4i
CMDG_TENDENCY_FOUR_WAVES=1 timeout -k 10 600 python -m pytest tests/test_gpu_moist.py tests/test_gpu_orders.py tests/test_gpu_roe_moist.py -x -q > gpurun_out/r4i/pytest_fourwaves.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/r4i/pytest_fourwaves.log
bash scripts/ab_env.sh "--workload bomex --steps 10 --warmup 3 --no-parity --no-secondary" CMDG_TENDENCY_FOUR_WAVES=0 CMDG_TENDENCY_FOUR_WAVES=1 CMDG_TENDENCY_FOUR_WAVES=0 CMDG_TENDENCY_FOUR_WAVES=1 2>&1 | tee gpurun_out/r4i/ab_fourwaves_bomex.txt
bash scripts/ab_env.sh "--workload bomex --bomex-ne 32 --steps 4 --warmup 2 --no-parity --no-secondary" CMDG_TENDENCY_FOUR_WAVES=0 CMDG_TENDENCY_FOUR_WAVES=1 2>&1 | tee gpurun_out/r4i/ab_fourwaves_bomex65536.txt

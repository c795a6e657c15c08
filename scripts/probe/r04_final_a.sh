# round-4 closing pass, part A: GPU suite + Held-Suarez profile set
mkdir -p gpurun_out/r4y
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4y/pytest_gpu.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/r4y/pytest_gpu.log
bash scripts/profile_bench.sh r04_heldsuarez_n30 "--steps 20 --warmup 5" 2>&1 | tail -8

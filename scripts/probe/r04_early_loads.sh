#!/bin/bash
# k_gradients / k_divgrad / k_gradlap: face tables read at kernel start (CMDG_EARLY_LOADS & 1), k_gradlap's
# volume loads before its first barrier (& 2).  build/libcmdg_e<bits>.so; default = 3; k_tendency as shipped.
mkdir -p gpurun_out/r4hg
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sphere.py tests/test_gpu_bubble.py tests/test_gpu_halo_direct.py -x -q > gpurun_out/r4hg/pytest_early.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4hg/pytest_early.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  scripts/ab.sh "--steps 20 --warmup 5" e0 e1 default 2>&1 | tee -a gpurun_out/r4hg/ab_early_hs.txt
done
scripts/ab.sh "--workload risingbubble --steps 20 --warmup 5" e0 e1 default 2>&1 | tee -a gpurun_out/r4hg/ab_early_rb.txt
scripts/ab.sh "--workload bomex --steps 10 --warmup 3" e0 e1 default 2>&1 | tee -a gpurun_out/r4hg/ab_early_bomex.txt

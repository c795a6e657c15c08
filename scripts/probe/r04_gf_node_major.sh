#!/bin/bash
# state_gradient_flux node-major for the atmosphere laws: parity, then the A/B against the reference
# layout (build/libcmdg_gfref.so = scripts/build_variant.sh gfref "-DCMDG_GF_NODE_MAJOR=0").
mkdir -p gpurun_out/r4hg
timeout -k 10 700 python -m pytest tests/test_gpu_bubble.py tests/test_gpu_moist.py tests/test_gpu_mms.py tests/test_gpu_parity.py tests/test_gpu_orders.py tests/test_gpu_halo_direct.py tests/test_gpu_filters.py tests/test_gpu_courant.py tests/test_gpu_roe_moist.py tests/test_gpu_variable_degree.py tests/test_gpu_plugins.py -x -q > gpurun_out/r4hg/pytest_gf.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r4hg/pytest_gf.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  scripts/ab.sh "--workload risingbubble --steps 20 --warmup 5" gfref default 2>&1 | tee -a gpurun_out/r4hg/ab_gf_rb.txt
  scripts/ab.sh "--workload bomex --steps 10 --warmup 3" gfref default 2>&1 | tee -a gpurun_out/r4hg/ab_gf_bomex.txt
done

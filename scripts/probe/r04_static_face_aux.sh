#!/bin/bash
# plus side of the static face columns from a per-face-node table (nfaux_static, kernels.h) against the
# gather (build/libcmdg_nofa.so = scripts/build_variant.sh nofa "-DCMDG_STATIC_FACE_AUX=0")
mkdir -p gpurun_out/r4hg
timeout -k 10 600 python -m pytest tests/test_gpu_create_contract.py tests/test_gpu_parity.py tests/test_gpu_sphere.py tests/test_gpu_bubble.py tests/test_gpu_moist.py tests/test_gpu_halo_direct.py tests/test_gpu_halo.py tests/test_gpu_mms.py tests/test_gpu_plugins.py -x -q > gpurun_out/r4hg/pytest_fa.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r4hg/pytest_fa.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  scripts/ab.sh "--steps 20 --warmup 5" nofa default 2>&1 | tee -a gpurun_out/r4hg/ab_fa_hs.txt
done
scripts/ab.sh "--workload risingbubble --steps 20 --warmup 5" nofa default nofa default 2>&1 | tee -a gpurun_out/r4hg/ab_fa_rb.txt
scripts/ab.sh "--workload bomex --steps 10 --warmup 3" nofa default nofa default 2>&1 | tee -a gpurun_out/r4hg/ab_fa_bomex.txt

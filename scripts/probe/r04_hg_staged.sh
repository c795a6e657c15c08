#!/bin/bash
# node-major Qhypervisc_grad: volume reads staged through LDS (default) against one record per lane
# straight from memory (build/libcmdg_hgdirect.so = -DCMDG_HG_STAGED=0) and the reference layout (hgref)
mkdir -p gpurun_out/r4hg
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_halo.py tests/test_gpu_halo_direct.py tests/test_gpu_sphere.py tests/test_gpu_orders.py -x -q > gpurun_out/r4hg/pytest_staged.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r4hg/pytest_staged.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  scripts/ab.sh "--steps 20 --warmup 5" hgref hgdirect default 2>&1 | tee -a gpurun_out/r4hg/ab_hs_staged.txt
done

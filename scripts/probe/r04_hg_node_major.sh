#!/bin/bash
# Qhypervisc_grad node-major inside the library: parity, then the A/B against the reference layout
# (build/libcmdg_hgref.so = scripts/build_variant.sh hgref "-DCMDG_HG_NODE_MAJOR=0").
mkdir -p gpurun_out/r4hg
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_halo.py tests/test_gpu_halo_direct.py tests/test_gpu_sphere.py tests/test_gpu_orders.py tests/test_gpu_heldsuarez_identities.py tests/test_gpu_dim2.py tests/test_gpu_plugins.py -x -q > gpurun_out/r4hg/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r4hg/pytest.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  scripts/ab.sh "--steps 20 --warmup 5" hgref default 2>&1 | tee -a gpurun_out/r4hg/ab_hs.txt
done

#!/bin/bash
# launch lists that are runs of consecutive elements passed as their first element (CMDG_CONTIG_ELEMS=0: off)
mkdir -p gpurun_out/r4hg
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sphere.py tests/test_gpu_bubble.py tests/test_gpu_moist.py tests/test_gpu_halo_direct.py tests/test_gpu_halo.py tests/test_gpu_ocean.py -x -q > gpurun_out/r4hg/pytest_contig.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4hg/pytest_contig.log
[ $rc -ne 0 ] && exit $rc
scripts/ab_env.sh "--steps 20 --warmup 5" CMDG_CONTIG_ELEMS=0 - CMDG_CONTIG_ELEMS=0 - 2>&1 | tee gpurun_out/r4hg/ab_contig_hs.txt
scripts/ab_env.sh "--workload risingbubble --steps 20 --warmup 5" CMDG_CONTIG_ELEMS=0 - CMDG_CONTIG_ELEMS=0 - 2>&1 | tee gpurun_out/r4hg/ab_contig_rb.txt
scripts/ab_env.sh "--workload ocean-split-explicit --steps 10 --warmup 3" CMDG_CONTIG_ELEMS=0 - CMDG_CONTIG_ELEMS=0 - 2>&1 | tee gpurun_out/r4hg/ab_contig_ocean.txt

#!/bin/bash
# k_tendency: dependent memory round trips issued early (CMDG_TEND_HOIST bits 1 / 2 / 4; kernels.h).
# build/libcmdg_h<bits>.so = scripts/build_variant.sh h<bits> "-DCMDG_TEND_HOIST=<bits>"; default = 7.
mkdir -p gpurun_out/r4hg
for i in 1 2; do
  scripts/ab.sh "--steps 20 --warmup 5" h0 h1 h3 h5 default 2>&1 | tee -a gpurun_out/r4hg/ab_hoist_hs.txt
done
scripts/ab.sh "--workload risingbubble --steps 20 --warmup 5" h0 h1 h3 h5 default 2>&1 | tee -a gpurun_out/r4hg/ab_hoist_rb.txt
scripts/ab.sh "--workload bomex --steps 10 --warmup 3" h0 h1 h3 h5 default 2>&1 | tee -a gpurun_out/r4hg/ab_hoist_bomex.txt

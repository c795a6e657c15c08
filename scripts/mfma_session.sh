#!/bin/bash
# GPU session for the MFMA clause: fp64 issue-rate microbenchmark, parity of the MFMA variant of
# the gradient-pass contraction, A/B of default / MFMA / no-contraction builds, MFMA counters.
set -o pipefail
mkdir -p gpurun_out
./build/fp64_peak > gpurun_out/r02_fp64_peak.jsonl || exit 1
cat gpurun_out/r02_fp64_peak.jsonl
CMDG_LIB=$PWD/build/libcmdg_mfma.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "held_suarez or tendency_matches_oracle or hyperdiffusion" > gpurun_out/r02_mfma_parity.log 2>&1; tail -3 gpurun_out/r02_mfma_parity.log
for i in 1 2; do bash scripts/ab.sh "--no-parity --no-secondary" default mfma nocontract keepq; done | tee gpurun_out/r02_ab_mfma.txt
export CMDG_LIB=$PWD/build/libcmdg_mfma.so
bash scripts/pmc.sh "--no-parity --no-secondary --steps 5" gpurun_out/r02_mfma_pmc.json "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" | grep k_

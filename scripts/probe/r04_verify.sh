# round-4 verification pass on the GPU box: memset probe, the priority
# failure with the fix and with the old fill, graph == eager on the strong rehearsal, GPU suite
mkdir -p gpurun_out/r4d
CMDG_HALO_PRIORITY=1 python scripts/probe/priority_order_diag.py > gpurun_out/r4d/diag_fixed.txt 2>&1; echo "fixed:"; grep worst gpurun_out/r4d/diag_fixed.txt
for st in torch system; do python scripts/probe/memset_null_stream_order.py $st 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4d/memset_probe_$st.txt; done
python scripts/measure_halo_exposure.py --scaling strong --size 8 --steps 100 --step-graph 2> gpurun_out/r4d/exp_graph.err | tail -1 > gpurun_out/r4d/exposure_strong_graph.json
python -c "
import json
d=json.load(open('gpurun_out/r4d/exposure_strong_graph.json')); print('graph', {k:d[k] for k in ('ms_per_step','host_enqueue_ms_per_step','graph_steps_replayed','graph_equals_eager','graph_vs_eager_max_abs_diff','graph_check_steps_replayed') if k in d})
"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4d/pytest_gpu.log 2>&1; echo pytest rc=$?; tail -4 gpurun_out/r4d/pytest_gpu.log
CMDG_TENDENCY_PAIRS=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_moist.py tests/test_gpu_orders.py tests/test_gpu_halo_direct.py tests/test_gpu_bubble.py tests/test_gpu_sphere.py -x -q > gpurun_out/r4d/pytest_pairs.log 2>&1; echo pytest pairs rc=$?; tail -4 gpurun_out/r4d/pytest_pairs.log
bash scripts/ab_env.sh "--steps 20 --warmup 5 --no-parity --no-secondary" CMDG_TENDENCY_PAIRS=0 CMDG_TENDENCY_PAIRS=1 CMDG_TENDENCY_PAIRS=0 CMDG_TENDENCY_PAIRS=1 2>&1 | tee gpurun_out/r4d/ab_pairs_hs.txt
bash scripts/ab_env.sh "--workload bomex --steps 10 --warmup 3 --no-parity --no-secondary" CMDG_TENDENCY_PAIRS=0 CMDG_TENDENCY_PAIRS=1 CMDG_TENDENCY_PAIRS=0 CMDG_TENDENCY_PAIRS=1 2>&1 | tee gpurun_out/r4d/ab_pairs_bomex.txt
bash scripts/ab_env.sh "--workload risingbubble --steps 20 --warmup 5 --no-parity --no-secondary" CMDG_TENDENCY_PAIRS=0 CMDG_TENDENCY_PAIRS=1 2>&1 | tee gpurun_out/r4d/ab_pairs_rb.txt

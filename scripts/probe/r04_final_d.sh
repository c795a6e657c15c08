# round-4 closing pass, part D: the whole GPU suite once more on the final library (log + observed maxima)
mkdir -p gpurun_out/r4y
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r4y/pytest_gpu_final.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/r4y/pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()"

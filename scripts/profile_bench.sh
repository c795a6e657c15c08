#!/bin/bash
# scripts/profile_bench.sh TAG "<bench args>": on the GPU box, for `python3 bench.py <args>`:
#   gpurun_out/TAG_bench.json          the bench line
#   gpurun_out/TAG_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary
#   gpurun_out/TAG_pmc_hbm_per_launch.json   FETCH_SIZE / WRITE_SIZE (separate --pmc passes)
# Copy what is to be judged into profiles/ afterwards.
set -o pipefail
tag="$1"; args="$2"
root=$PWD
mkdir -p gpurun_out
python3 bench.py $args > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -20 gpurun_out/${tag}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
(cd $root && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p --output-format csv -- python3 bench.py $args --no-cpu --no-events --no-parity --no-secondary > /tmp/prof_$tag.log 2>&1) || { tail -20 /tmp/prof_$tag.log; exit 1; }
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" $root/gpurun_out/${tag}_kernel_stats.csv
head -6 $root/gpurun_out/${tag}_kernel_stats.csv | cut -c1-60,200-
cd $root && bash scripts/pmc.sh "$args --no-parity --no-secondary" gpurun_out/${tag}_pmc_hbm_per_launch.json "FETCH_SIZE" "WRITE_SIZE"

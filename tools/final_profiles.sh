#!/bin/bash
# usage (GPU box): tools/final_profiles.sh <tag>   -> gpurun_out/final/: bench line, rocprofv3 kernel stats of the same
# command, per-layer conv table, PMC HBM traffic of the conv family (separate --pmc passes)
TAG=${1:-v4}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/r01_bench_$TAG.json 2> $O/bench.err || exit 1
rm -rf /tmp/ks
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> /tmp/ks.err || exit 1
cp $(ls /tmp/ks/*/*kernel_stats.csv | head -1) $O/r01_bench_kernel_stats_$TAG.csv
rm -rf /tmp/kt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 4 --warmup 2 --stages detect --knn-n 0 --no-cpu-baseline --overlap 0 --depth 1 > /tmp/kt.log 2>&1 || exit 1
python3 $R/tools/conv_table.py /tmp/kt n 64 > $O/r01_conv_layers_$TAG.txt
bash $R/tools/pmc_traffic.sh k_conv 8.125 $O/r01_traffic_conv_yolov8n_64x640x640.json -- $R/bench.py --steps 5 --warmup 3 --stages detect --knn-n 0 --no-cpu-baseline --overlap 0 --depth 1 > $O/traffic.log 2>&1
tail -1 $O/traffic.log

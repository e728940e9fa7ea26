#!/bin/bash
# usage (GPU box): tools/final_profiles.sh <round-tag>   -> gpurun_out/final/: the default bench line, the rocprofv3
# kernel summary of the same command, the per-layer conv table, and the PMC HBM traffic of the conv family at both
# bench shapes (separate --pmc passes).  Copy what should be judged into profiles/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench_pre.json 2> $O/bench.err || exit 1
echo "bench done"
rm -rf /tmp/ks
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py > $O/${TAG}_bench_under_rocprof.json 2> /tmp/ks.err || exit 1
cp $(ls /tmp/ks/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
echo "kernel stats done"
rm -rf /tmp/kt
DET="--steps 4 --warmup 2 --stages detect --knn-n 0 --no-cpu-baseline --overlap 0 --depth 1 --no-1080p"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/bench.py $DET > /tmp/kt.log 2>&1 || exit 1
python3 $R/tools/conv_table.py /tmp/kt n 64 > $O/${TAG}_conv_layers.txt
echo "conv table done"
DET5="--steps 5 --warmup 3 --stages detect --knn-n 0 --no-cpu-baseline --overlap 0 --depth 1 --no-1080p"
bash $R/tools/pmc_traffic.sh k_conv 8.125 $O/traffic_conv_yolov8n_64x640x640.json -- $R/bench.py $DET5 > $O/traffic_640.log 2>&1
tail -1 $O/traffic_640.log
bash $R/tools/pmc_traffic.sh k_conv 8.125 $O/traffic_conv_yolov8n_64x1080x1920.json -- $R/bench.py $DET5 --height 1080 --width 1920 > $O/traffic_1080.log 2>&1
tail -1 $O/traffic_1080.log
# the bench line again, now with the traffic files of THIS tree in place
mkdir -p $R/profiles && cp $O/traffic_conv_*.json $R/profiles/
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/bench2.err || exit 1
tail -c 600 $O/${TAG}_bench.json

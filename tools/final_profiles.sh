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
DET="--steps 4 --warmup 2 --stages detect --knn-n 0 --no-cpu-baseline --overlap 0 --depth 1 --no-1080p --no-cfg3 --no-cfg4 --no-cfg5"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/bench.py $DET > /tmp/kt.log 2>&1 || exit 1
python3 $R/tools/conv_table.py /tmp/kt n 64 > $O/${TAG}_conv_layers.txt
echo "conv table done"
DET5="--steps 5 --warmup 3 --stages detect --knn-n 0 --no-cpu-baseline --overlap 0 --depth 1 --no-1080p --no-cfg3 --no-cfg4 --no-cfg5"
bash $R/tools/pmc_traffic.sh k_conv 8.125 $O/traffic_conv_yolov8n_64x640x640.json -- $R/bench.py $DET5 > $O/traffic_640.log 2>&1
tail -1 $O/traffic_640.log
bash $R/tools/pmc_traffic.sh k_conv 8.125 $O/traffic_conv_yolov8n_64x1080x1920.json -- $R/bench.py $DET5 --height 1080 --width 1920 > $O/traffic_1080.log 2>&1
tail -1 $O/traffic_1080.log
# the bench line again, now with the traffic files of THIS tree in place
mkdir -p $R/profiles && cp $O/traffic_conv_*.json $R/profiles/
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/bench2.err || exit 1
tail -c 600 $O/${TAG}_bench.json

# ---- round 3: the IVF-PQ list-major scan (K10s) at 10 M x 384 (kernel statistics, PMC counters, both scan modes) and the
# encoder's time per segment against the batch size
python3 $R/tools/ivfpq_bench.py 10000000 1024 > $O/${TAG}_ivfpq_10M.json 2> $O/ivfpq.err || exit 1
tail -c 400 $O/${TAG}_ivfpq_10M.json
rm -rf /tmp/prof
IVFPQ_LISTS_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $R/tools/ivfpq_bench.py 10000000 1024 > /tmp/ivf.json 2>/tmp/ivf.err || exit 1
python3 $R/tools/kstats.py /tmp/prof 22 60 | grep -E "k_lscan|k_lrerank|k_ivfpq_scan|k_flat_l2<32|topk_merge_heads|k_ivfpq_tables|k_lbin|k_term1|k_inv_|k_lq_fill|k_q_prep|k_tau_probe|k_probe_merge" > $O/${TAG}_ivfpq_search_kernels.txt
cat $O/${TAG}_ivfpq_search_kernels.txt
export IVFPQ_LISTS_ONLY=1
bash $R/tools/pmc.sh k_lscan $O/${TAG}_pmc_ivfpq_scan.txt -- $R/tools/ivfpq_bench.py 10000000 1024 > /dev/null 2>&1
unset IVFPQ_LISTS_ONLY
head -n 8 $O/${TAG}_pmc_ivfpq_scan.txt
python3 $R/tools/embed_batch_sweep.py 2>/dev/null | tail -n 1 > $O/${TAG}_embed_batch_sweep.json
cat $O/${TAG}_embed_batch_sweep.json

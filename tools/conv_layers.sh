# usage (GPU box): tools/conv_layers.sh <tag>: per-layer conv table of the detect-only forward -> gpurun_out/r03_conv_layers_<tag>.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1
ARGS="$R/bench.py --steps 6 --warmup 3 --stages detect --knn-n 0 --no-cpu-baseline --overlap 0 --depth 1 --no-1080p --no-cfg3 --no-cfg4 --no-cfg5"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt &&
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $ARGS > /tmp/kt.log 2>&1 &&
python3 $R/tools/conv_table.py /tmp/kt n 64 > $O/r03_conv_layers_$TAG.txt; tail -1 /tmp/kt.log | cut -c1-300; grep -v "lazy\|fused" $O/r03_conv_layers_$TAG.txt | awk '{s+=$(NF-3)} END {print "sum_us", s}'; head -8 $O/r03_conv_layers_$TAG.txt

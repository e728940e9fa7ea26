# PMC counters (tools/pmc.sh passes) for the six slowest conv launches of a 64 x 640 x 640 YOLOv8n forward -> gpurun_out/r03_pmc_conv_*.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="$R/bench.py --steps 5 --warmup 3 --stages detect --knn-n 0 --no-cpu-baseline --overlap 0 --depth 1 --no-1080p --no-cfg3 --no-cfg4 --no-cfg5"
i=0
for k in "k_conv_stem_chain<0>" "k_conv3x3_chain<1, false, 1, 2>" "k_conv3x3_persist<3, 1, 3, false, false, 8>" "k_conv3x3_persist<4, 2, 1, false, true, 4>" "k_conv3x3_persist<3, 1, 2, false, false, 4>" "k_conv3x3_chain<2, false, 3, 4>"; do
  i=$((i+1))
  name=$(echo "$k" | tr -d ' <>' | tr ',' '_')
  bash $R/tools/pmc.sh "$k" $O/r03_pmc_conv_$name.txt -- $B > /dev/null 2>&1
  sed -i "1i # kernel: $k (avg over its dispatches in 64x640x640 YOLOv8n forwards, detect only, one stream)" $O/r03_pmc_conv_$name.txt
  echo "$i $k"; head -4 $O/r03_pmc_conv_$name.txt
done

#!/bin/bash
# rocprofv3 kernel statistics of the long-read bench (BASELINE.json config 5 at reduced read count), serial pass
# usage: ont_stats.sh <tag> [reads]
tag=$1; n=${2:-20000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ks_ont
BWAMEM_HIP_STREAMS=1 BWAMEM_HIP_SEED_AHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_ont -o ks -- python3 $R/bench.py --ont --read-len 10000 --reads $n --steps 1 --warmup 1 --cpu-sample 0 --h2h-calls 0 > $R/gpurun_out/ont_stats_$tag.log 2>&1
f=$(find /tmp/ks_ont -name "*kernel_stats.csv" | head -1)
grep '"k_\|void k_' $f | sed 's/(DevIndex[^"]*"/"/; s/(MemOpt[^"]*"/"/; s/(TileView[^"]*"/"/' | cut -c1-110 > $R/gpurun_out/ont_stats_$tag.csv
cat $R/gpurun_out/ont_stats_$tag.csv
tail -1 $R/gpurun_out/ont_stats_$tag.log | cut -c1-300

#!/bin/bash
# rocprofv3 kernel statistics of the bench command, as committed under profiles/ (product kernels in full, the torch
# kernels of the untimed index build / read generation cut to the top four).  usage: kernel_stats.sh <tag> [bench args...]
#   default run: four tiles and a seeding chunk in flight (the timed configuration)
#   serial run : BWAMEM_HIP_STREAMS=1 BWAMEM_HIP_SEED_AHEAD=0, what bench.py's roofline pass measures
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
filter() { python3 - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]] + [r for r in rows[1:] if "k_" in r[0][:14]]
out += [r for r in rows[1:] if "k_" not in r[0][:14]][:4]
csv.writer(open(sys.argv[2], "w"), quoting=csv.QUOTE_MINIMAL).writerows(out)
PY
}
rm -rf /tmp/ks_a /tmp/ks_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_a -o ks -- python3 $R/bench.py --cpu-sample 0 "$@" > $R/gpurun_out/ks_${tag}.log 2>&1
f=$(find /tmp/ks_a -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && filter $f $R/gpurun_out/rocprofv3_kernel_stats_bench_${tag}.csv
BWAMEM_HIP_STREAMS=1 BWAMEM_HIP_SEED_AHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_b -o ks -- python3 $R/bench.py --cpu-sample 0 --steps 1 --warmup 0 "$@" > $R/gpurun_out/ks_${tag}_serial.log 2>&1
f=$(find /tmp/ks_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && filter $f $R/gpurun_out/rocprofv3_kernel_stats_bench_${tag}_serial.csv
tail -1 $R/gpurun_out/ks_${tag}_serial.log | cut -c1-400

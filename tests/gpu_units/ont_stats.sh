#!/bin/bash
# long reads (200 000 x 10 kb, the bench's --ont batch) through the torch-free driver under rocprofv3: kernel statistics with the
# tiles in flight as in production and with one stream (isolated kernels).  usage: ont_stats.sh <tag> [reads]
tag=$1; reads=${2:-200000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python3 $R/bench.py --ont --read-len 10000 --reads $reads --steps 1 --warmup 0 --h2h-calls 0 --cpu-sample $reads --dump-only --keep-image /tmp/prof.img --dump-request /tmp/prof.req > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
hipcc -O2 -o /tmp/drive $R/tests/gpu_units/drive.cpp -L$R/gatk-bwamem-jni_amd -lbwamem_hip -Wl,-rpath,$R/gatk-bwamem-jni_amd || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_a -o kt -- /tmp/drive /tmp/prof.img /tmp/prof.req 2 > /tmp/kt_a.log 2>&1
f=$(ls /tmp/kt_a/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/ont_${tag}_kernel_stats.csv
export BWAMEM_HIP_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_b -o kt -- /tmp/drive /tmp/prof.img /tmp/prof.req 2 > /tmp/kt_b.log 2>&1
f=$(ls /tmp/kt_b/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/ont_${tag}_kernel_stats_serial.csv
tail -n 2 /tmp/kt_a.log; tail -n 2 /tmp/kt_b.log

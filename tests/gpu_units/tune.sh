#!/bin/bash
# Tuning / breakdown helper: index + 2 M-read request from bench.py once, then the torch-free driver under a few
# environment settings, and one rocprofv3 kernel-trace pass for the per-kernel split.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --reads 2000000 --steps 1 --warmup 0 --cpu-sample 2000000 --keep-image /tmp/prof.img --dump-request /tmp/prof.req > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
hipcc -O2 -o /tmp/drive $R/tests/gpu_units/drive.cpp -L$R/gatk-bwamem-jni_amd -lbwamem_hip -Wl,-rpath,$R/gatk-bwamem-jni_amd || exit 1
export BWAMEM_HIP_STREAMS=1
OUT=$R/gpurun_out/tune.log
: > $OUT
/tmp/drive /tmp/prof.img /tmp/prof.req 1 > /dev/null
t() { echo "== $*" >> $OUT; env "$@" /tmp/drive /tmp/prof.img /tmp/prof.req 2 >> $OUT 2>&1; }
t X=default
for cfg in "$@"; do t $cfg; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- /tmp/drive /tmp/prof.img /tmp/prof.req 2 > /tmp/kt.log 2>&1
f=$(ls /tmp/kt/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/tune_kernel_stats.csv
cat $OUT

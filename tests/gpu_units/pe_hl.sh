#!/bin/bash
# paired-end reads of the human-like genome through the torch-free driver: phase clocks of the pairing stage
# (BWAMEM_HIP_DEBUGK=8192) and rocprofv3 kernel statistics.  usage: pe_hl.sh <tag> [reads]
tag=$1; reads=${2:-1200000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --genome humanlike --paired --reads $reads --steps 1 --warmup 0 --h2h-calls 0 --cpu-sample $reads --dump-only --keep-image /tmp/prof.img --dump-request /tmp/prof.req > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
hipcc -O2 -o /tmp/drive $R/tests/gpu_units/drive.cpp -L$R/gatk-bwamem-jni_amd -lbwamem_hip -Wl,-rpath,$R/gatk-bwamem-jni_amd || exit 1
export BWAMEM_HIP_STREAMS=1
/tmp/drive /tmp/prof.img /tmp/prof.req 1 0x2 > /dev/null
BWAMEM_HIP_DEBUGK=8192 /tmp/drive /tmp/prof.img /tmp/prof.req 1 0x2 > $R/gpurun_out/pe_hl_${tag}_clk.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_pe -o kt -- /tmp/drive /tmp/prof.img /tmp/prof.req 2 0x2 > /tmp/kt_pe.log 2>&1
f=$(ls /tmp/kt_pe/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/pe_hl_${tag}_kernel_stats.csv
tail -3 $R/gpurun_out/pe_hl_${tag}_clk.log

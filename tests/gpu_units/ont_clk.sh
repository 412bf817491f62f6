#!/bin/bash
# phase clocks of the wave-form global alignment on a long-read batch (BWAMEM_HIP_DEBUGK bit 0x2000).  usage: ont_clk.sh [reads]
reads=${1:-50000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python3 $R/bench.py --ont --read-len 10000 --reads $reads --steps 1 --warmup 0 --h2h-calls 0 --cpu-sample $reads --dump-only --keep-image /tmp/prof.img --dump-request /tmp/prof.req > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
hipcc -O2 -o /tmp/drive $R/tests/gpu_units/drive.cpp -L$R/gatk-bwamem-jni_amd -lbwamem_hip -Wl,-rpath,$R/gatk-bwamem-jni_amd || exit 1
BWAMEM_HIP_STREAMS=1 BWAMEM_HIP_DEBUGK=8192 /tmp/drive /tmp/prof.img /tmp/prof.req 2 > $R/gpurun_out/ont_clk.log 2>&1
tail -n 6 $R/gpurun_out/ont_clk.log

#!/bin/bash
# SQ instruction counters of the hot kernels only (one --pmc pass; see pmc.sh for the full set).  usage: pmc_sq.sh [bench args for the request, e.g. --genome humanlike]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --reads 2000000 --cpu-sample 2000000 --dump-only --keep-image /tmp/prof.img --dump-request /tmp/prof.req "$@" > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
hipcc -O2 -o /tmp/drive $R/tests/gpu_units/drive.cpp -L$R/gatk-bwamem-jni_amd -lbwamem_hip -Wl,-rpath,$R/gatk-bwamem-jni_amd || exit 1
export BWAMEM_HIP_STREAMS=1
/tmp/drive /tmp/prof.img /tmp/prof.req 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_sq -o p -- /tmp/drive /tmp/prof.img /tmp/prof.req 1 > /tmp/pmc_sq.log 2>&1
f=$(ls /tmp/pmc_sq/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] && { head -1 $f > $R/gpurun_out/pmc_sq.csv; grep -E 'k_(seed|extend|final_se|sa|chain|gcigar|pe_)' $f >> $R/gpurun_out/pmc_sq.csv; wc -l $R/gpurun_out/pmc_sq.csv; }

#!/bin/bash
# SQ counters of the paired-end kernels (torch-free driver, 600 k reads = 300 k pairs of the bench's paired workload)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --reads 600000 --paired --steps 1 --warmup 0 --cpu-sample 600000 --keep-image /tmp/prof.img --dump-request /tmp/prof.req > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
hipcc -O2 -o /tmp/drive $R/tests/gpu_units/drive.cpp -L$R/gatk-bwamem-jni_amd -lbwamem_hip -Wl,-rpath,$R/gatk-bwamem-jni_amd || exit 1
export BWAMEM_HIP_STREAMS=1
/tmp/drive /tmp/prof.img /tmp/prof.req 1 0x2
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_pe -o p -- /tmp/drive /tmp/prof.img /tmp/prof.req 1 0x2 > /tmp/pmc_pe.log 2>&1
f=$(ls /tmp/pmc_pe/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] && { head -1 $f > $R/gpurun_out/pmc_pe_sq.csv; grep -E 'k_pe_' $f >> $R/gpurun_out/pmc_pe_sq.csv; wc -l $R/gpurun_out/pmc_pe_sq.csv; }
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_pe -o kt -- /tmp/drive /tmp/prof.img /tmp/prof.req 1 0x2 > /tmp/kt_pe.log 2>&1
f=$(ls /tmp/kt_pe/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/pe_kernel_stats.csv

#!/bin/bash
# experiment: k_pe_out time with parts disabled (debug bits), 1 M reads of the bench's paired workload
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --reads 1000000 --paired --steps 1 --warmup 0 --cpu-sample 1000000 --keep-image /tmp/prof.img --dump-request /tmp/prof.req > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
hipcc -O2 -o /tmp/drive $R/tests/gpu_units/drive.cpp -L$R/gatk-bwamem-jni_amd -lbwamem_hip -Wl,-rpath,$R/gatk-bwamem-jni_amd || exit 1
export BWAMEM_HIP_STREAMS=1
for dbg in 0; do
    rm -rf /tmp/kt_pe
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_pe -o kt -- /tmp/drive /tmp/prof.img /tmp/prof.req 1 $([ $dbg = 0 ] && echo 0x2 || echo 0x0) > /tmp/kt_pe.log 2>&1
  f=$(find /tmp/kt_pe -name "*kernel_stats.csv" | head -1)
  [ -z "$f" ] && { tail -5 /tmp/kt_pe.log; ls -R /tmp/kt_pe | head; continue; }
  python3 -c "
import csv,sys
for r in csv.reader(open('$f')):
    if r and ('k_pe_out' in r[0] or 'k_extend' in r[0] or 'k_gcigar' in r[0] or 'k_final' in r[0]): print('dbg=$dbg', r[0][:24], r[1], float(r[2])/1e6, 'ms')" | tee -a $R/gpurun_out/exp_pe.log
done

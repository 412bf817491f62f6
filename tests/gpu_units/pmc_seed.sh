#!/bin/bash
# HBM traffic of k_seed from the PMC counters, with the calibration the microarchitecture guide asks for: FETCH_SIZE is
# first read on a kernel with a known byte count in the same access pattern (gather_bench: dependent random 32-byte
# block reads from a 3 GiB table), then on k_seed (torch-free driver, 2 M reads of the bench workload, one seeding chunk).
# Separate --pmc passes, no tracing options.  Summaries land in gpurun_out/ (copy the ones to keep into profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --reads 2000000 --steps 1 --warmup 0 --cpu-sample 2000000 --keep-image /tmp/prof.img --dump-request /tmp/prof.req > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
hipcc -O2 -o /tmp/drive $R/tests/gpu_units/drive.cpp -L$R/gatk-bwamem-jni_amd -lbwamem_hip -Wl,-rpath,$R/gatk-bwamem-jni_amd || exit 1
hipcc -O3 --offload-arch=gfx950 -o /tmp/gather_bench $R/tests/gpu_units/gather_bench.hip 2>/dev/null || exit 1
export BWAMEM_HIP_STREAMS=1 BWAMEM_HIP_SEED_AHEAD=0
echo "== calibration: gather_bench under --pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_cal -o c -- /tmp/gather_bench 3.0 > /tmp/pmc_cal.log 2>&1
f=$(ls /tmp/pmc_cal/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] && { head -1 $f > $R/gpurun_out/pmc_cal_fetch.csv; grep k_gather $f >> $R/gpurun_out/pmc_cal_fetch.csv; wc -l $R/gpurun_out/pmc_cal_fetch.csv; }
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== k_seed under --pmc $c"
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- /tmp/drive /tmp/prof.img /tmp/prof.req 1 > /tmp/pmc_$c.log 2>&1
  tail -1 /tmp/pmc_$c.log
  f=$(ls /tmp/pmc_$c/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && { head -1 $f > $R/gpurun_out/pmc_seed_$c.csv; grep -E 'k_(seed|sa|extend|gcigar|final_se)' $f >> $R/gpurun_out/pmc_seed_$c.csv; wc -l $R/gpurun_out/pmc_seed_$c.csv; }
done

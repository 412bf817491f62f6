#!/bin/bash
# PMC counters for the hot kernels.  The index and a 2 M-read request are produced once by bench.py, then a
# torch-free driver is profiled (separate --pmc passes, as the guide prescribes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --reads 2000000 --steps 1 --warmup 0 --cpu-sample 2000000 --keep-image /tmp/prof.img --dump-request /tmp/prof.req > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
hipcc -O2 -o /tmp/drive $R/tests/gpu_units/drive.cpp -L$R/gatk-bwamem-jni_amd -lbwamem_hip -Wl,-rpath,$R/gatk-bwamem-jni_amd || exit 1
export BWAMEM_HIP_STREAMS=1
/tmp/drive /tmp/prof.img /tmp/prof.req 1
run() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- /tmp/drive /tmp/prof.img /tmp/prof.req 1 > /tmp/pmc_$name.log 2>&1
  echo "== $name rc=$?"; tail -2 /tmp/pmc_$name.log
  f=$(ls /tmp/pmc_$name/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && { head -1 $f > $R/gpurun_out/pmc_$name.csv; grep -E 'k_(seed|extend|final_se|sa|chain|gcigar)' $f >> $R/gpurun_out/pmc_$name.csv; wc -l $R/gpurun_out/pmc_$name.csv; }
}
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
run write WRITE_SIZE GRBM_GUI_ACTIVE

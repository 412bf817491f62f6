cd /tmp && export TMPDIR=/tmp
export BWAMEM_HIP_STREAMS=1
run() { # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py --reads 2000000 --steps 1 --warmup 0 --cpu-sample 0 > /tmp/pmc_$name.log 2>&1
  echo "== $name rc=$?"
  f=$(ls /tmp/pmc_$name/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && { head -1 $f; grep -E '"k_(seed|extend|final_se|sa)' $f | head -400 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$name.csv; head -1 $f > $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_header.csv; wc -l $GRAFT_REPO_ROOT/gpurun_out/pmc_$name.csv; }
}
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
run write WRITE_SIZE GRBM_GUI_ACTIVE

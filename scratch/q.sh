python -m pytest tests/test_units.py tests/test_gpu_parity.py -x -q -m gpu -k "sort_regs or chain_flt or repeat or golden or small or long" 2>&1 | tail -3
run() { label=$1; shift
  env "$@" > gpurun_out/q.json 2> gpurun_out/q.err || { echo "$label failed"; tail -2 gpurun_out/q.err; return 0; }
  python - "$label" <<PY
import json,sys
d=json.load(open("gpurun_out/q.json")); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), d["kernel_ms_isolated_pass"], d.get("parity_sample"), d.get("parity_sample_tail"))
PY
}
run hl GPU_MAX_HW_QUEUES=8 python bench.py --genome humanlike --steps 2 --warmup 2 --h2h-calls 0 --cpu-sample 100000
cp gpurun_out/q.json gpurun_out/hl_v7.json
run iid python bench.py --steps 5 --warmup 2 --h2h-calls 0 --cpu-sample 0

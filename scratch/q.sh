run() { label=$1; shift
  env "$@" > gpurun_out/q.json 2> gpurun_out/q.err || { echo "$label failed"; tail -2 gpurun_out/q.err | cut -c1-200; return 0; }
  python - "$label" <<PY
import json,sys
d=json.load(open("gpurun_out/q.json")); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), d["kernel_ms_isolated_pass"], d["host_to_host"] and d["host_to_host"]["seconds_per_call"])
PY
}
P="python bench.py --paired --reads 20000000 --steps 3 --cpu-sample 0 --h2h-calls 0"
run pe_q4 GPU_MAX_HW_QUEUES=4 $P
run pe_q8 GPU_MAX_HW_QUEUES=8 $P
run ont python bench.py --ont --read-len 10000 --reads 100000 --steps 2 --cpu-sample 2000
cp gpurun_out/q.json gpurun_out/bench_ont_v3.json

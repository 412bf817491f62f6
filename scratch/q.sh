run() { label=$1; shift
  env "$@" > gpurun_out/q.json 2> gpurun_out/q.err || { echo "$label failed"; tail -2 gpurun_out/q.err | cut -c1-200; return 0; }
  python - "$label" <<PY
import json,sys
d=json.load(open("gpurun_out/q.json")); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), d["kernel_ms_isolated_pass"])
PY
}
A="python bench.py --genome humanlike --steps 2 --warmup 2 --h2h-calls 0 --cpu-sample 0"
run hl_3x_s3 GPU_MAX_HW_QUEUES=8 BWAMEM_HIP_TILE=1179648 BWAMEM_HIP_TILE_GB=72 BWAMEM_HIP_STREAMS=3 $A
run hl_4x_s3 GPU_MAX_HW_QUEUES=8 BWAMEM_HIP_TILE=1572864 BWAMEM_HIP_TILE_GB=96 BWAMEM_HIP_STREAMS=3 $A
run hl_4x_s2 GPU_MAX_HW_QUEUES=8 BWAMEM_HIP_TILE=1572864 BWAMEM_HIP_TILE_GB=96 BWAMEM_HIP_STREAMS=2 $A
run hl_2x_s3 GPU_MAX_HW_QUEUES=8 BWAMEM_HIP_STREAMS=3 $A

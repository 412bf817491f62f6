ulimit -c 0
for dk in 8192 16384 32768; do
BWAMEM_HIP_DEBUGK=$dk timeout -k 10 300 python bench.py --paired --genome humanlike --reads 600000 --steps 1 --warmup 1 --h2h-calls 0 --cpu-sample 0 > gpurun_out/q.json 2> gpurun_out/q.err
python - $dk <<PY
import json,sys
try:
    d=json.load(open("gpurun_out/q.json")); print(sys.argv[1], round(d["ms_per_step"],1), d["kernel_ms_isolated_pass"]["final"])
except Exception as e: print("no json", e)
PY
done

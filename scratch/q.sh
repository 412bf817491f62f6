ulimit -c 0
for t in 393216 196608; do
BWAMEM_HIP_TILE=$t timeout -k 10 400 python bench.py --paired --genome humanlike --reads 4000000 --steps 2 --warmup 1 --cpu-sample 0 --h2h-calls 0 > gpurun_out/q.json 2> gpurun_out/q.err
python - $t <<PY
import json,sys
try:
    d=json.load(open("gpurun_out/q.json")); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), d["kernel_ms_isolated_pass"]["final"])
except Exception as e: print("no json", e)
PY
done

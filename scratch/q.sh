ulimit -c 0
timeout -k 10 500 python bench.py --paired --genome humanlike --reads 600000 --steps 2 --warmup 1 --h2h-calls 3 --cpu-sample 60000 --cpu-reps 1 > gpurun_out/pehl_fix.json 2> gpurun_out/pehl_fix.err
echo "rc=$?"
grep -v "^\[bench\] *[0-9.]*s \(synth\|suffix\|index\)" gpurun_out/pehl_fix.err | tail -8 | cut -c1-250
python - <<PY
import json,sys
try:
    d=json.load(open("gpurun_out/pehl_fix.json")); print(round(d["value"]), round(d["ms_per_step"],1), d["host_to_host"]["seconds_per_call"], d["cpu_baseline"]["value"], d.get("parity_sample"), d["kernel_ms_isolated_pass"], d["counters"])
except Exception as e: print("no json", e)
PY

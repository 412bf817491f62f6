#!/usr/bin/env python3
"""gpurun_out/pmc_seed_{FETCH,WRITE}_SIZE.csv + pmc_cal_fetch.csv (tests/gpu_units/pmc_seed.sh) -> the summary bench.py reads for
`roofline.traffic` (profiles/rNN_pmc_k_seed.json).  usage: pmc_seed_summary.py <n_ext of the profiled run> <out.json> [note]"""
import csv, json, sys

def total(path, kernel):
    return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"])

n_ext = int(sys.argv[1])
fetch = total("gpurun_out/pmc_seed_FETCH_SIZE.csv", "k_seed<")
write = total("gpurun_out/pmc_seed_WRITE_SIZE.csv", "k_seed<")
cal_rows = [r for r in csv.DictReader(open("gpurun_out/pmc_cal_fetch.csv")) if "k_gather<1" in r["Kernel_Name"] or "k_gather" in r["Kernel_Name"]]
cal = None
if cal_rows:
    r = cal_rows[0]
    known = 262144000
    cal = {"kernel": "gather_bench " + r["Kernel_Name"].split("(")[0] + " (dependent random 32-byte block gathers, 3 GiB table)", "known_gathers": known,
           "fetch_bytes_per_gather": float(r["Counter_Value"]) * 1024 / known}
out = {"kernel": "k_seed", "workload": sys.argv[3] if len(sys.argv) > 3 else "2,000,000 x 150bp reads of the bench workload, torch-free driver", "n_ext": n_ext,
       "fetch_size_kb": fetch, "write_size_kb": write, "fetch_bytes_per_ext": fetch * 1024 / n_ext, "write_bytes_per_ext": write * 1024 / n_ext, "calibration": cal,
       "note": "rocprofv3 --pmc FETCH_SIZE (KB x 1024), separate pass, tests/gpu_units/pmc_seed.sh; raw counter value, uncorrected: see `calibration` for what the same counter reads on a kernel with a known byte count"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out)[:300])

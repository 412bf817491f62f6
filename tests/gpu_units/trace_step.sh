#!/bin/bash
# kernel timeline of whole steps (rocprofv3 --kernel-trace, timestamps per dispatch): how the seeding chunks and the tiles of a
# 10 M-read call overlap.  Torch-free driver, three passes over the bench's batch.  usage: trace_step.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --reads 10000000 --cpu-sample 10000000 --dump-only --keep-image /tmp/prof.img --dump-request /tmp/prof.req "$@" > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
hipcc -O2 -o /tmp/drive $R/tests/gpu_units/drive.cpp -L$R/gatk-bwamem-jni_amd -lbwamem_hip -Wl,-rpath,$R/gatk-bwamem-jni_amd || exit 1
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- /tmp/drive /tmp/prof.img /tmp/prof.req 3 > /tmp/kt.log 2>&1
f=$(ls /tmp/kt/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 - "$f" "$R/gpurun_out/trace_${tag}.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["kernel", "start_ns", "end_ns", "queue", "grid", "lds"])
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    w.writerow([r["Kernel_Name"].split("(")[0][:40], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", ""), r.get("Grid_Size", ""), r.get("LDS_Block_Size", "")])
PY
tail -2 /tmp/kt.log

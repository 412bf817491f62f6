#!/bin/bash
# builds the unit harness at several optimisation levels on the GPU box and runs every variant under a timeout
cd "$(dirname "$0")"
INC=../../gatk-bwamem-jni_amd/csrc
for O in O3 O1 O0; do
  hipcc --offload-arch=gfx950 -$O -std=c++17 -ffp-contract=off -I$INC -o /tmp/sort_unit_$O sort_unit.hip 2>&1 | grep -E "error" 
  for v in 0 1 2; do
    echo "== -$O variant $v"; timeout -k 2 10 /tmp/sort_unit_$O $v; echo "rc=$?"
  done
done

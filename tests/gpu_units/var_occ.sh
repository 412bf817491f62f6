# compares builds of the library: bash tests/gpu_units/var_occ.sh <variant> ...   (gatk-bwamem-jni_amd/_var_<variant>.so is copied over libbwamem_hip.so on the box)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/var
cp $R/gatk-bwamem-jni_amd/libbwamem_hip.so /tmp/base.so
for v in base "$@"; do
  if [ $v != base ]; then cp $R/gatk-bwamem-jni_amd/_var_$v.so $R/gatk-bwamem-jni_amd/libbwamem_hip.so; fi
  python3 $R/bench.py --gpus 1 --steps 6 --warmup 3 --no-secondary --cpu-sample 20000 --cpu-reps 1 --h2h-calls 1 > $R/gpurun_out/var/head_$v.json 2> $R/gpurun_out/var/head_$v.err
  echo done $v
done
cp /tmp/base.so $R/gatk-bwamem-jni_amd/libbwamem_hip.so

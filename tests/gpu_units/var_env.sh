# the headline bench (short) under different settings of one environment knob: bash tests/gpu_units/var_env.sh NAME v1 v2 ...
set -e
R=$GRAFT_REPO_ROOT
name=$1; shift
mkdir -p $R/gpurun_out/var
for v in "$@"; do
  env $name=$v python3 $R/bench.py --gpus 1 --steps 6 --warmup 3 --no-secondary --cpu-sample 20000 --cpu-reps 1 --h2h-calls 1 > $R/gpurun_out/var/env_$v.json 2> $R/gpurun_out/var/env_$v.err
  echo done $v
done

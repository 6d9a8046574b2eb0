#!/bin/bash
# ab_bench.sh TAG "ENV_A" "ENV_B" [rounds]: bench.py alternately under two environments on ONE box (boxes of the pool differ by up to
# 10 %, so only same-box pairs compare); prints scenarios/s, ms/step, attention / chain fractions per run.
TAG=$1; A=$2; B=$3; N=${4:-2}
F="--steps 20 --warmup 5 --no-cpu-baseline --no-exact-pass --no-isolated-pass"
for i in $(seq 1 $N); do
  for v in A B; do
    if [ $v = A ]; then E=$A; else E=$B; fi
    env $E timeout -k 10 200 python bench.py $F > gpurun_out/${TAG}_${v}${i}.json 2>> gpurun_out/${TAG}.err || exit 1
    python - "$v$i [$E]" gpurun_out/${TAG}_${v}${i}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'], 1), 'scen/s', round(d['ms_per_step'], 2), 'ms/step', 'attn frac', round(d['roofline']['frac'], 3),
      'mfma frac', round(d['mfma']['frac'], 3), 'node ms', round(d['mfma']['node']['ms_per_step'], 2), 'fourier ms',
      round(d['mfma']['fourier']['ms_per_step'], 2))
PY
  done
done

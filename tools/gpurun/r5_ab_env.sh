#!/bin/bash
# round 5: A/B of an environment switch through the bench line's forward (A B A B on one box).  usage: r5_ab_env.sh VAR A_VALUE B_VALUE [tag]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
VAR=$1; A=$2; B=$3; TAG=${4:-$1}
for v in "$A" "$B" "$A" "$B"; do
  env $VAR=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-query --no-cpu-baseline --no-exclusive 2> gpurun_out/r5_ab_$TAG.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$VAR=$v', round(d['value'],1), 'img/s  sustained', round(d['sustained'].get('images_per_s',0),1), [ (k['kernel'][:24], round(k['avg_us'],1)) for k in d['kernels'][:5]])" | tee -a gpurun_out/r5_ab_$TAG.txt
done

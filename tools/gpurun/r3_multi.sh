#!/bin/bash
# round 3: the multi-rank product path (shards / workers / synthetic corpus per rank, second gather) and a 2-rank rehearsal of bench.py
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_e2e.py tests/test_pipeline.py -q -rf --durations=8 > gpurun_out/r3_multi.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r3_multi.log | cut -c1-250
HIPTS_BENCH_BACKEND=gloo HIPTS_BENCH_NO_SUSTAINED=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 1 --no-query --no-cpu-baseline --no-exclusive > gpurun_out/bench2.json 2> gpurun_out/bench2.err
echo "bench2 rc=$?"; tail -5 gpurun_out/bench2.err | cut -c1-200; python -c "
import json; d=json.load(open('gpurun_out/bench2.json')); print(d['n_gpus'], d['value'], d['output_check'])"

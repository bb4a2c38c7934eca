#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
HIPTS_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench_n2.json 2> gpurun_out/bench_n2.err; echo "rc=$?"; tail -5 gpurun_out/bench_n2.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_n2.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ['value','n_gpus','ms_per_step','scaling']}, d['config']['parallelism'])"

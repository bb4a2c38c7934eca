#!/bin/bash
# usage: gpu_run.sh [tests] [bench] [prof]
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for what in "$@"; do
case $what in
tests) timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -25 || exit 1 ;;
smoke) timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 || exit 1 ;;
bench) timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -30 gpurun_out/bench.err; cat gpurun_out/bench.json ;;
benchq) timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query --no-exclusive > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -30 gpurun_out/bench.err; cat gpurun_out/bench.json ;;
prof) cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query --no-exclusive > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err; cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/prof.err; find gpurun_out/prof -name "*kernel_stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" ;;
esac
done

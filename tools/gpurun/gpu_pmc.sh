#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-query --no-exclusive > $R/gpurun_out/pmc_$c.json 2> $R/gpurun_out/pmc_$c.err || { tail -5 $R/gpurun_out/pmc_$c.err; exit 1; }
done
cd $R && python tools/pmc_traffic.py gpurun_out/pmc gpurun_out/pmc_traffic.json
find gpurun_out/pmc -name "*.csv" | head

#!/bin/bash
# round 5 profiles: rocprofv3 kernel stats of the bench command and of the query / CCIP / EVA02 drivers, then the PMC passes
# (FETCH_SIZE, WRITE_SIZE, MFMA busy), each its own run.  Output under gpurun_out/r05/ (copied into profiles/ by hand).
TAG=${1:-b}
mkdir -p gpurun_out/r05
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
export HIPTS_BENCH_NO_SUSTAINED=1
cd /tmp && export TMPDIR=/tmp
stats() {  # name, command...
  local name=$1; shift
  rm -rf $R/gpurun_out/prof_$name
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -- "$@" > $R/gpurun_out/r05/r05_${TAG}_${name}_under_rocprof.out 2> $R/gpurun_out/prof_$name.err || { tail -5 $R/gpurun_out/prof_$name.err; return 1; }
  f=$(find $R/gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $R/gpurun_out/r05/r05_${TAG}_${name}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:        # kernel names shortened to 160 characters (torch's fill kernels run to kilobytes)
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    for r in rows:
        w.writerow([r[0][:160]] + r[1:])
print("".join(",".join(r[:5])[:150] + "\n" for r in rows[:9]))
PY
}
stats bench python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query --no-exclusive || exit 1
stats query python3 $R/tools/query_bench.py || exit 1
stats ccip python3 $R/tools/ccip_bench.py 1 || exit 1
stats eva python3 $R/tools/eva_bench.py || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc/$c
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-query --no-exclusive > $R/gpurun_out/pmc_$c.json 2> $R/gpurun_out/pmc_$c.err || { tail -5 $R/gpurun_out/pmc_$c.err; exit 1; }
done
rm -rf $R/gpurun_out/pmc_mfma
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-query --no-exclusive > $R/gpurun_out/pmc_mfma.json 2> $R/gpurun_out/pmc_mfma.err || { tail -5 $R/gpurun_out/pmc_mfma.err; exit 1; }
cd $R
python tools/pmc_traffic.py gpurun_out/pmc gpurun_out/r05/r05_${TAG}_pmc_traffic.json | head -8
python tools/pmc_mfma.py gpurun_out/pmc_mfma gpurun_out/r05/r05_${TAG}_pmc_mfma.json | head -8
unset HIPTS_BENCH_NO_SUSTAINED
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/r05_${TAG}_bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r05/r05_${TAG}_bench.json')); print('images/s', round(d['value'],1), 'sustained', round(d['sustained']['images_per_s'],1), 'frac', round(d['model_mfma_frac'],3)); q=d['query']; print('batched qps', round(q['batched_qps']), 'single', round(q['single_query_qps']), 'd2v train', round(q['d2v_train_doc_epochs_per_s']), 'cpu', round(q['d2v_train_cpu_port_doc_epochs_per_s'],1)); print(d['eva02_large'].get('images_per_s_batch10'), d['ccip'].get('images_per_s_batch20'))"

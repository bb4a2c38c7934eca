#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc2
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/sq -- python3 $R/tools/gemm_bench.py "$@" > $R/gpurun_out/pmc2_sq.log 2>&1 || { tail -5 $R/gpurun_out/pmc2_sq.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/fetch -- python3 $R/tools/gemm_bench.py "$@" > $R/gpurun_out/pmc2_f.log 2>&1 || { tail -5 $R/gpurun_out/pmc2_f.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/write -- python3 $R/tools/gemm_bench.py "$@" > $R/gpurun_out/pmc2_w.log 2>&1 || { tail -5 $R/gpurun_out/pmc2_w.log; exit 1; }
cd $R
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc2/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm' not in r['Kernel_Name']: continue
        key=(r['Kernel_Name'].split('::')[-1][:28], r['Grid_Size'])
        agg[key][r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
for key,c in agg.items():
    print(key)
    for name,vals in sorted(c.items()):
        v=[x for x,_ in vals][3:]; d=[t for _,t in vals][3:]
        if v: print('   %-28s mean %.4g   (dur %.1f us)'%(name, sum(v)/len(v), sum(d)/len(d)/1e3))
PY

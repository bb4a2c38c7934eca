#!/bin/bash
# round 4: counters of the matrix-core depthwise 7x7 alone (LDS conflicts / activity, instruction mix, texture-address and L2 activity)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/dw_pmc1 $R/gpurun_out/dw_pmc2 $R/gpurun_out/dw_pmc3
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/dw_pmc1 -- python3 $R/tools/dwconv_bench.py > $R/gpurun_out/dw_pmc1.log 2>&1 || { tail -5 $R/gpurun_out/dw_pmc1.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/dw_pmc2 -- python3 $R/tools/dwconv_bench.py > $R/gpurun_out/dw_pmc2.log 2>&1 || { tail -5 $R/gpurun_out/dw_pmc2.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/dw_pmc3 -- python3 $R/tools/dwconv_bench.py > $R/gpurun_out/dw_pmc3.log 2>&1 || { tail -5 $R/gpurun_out/dw_pmc3.log; }
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/dw_pmc1", "gpurun_out/dw_pmc2", "gpurun_out/dw_pmc3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = (row.get("Kernel_Name") or row.get("Kernel Name")).replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            name += " grid %s" % row.get("Grid_Size")
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"]); n[name].add(row.get("Dispatch_Id"))
    for k, c in sorted(agg.items()):
        if "dwconv7" not in k: continue
        L = max(len(n[k]), 1)
        print(k[:70], "launches", L)
        for cn, v in sorted(c.items()): print("   %-32s %16.0f per launch" % (cn, v / L))
PY

#!/bin/bash
# round 3: PMC passes over the attention kernel alone (tools/attn2_check.py time <variants>): where do the wave cycles go
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
V=${1:-5}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
rm -rf $R/gpurun_out/attn_pmc1 $R/gpurun_out/attn_pmc2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc1 -- python3 $R/tools/attn2_check.py time $V > $R/gpurun_out/attn_pmc1.log 2>&1 || { tail -5 $R/gpurun_out/attn_pmc1.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc2 -- python3 $R/tools/attn2_check.py time $V > $R/gpurun_out/attn_pmc2.log 2>&1 || { tail -5 $R/gpurun_out/attn_pmc2.log; }
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/attn_pmc1", "gpurun_out/attn_pmc2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = (row.get("Kernel_Name") or row.get("Kernel Name")).replace("void ", "").replace("hipts::(anonymous namespace)::", "").split("(")[0]
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"]); n[name].add(row.get("Dispatch_Id"))
    for k, c in agg.items():
        if "attn" not in k: continue
        L = max(len(n[k]), 1)
        print(k[:60], "launches", L)
        for cn, v in sorted(c.items()): print("   %-28s %14.0f per launch" % (cn, v / L))
PY

#!/bin/bash
# round 3: instruction mix and MFMA / VALU co-execution of the attention kernel alone
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
V=${1:-5}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/attn_pmc3 $R/gpurun_out/attn_pmc4
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc3 -- python3 $R/tools/attn2_check.py time $V > $R/gpurun_out/attn_pmc3.log 2>&1 || { tail -5 $R/gpurun_out/attn_pmc3.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc4 -- python3 $R/tools/attn2_check.py time $V > $R/gpurun_out/attn_pmc4.log 2>&1 || { tail -5 $R/gpurun_out/attn_pmc4.log; }
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/attn_pmc3", "gpurun_out/attn_pmc4"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = (row.get("Kernel_Name") or row.get("Kernel Name")).replace("void ", "").replace("hipts::(anonymous namespace)::", "").split("(")[0]
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"]); n[name].add(row.get("Dispatch_Id"))
    for k, c in agg.items():
        if "attn2" not in k: continue
        L = max(len(n[k]), 1)
        print(k[:60], "launches", L)
        for cn, v in sorted(c.items()): print("   %-28s %14.0f per launch" % (cn, v / L))
PY

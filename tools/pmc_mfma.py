#!/usr/bin/env python3
"""Parse a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (with --kernel-trace for the durations) into
per-kernel matrix-pipe utilisation and effective clock, as MI355X_MICROARCH.md prescribes: SQ_VALU_MFMA_BUSY_CYCLES counts
shader cycles (16 per v_mfma_f32_16x16x32_bf16) summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs,
so elapsed cycles = GRBM_GUI_ACTIVE / 8 and effective clock = that / kernel wall time (reads high on dispatches < 0.3 ms)."""
import csv, glob, json, os, sys, collections
root, out = sys.argv[1], sys.argv[2]
SIMDS = 256 * 4
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
dur = collections.defaultdict(float)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        name = (row.get("Kernel_Name") or row.get("Kernel Name")).replace("void ", "").replace("hipts::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row.get("Dispatch_Id"), name)
        if key not in seen:
            seen.add(key)
            n[name] += 1
            if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                dur[name] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
res = {}
for k, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui <= 0:
        continue
    avg_ms = dur[k] / max(n[k], 1) / 1e6
    # GRBM_GUI_ACTIVE / 8 / wall time reads HIGH on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back): the quotient is
    # not a clock there (round-4 files showed 2.2-3.7 GHz on a 2.4 GHz part), and mfma_busy_frac, which divides by the same cycle count,
    # is then a LOWER bound.  The trustworthy clock of a GEMM loop is the in-kernel one (bench.py: roofline.clock.main_loop_ghz).
    short = avg_ms < 0.3
    res[k] = {"launches": n[k], "mfma_busy_frac": busy / (gui * SIMDS), "mfma_busy_frac_is_lower_bound": short,
              "elapsed_cycles_per_launch": gui / max(n[k], 1), "avg_dispatch_ms": avg_ms,
              "effective_clock_ghz": None if short or dur[k] <= 0 else gui / dur[k]}
json.dump({"method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                     "--no-query --no-exclusive`; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the share of "
                     "SIMD-cycles AT THE CLOCK THE KERNEL RAN AT in which the matrix pipe was busy (two-stream forward: a launch shares the chip)",
           "note": "effective_clock_ghz is null, and mfma_busy_frac a lower bound (mfma_busy_frac_is_lower_bound), for kernels whose average "
                   "dispatch is shorter than 0.3 ms: GRBM_GUI_ACTIVE / 8 over-counts elapsed cycles there",
           "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_frac"])[:12]:
    print("%-44s n=%4d  MFMA busy %s%5.1f %%  clock %s GHz" % (k[:44], v["launches"], ">= " if v["mfma_busy_frac_is_lower_bound"] else "", 100 * v["mfma_busy_frac"],
                                                             ("%.2f" % v["effective_clock_ghz"]) if v["effective_clock_ghz"] else "-"))

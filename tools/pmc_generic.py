#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel name: python tools/pmc_generic.py <dir of a --pmc --kernel-trace run> [top N].
Prints per kernel: launches, average dispatch time, every counter's per-launch sum and its value divided by GRBM_GUI_ACTIVE / 8 (elapsed
cycles: a busy share when the counter counts cycles of ONE unit; divide by the unit count yourself, e.g. 256 for TA_TA_BUSY_sum)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 10
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int); dur = collections.defaultdict(float)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        name = (row.get("Kernel_Name") or row.get("Kernel Name")).replace("void ", "").replace("hipts::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row.get("Dispatch_Id"), name)
        if key not in seen:
            seen.add(key); n[name] += 1
            if row.get("Start_Timestamp") and row.get("End_Timestamp"): dur[name] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
for name in sorted(agg, key=lambda k: -dur[k])[:top]:
    c = agg[name]; gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    print("%-70s n=%4d avg %8.1f us" % (name, n[name], dur[name] / max(n[name], 1) / 1e3))
    for k, v in sorted(c.items()):
        print("    %-40s %14.0f per launch  %8.3f of elapsed cycles" % (k, v / max(n[name], 1), v / gui if gui else 0.0))

#!/usr/bin/env python3
"""Development aid: per-CU timeline of the dw GEMM from the trace file written by hiptsdbg_gemm_time."""
import sys, collections
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
t0 = min(r[3] for r in rows)
cus = collections.defaultdict(list)
for b, hw, xcc, ts, tl, te in rows:
    cus[(xcc & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)].append((ts - t0, tl - t0, te - t0, b, (hw >> 4) & 3, hw & 0xf))
print("CUs seen:", len(cus), "WGs:", len(rows))
for k in sorted(cus)[:3]:
    print("CU", k)
    for w in sorted(cus[k]):
        print("   start %6d  loop_end %6d  end %6d   (x10ns)  wg %4d simd %d waveslot %d" % w)
# aggregate: mean loop time, mean epilogue time, overlap fraction
loop = [r[4] - r[3] for r in rows]; epi = [r[5] - r[4] for r in rows]
print("mean loop %.1f  mean epilogue %.1f (x10 ns)" % (sum(loop) / len(loop), sum(epi) / len(epi)))

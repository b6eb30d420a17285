#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace run (dev tool):  python tools/trace_top.py <output dir> [n]"""
import collections, csv, glob, os, sys
f = max(glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    n = n[:n.index("(")] if "(" in n else n
    agg[n[:100]][0] += 1; agg[n[:100]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"{f}: total kernel time {tot / 1e6:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k:<102s} calls {v[0]:6d}  total {v[1] / 1e3:10.1f} us  avg {v[1] / 1e3 / v[0]:9.1f} us  {100.0 * v[1] / tot:5.1f} %")

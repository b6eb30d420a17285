#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv -> a small markdown table per kernel (mean per dispatch).
  python tools/summarize_pmc.py <out.md> <title> <dir> [<dir> ...]      (one directory per --pmc pass)"""
import collections, csv, glob, sys
out, title, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mi355q" not in k:
                continue
            name = k[k.index("mi355q::") + 8:].split("(")[0]
            vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for v in vals.values() for c in v})
with open(out, "w") as fo:
    fo.write(f"# {title}\n\nmean per dispatch, summed over the chip as rocprofv3 reports them (GRBM_GUI_ACTIVE is the sum over the 8 XCDs); MfmaUtil% = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) x 1024 SIMDs) x 100 (rocprofv3's derived-metric formula, with max-over-XCD replaced by the mean)\n\n")
    fo.write("| kernel | dispatches | " + " | ".join(counters) + " | MfmaUtil% |\n|---|---|" + "---|" * (len(counters) + 1) + "\n")
    for k, v in sorted(vals.items()):
        n = max(len(x) for x in v.values())
        m = {c: (sum(v[c]) / len(v[c]) if v[c] else float("nan")) for c in counters}
        util = 100.0 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (m.get("GRBM_GUI_ACTIVE", float("nan")) / 8 * 1024) if "GRBM_GUI_ACTIVE" in m else float("nan")
        fo.write(f"| {k} | {n} | " + " | ".join(f"{m[c]:.4g}" for c in counters) + f" | {util:.1f} |\n")
print(open(out).read())

#!/usr/bin/env python3
"""Turn rocprofv3 output directories (gpurun_out/...) into the small committed summaries under profiles/.

  python tools/summarize_profile.py <round-tag> <kernel-trace-dir> [<pmc-dir>]

kernel-trace-dir: output of `rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 bench.py ...`
pmc-dir         : output of `rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -- python3 bench.py --no-graph ...`
FETCH_SIZE is in KiB and, on gfx950, reports exactly 1/2 of the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM): the tables below show it doubled."""
import collections
import csv
import glob
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]


def short(name):
    if "k_gemv_fast" in name:
        return "k_gemv_fast" + name[name.index("<"):name.index(">") + 1]
    if "k_plan" in name:
        return "k_plan" + name[name.index("<"):name.index(">") + 1]
    return name.split("(")[0][:70]


def main():
    tag, kdir = sys.argv[1], sys.argv[2]
    pdir = sys.argv[3] if len(sys.argv) > 3 else None
    out = ROOT / "profiles"
    out.mkdir(exist_ok=True)
    import os
    newest = lambda pat: max(glob.glob(pat, recursive=True), key=os.path.getmtime)       # (gpurun merges every pass into the same local directory)
    stats = newest(f"{kdir}/**/*kernel_stats.csv")
    trace = newest(f"{kdir}/**/*kernel_trace.csv")
    rows = list(csv.DictReader(open(stats)))
    with open(out / f"{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:12]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    # per (kernel, grid) class from the trace: lets one match a launch class (qkv, wo, gate|up, down, output) to its duration
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "k_gemv_fast" in r["Kernel_Name"] or "k_plan" in r["Kernel_Name"]:
            agg[(short(r["Kernel_Name"]), r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out / f"{tag}_gemv_launch_classes.md", "w") as f:
        f.write(f"# {tag}: k_plan / k_gemv_fast launches by (instantiation, grid) -- rocprofv3 --kernel-trace, kernel-only durations\n\n")
        f.write("k_gemv_fast template args: <family (0=Q8_K acts, 1=Q8_0 acts), weight ggml type id, N columns, ring depth D, round-even, multi-matrix>; "
                "k_plan<bit set of ggml weight type ids> (20480 = Q4_K|Q6_K) is one whole token per launch\n\n")
        f.write("| kernel | grid threads | LDS B | VGPR | SGPR | calls | min us | median us | mean us | max us |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            v.sort()
            f.write(f"| {k[0]} | {k[1]} | {k[2]} | {k[3]} | {k[4]} | {len(v)} | {v[0]/1e3:.2f} | {v[len(v)//2]/1e3:.2f} | {sum(v)/len(v)/1e3:.2f} | {v[-1]/1e3:.2f} |\n")
    if pdir:
        cc = newest(f"{pdir}/**/*counter_collection.csv")
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(cc)):
            if ("k_gemv_fast" in r["Kernel_Name"] or "k_plan" in r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE":
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        with open(out / f"{tag}_pmc_fetch_size.md", "w") as f:
            f.write(f"# {tag}: HBM traffic of k_plan / k_gemv_fast from rocprofv3 --pmc FETCH_SIZE (own pass)\n\n")
            f.write("FETCH_SIZE [KiB] x 2 (gfx950 correction for 16 B/lane streaming reads) per launch; distinct values = distinct launch shapes.\n\n")
            f.write("| kernel | launches | distinct corrected MB per launch (count) |\n|---|---|---|\n")
            for k, v in agg.items():
                c = collections.Counter(round(x * 2 * 1024 / 1e6, 1) for x in v)
                f.write(f"| {k} | {len(v)} | " + ", ".join(f"{mb} ({n})" for mb, n in sorted(c.items())[:14]) + " |\n")
            import json, statistics
            traffic = {k: int(statistics.median(v) * 2 * 1024) for k, v in agg.items() if k.startswith("k_plan")}
            from bench import kernel_source_sha16           # the measurement is quoted by bench.py only while the kernel sources are unchanged
            (out / f"{tag}_traffic.json").write_text(json.dumps({"unit": "bytes per launch", "source": f"profiles/{tag}_pmc_fetch_size.md",
                "kernel_source_sha16": kernel_source_sha16(),
                "method": "rocprofv3 --pmc FETCH_SIZE (own pass of `bench.py --steps 8`), median over launches, KiB x 1024 x 2 (gfx950 wide-read correction)",
                "kernels": traffic}, indent=1) + "\n")
            f.write("\nAlgorithmic MB per launch of the Llama-3-8B Q4_K_M token (bench.py): wq+wk(+wv) 11.8/14.2, wv(q6_K) 3.4, wo 9.4, "
                    "gate|up 66.1, down 33.0 (q4_K) / 48.2 (q6_K), output 430.9; the whole token (one k_plan launch) 4616.3 of weights "
                    "+ 0.131 per cached position of K and V (32 layers x 2 x 1024 f16 x 2 B).\n")
    print("wrote", sorted(p.name for p in out.glob(f"{tag}_*")))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Kernel timeline of ONE resident decoder layer through the plugin (dev tool): reads the rocprofv3 --kernel-trace CSV of
`layer_parity 1 MI355_0 8b <steps> <iters>` and writes profiles/<round>_plugin_layer.md (the last whole graph_compute).
usage: summarize_layer_trace.py <round> <trace_dir> [label]"""
import csv, glob, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
rnd, d = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else "default (fusions on, launch graphs off)"
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a graph_compute ends with a host synchronize: the next one starts after the longest gaps; take the last complete group
gaps = [(int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]), i) for i in range(1, len(rows))]
cuts = sorted(i for g, i in gaps if g > 15000)          # > 15 us between kernels = a new graph_compute
if len(cuts) < 2:
    sys.exit("could not find two graph_compute boundaries in the trace")
a, b = cuts[-2], cuts[-1]
grp = rows[a:b]
t0 = int(grp[0]["Start_Timestamp"])
out = [f"# {rnd}: kernel timeline of one Llama-3-8B decoder layer (n_tokens = 1, n_kv = 512) through the plugin -- {label}", "",
       "`rocprofv3 --kernel-trace -- oracle/_ref/avx2/layer_parity 1 MI355_0 8b 2 50` with `MI355_NO_GRAPHS=1` (so that every kernel is a separate",
       "dispatch in the trace); the last whole `ggml_backend_graph_compute`.  `gap` = idle time on the device before the kernel.", "",
       "| start us | duration us | gap us | kernel |", "|---:|---:|---:|---|"]
prev = None
for r in grp:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("mi355q::", "").replace("void ", "")
    name = name.split("(")[0]
    out.append(f"| {(s - t0) / 1000:.1f} | {(e - s) / 1000:.1f} | {((s - prev) / 1000) if prev else 0:.1f} | `{name}` |")
    prev = e
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp) / 1000
span = (int(grp[-1]["End_Timestamp"]) - t0) / 1000
host_gap = (int(rows[b]["Start_Timestamp"]) - int(grp[-1]["End_Timestamp"])) / 1000
out += ["", f"{len(grp)} dispatches, {busy:.1f} us of kernel time in a {span:.1f} us span; {host_gap:.1f} us pass between the last kernel and the first one of the next",
        "graph_compute (stream synchronize + issuing the next graph on the host; paid once per token, not per layer, in a whole model)."]
p = ROOT / "profiles" / f"{rnd}_plugin_layer.md"
p.write_text("\n".join(out) + "\n")
print(f"wrote {p}: {len(grp)} dispatches, {busy:.1f} us busy, {span:.1f} us span, host gap {host_gap:.1f} us")

#!/usr/bin/env python3
"""In-kernel timeline of the decode plan (diagnostic build libmi355q_dbg.so; dev tool, GPU box only).
Per stage and workgroup, waves 0 and 15 stamp: 0 stage entry, 1 prime issued, 2 barrier passed, 3 activations in LDS,
4 rows done, 5 y stores acknowledged.   Usage: MI355Q_LIB=.../libmi355q_dbg.so python tools/planstamps.py [--layers 2]"""
import argparse, ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import numpy as np, torch
import ggml_mi355 as g
from ggml_mi355 import workloads as wl
import bench as B

ap = argparse.ArgumentParser(); ap.add_argument("--layers", type=int, default=3); ap.add_argument("--no-depends", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
L = g.lib()
specs = [s for s in wl.llama_matmuls(dict(wl.LLAMA3_8B), "Q4_K_M") if 0 <= s.layer < a.layers]
stage = B.Stage(torch, g, specs, True, dev)
plan = g.Plan([(ws, x, ys, (i > 0 and not a.no_depends)) for i, (ws, x, ys, _) in enumerate(stage.groups)])
n = plan.launch_stages
grid = 256
buf = torch.zeros(n * grid * 2 * 8, dtype=torch.int64, device=dev)
L.mi355q_debug_set_plan_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
assert L.mi355q_debug_set_plan_stamps(buf.data_ptr(), n) == 0
plan.run(); torch.cuda.synchronize()
assert plan.status() == 0
s = buf.cpu().numpy().reshape(n, grid, 2, 8).astype(np.float64)
t0 = s[0, :, :, 0][s[0, :, :, 0] > 0].min()
s = np.where(s > 0, (s - t0) / 100.0, np.nan)          # us
labs = ["entry", "primed", "barrier ok", "acts in LDS", "rows done", "y acked"]
print(f"{n} launch stages; times in us since the first stage entry; per stamp: min / median / max over workgroups (wave 0 | wave 15)")
prev_end = 0.0
for st in range(n):
    grp_bytes = None
    line = [f"stage {st:3d}"]
    for i, lab in enumerate(labs):
        c0, c1 = s[st, :, 0, i], s[st, :, 1, i]
        if np.all(np.isnan(c0)):
            continue
        line.append(f"{lab} {np.nanmin(c0):7.2f}/{np.nanmedian(c0):7.2f}/{np.nanmax(c0):7.2f} | {np.nanmedian(c1):7.2f}")
    print("  ".join(line))

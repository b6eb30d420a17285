#!/usr/bin/env python3
"""In-kernel timeline of the decode plan (diagnostic build libmi355q_dbg.so; dev tool, GPU box only).
Per stage and workgroup, consumer waves 0 and 14 stamp  GEMV: 0 entry, 2 operands gathered (producers polled), 6 all waves gathered (barrier passed), 7 this wave's spans quantized, 3 activations quantized in LDS (barrier passed), 1 this wave's first row has landed, 4 rows done; 5 = the LOADER (wave 15) has requested the stage's last page;
ATTN: 0 entry, 1 q/k/v gathered, 2 roped + stored, 3 scores, 4 softmax, 5 P.V published;  COMBINE: 0 entry, 2 merged.
The register-ring engine (default) stamps waves 0 and 15 with the same numbering (no loader stamp, no cycle sums); MI355Q_PLAN_ENGINE=ring selects the other.
Usage: MI355Q_LIB=.../libmi355q_dbg.so [MI355Q_PLAN_ENGINE=ring] python tools/planstamps.py [--layers 2] [--pos 100] [--n-ctx 128]"""
import argparse, ctypes, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import numpy as np, torch
import ggml_mi355 as g
from ggml_mi355 import workloads as wl
import bench as B

ap = argparse.ArgumentParser(); ap.add_argument("--layers", type=int, default=2); ap.add_argument("--pos", type=int, default=100)
ap.add_argument("--n-ctx", type=int, default=128); ap.add_argument("--csv", default=None, help="write the per-stage medians of every stamp (wave 0; us) to this file")
a = ap.parse_args()
ring = os.environ.get("MI355Q_PLAN_ENGINE") == "ring"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
L = g.lib()
cfg = dict(wl.LLAMA3_8B)
specs = [s for s in wl.llama_matmuls(cfg, "Q4_K_M") if 0 <= s.layer < a.layers]
stage = B.Stage(torch, g, specs, True, dev)
act = torch.randn((1, cfg["n_embd"]), dtype=torch.float32, device=dev)
plan = stage.make_decode_plan(cfg, act, a.n_ctx, False)
n = plan.launch_stages
grid = 256
buf = torch.zeros(n * grid * 2 * 8 + n * grid * 8 + n * grid * 16 * 4, dtype=torch.int64, device=dev)
setter = L.mi355q_debug_set_ring_stamps if ring else L.mi355q_debug_set_plan_stamps
setter.argtypes = [ctypes.c_void_p, ctypes.c_int]
stage.set_token(a.pos)
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
assert setter(buf.data_ptr(), n) == 0
plan.run(); torch.cuda.synchronize()
assert plan.status() == 0
raw = buf.cpu().numpy()
prof = raw[n * grid * 16:n * grid * 24].reshape(n, grid, 8).astype(np.float64)
pw = raw[n * grid * 24:].reshape(n, grid, 16, 4).astype(np.float64)
s = raw[:n * grid * 16].reshape(n, grid, 2, 8).astype(np.float64)
t0 = s[0, :, :, 0][s[0, :, :, 0] > 0].min()
s = np.where(s > 0, (s - t0) / 100.0, np.nan)          # us
print(f"{n} launch stages; times in us since the first stage entry; per stamp: min / median / max over workgroups (wave 0 | median wave 14)")
for st in range(n):
    line = [f"stage {st:3d}"]
    for i in (5, 0, 2, 6, 7, 3, 1, 4):
        c0, c1 = s[st, :, 0, i], s[st, :, 1, i]
        if np.all(np.isnan(c0)):
            continue
        line.append(f"[{i}] {np.nanmin(c0):7.2f}/{np.nanmedian(c0):7.2f}/{np.nanmax(c0):7.2f} | {np.nanmedian(c1):7.2f}")
    print("  ".join(line))
if a.csv:
    with open(a.csv, "w") as f:
        f.write("stage," + ",".join(f"s{i}_min,s{i}_med,s{i}_max" for i in range(8)) + "\n")
        for st in range(n):
            row = []
            for i in range(8):
                c0 = s[st, :, 0, i]
                row += ["", "", ""] if np.all(np.isnan(c0)) else [f"{np.nanmin(c0):.2f}", f"{np.nanmedian(c0):.2f}", f"{np.nanmax(c0):.2f}"]
            f.write(f"{st}," + ",".join(row) + "\n")
if not ring:
    # the attention stages, workgroup 0: every stamp of wave 0 and of wave 15 (0 entry, 1 q/k/v gathered, 2 roped + stored, 6 scores loop left, 3 scores (barrier), 4 softmax, 7 P.V loop left, 5 published)
    for st in range(n):
        r = s[st, 0]
        if np.isnan(r[0, 6]) or not np.isnan(s[st, grid - 1, 0, 0]):        # (only the first n_head x n_split workgroups run an attention stage)
            continue
        order = (0, 1, 2, 6, 3, 4, 7, 5)
        print(f"stage {st:3d} wg 0  wave 0: " + " ".join(f"[{i}] {r[0, i] - r[0, 0]:6.2f}" for i in order) + "   wave 15: " + " ".join(f"[{i}] {r[1, i] - r[0, 0]:6.2f}" for i in order))
    sys.exit(0)

print("step-loop cycle sums of wave 0 per stage (median over workgroups, shader cycles): bookkeeping | try next | arithmetic + terms + close | waiting for LANDED | waiting for a free SLOT || loader (cumulative): cycles reading the consumers' heads | cycles in s_waitcnt vmcnt | cycles issuing DMA")
for st in range(n):
    if prof[st].sum() > 0:
        v = int(np.median(prof[st, :, 5]))
        print(f"stage {st:3d}  " + "  ".join(f"{np.median(prof[st, :, i]):9.0f}" for i in range(5)) + f"  || tail {v >> 32} page {v & 0xFFFFFFFF}  blocked {np.median(prof[st, :, 6]):.0f}  tailread {np.median(prof[st, :, 7]):.0f}")

print("per consumer wave (median over workgroups): arithmetic cycles | landed-wait cycles | other cycles | finished rows at (us)")
for st in range(n):
    if pw[st].sum() > 0:
        fin = np.where(pw[st, :, :15, 2] > 0, (pw[st, :, :15, 2] - t0) / 100.0, np.nan)
        print(f"stage {st:3d} arith " + " ".join(f"{np.median(pw[st, :, w, 0]):6.0f}" for w in range(15)))
        print(f"          wait  " + " ".join(f"{np.median(pw[st, :, w, 1]):6.0f}" for w in range(15)))
        print(f"          done  " + " ".join(f"{np.nanmedian(fin[:, w]):6.1f}" for w in range(15)))

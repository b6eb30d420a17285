#!/usr/bin/env python3
"""Development tool: the decode plan (one persistent launch per token) vs the per-matmul hipGraph on the
Llama-3-8B chain; also a dependent-chain correctness check (y of one stage IS x of the next).
  python tools/planbench.py [--ftype Q4_K_M] [--reps 50]"""
import argparse, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import torch
import ggml_mi355 as g
from ggml_mi355 import workloads as wl
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--ftype", default="Q4_K_M"); ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--layers", type=int, default=32); ap.add_argument("--no-depends", action="store_true")
ap.add_argument("--no-check", action="store_true", help="timing only (debug modes that produce garbage)")
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)

# ---- correctness: a true chain, square matrices, y_s is x_{s+1}
for t in (() if a.no_check else (g.Q4_K, g.Q6_K, g.Q8_0)):
    K = 4096
    class S: pass
    ws = []
    for i in range(6):
        s = S(); s.type = t; s.M = K; s.K = K; s.nbytes = g.row_size(t, K) * K
        ws.append(B.device_random_weight(torch, g, s, dev))
    x0 = torch.randn((1, K), device=dev)
    bufs = [x0] + [torch.zeros((1, K), device=dev) for _ in ws]
    ref = [x0]
    for w in ws:
        ref.append(g.mul_mat(w, ref[-1]))
    plan = g.Plan([([w], bufs[i], [bufs[i + 1]], i > 0) for i, w in enumerate(ws)])
    for rep in range(3):
        for b in bufs[1:]: b.zero_()
        plan.run(); torch.cuda.synchronize()
        assert plan.status() == 0, "plan aborted"
        ok = all(torch.equal(bufs[i + 1].view(torch.int32), ref[i + 1].view(torch.int32)) for i in range(len(ws)))
        if not ok:
            for i in range(len(ws)):
                d = (bufs[i + 1] - ref[i + 1]).abs()
                print("  stage", i, "equal", torch.equal(bufs[i + 1], ref[i + 1]), "maxdiff", float(d.max()), "nbad", int((d > 0).sum()),
                      "refmax", float(ref[i + 1].abs().max()), "first bad rows", (d[0] > 0).nonzero()[:8].flatten().tolist(), flush=True)
            bad0 = ((bufs[1] != ref[1]) | bufs[1].isnan())[0].nonzero().flatten()
            print("  stage0 bad rows:", len(bad0), "by wave(row%16):", torch.bincount(bad0 % 16, minlength=16).tolist())
            wg = torch.bincount(bad0 // 16, minlength=256)
            print("  bad rows per WG: WGs with any bad", int((wg > 0).sum()), "all16", int((wg == 16).sum()), "first WGs", (wg > 0).nonzero().flatten()[:40].tolist())
            # independent stages (no dependency): every stage on x0
            p2 = g.Plan([([w], x0, [bufs[i + 1]], False) for i, w in enumerate(ws)])
            p2.run(); torch.cuda.synchronize()
            for i, w in enumerate(ws):
                r = g.mul_mat(w, x0)
                print("  independent stage", i, "equal", torch.equal(bufs[i + 1], r), "nbad", int(((bufs[i + 1] - r).abs() > 0).sum()), flush=True)
        print(g.TYPE_NAMES[t], "chain rep", rep, "bit-exact" if ok else "MISMATCH", flush=True)
        assert ok
    plan.close()

# ---- timing on the token chain
cfg = dict(wl.LLAMA3_8B)
specs = [s for s in wl.llama_matmuls(cfg, a.ftype) if s.layer < a.layers]
stage = B.Stage(torch, g, specs, True, dev)
stage.run(); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    stage.run()
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
tg = timeit(graph.replay, a.reps)
refs = [[y.clone() for y in ys] for _, _, ys, _ in stage.groups]
plan = g.Plan([(ws, x, ys, (i > 0 and not a.no_depends)) for i, (ws, x, ys, _) in enumerate(stage.groups)])
for _, _, ys, _ in stage.groups:
    for y in ys: y.zero_()
plan.run(); torch.cuda.synchronize()
assert plan.status() == 0, "plan aborted"
bad = sum(0 if torch.equal(y.view(torch.int32), r.view(torch.int32)) else 1 for (_, _, ys, _), rs in zip(stage.groups, refs) for y, r in zip(ys, rs))
print("plan outputs vs per-matmul launches:", "bit-exact" if bad == 0 else f"{bad} tensors differ", flush=True)
tp = timeit(plan.run, a.reps)
assert plan.status() == 0
nb = stage.bytes
print(f"bytes/token {nb/1e9:.3f} GB  launch stages {plan.launch_stages}")
print(f"graph of {len(stage.groups)} launches: {tg*1e3:.3f} ms  {nb/tg/1e9:.0f} GB/s  {1/tg:.0f} tok/s")
print(f"plan (1 launch)          : {tp*1e3:.3f} ms  {nb/tp/1e9:.0f} GB/s  {1/tp:.0f} tok/s")

#!/usr/bin/env python3
"""bench.py -- llama-bench tg128 on the quantized-matmul hot path, Llama-3-8B Q4_K_M, on MI355X.

One "step" = one generated token = one pass over every quantized matmul weight the token touches
(32 x {attn_q, attn_k, attn_v, attn_output, ffn_gate, ffn_up, ffn_down} + output: 225 matrices, 4.616 GB,
BASELINE.md section 3), N=1 activation column.  Default (--launch plan): the WHOLE decode step of the llama graph
(src/llama-model.cpp llm_build_llama) is ONE persistent launch (mi355q_plan_*): per layer rms_norm*w -> q|k|v, rope +
KV-cache store + causal attention over the f16 cache, attn_output, residual + rms_norm*w -> gate|up, SiLU*up -> down,
then the output norm and the output matrix -- every value handed from stage to stage on the device, the token's position,
window length, cache slots and mask uploaded per token as llama.cpp uploads its graph inputs, and a stream synchronize per
token as llama-bench does (llama_decode + llama_synchronize).  The context grows from empty as in tg128.
--launch graph: the quantized matmuls alone, one launch per step (129), captured in a hipGraph (round 1's path, no glue).
Weights, KV cache and the token embedding input are resident in HBM before the timed region.

  python bench.py [--gpus N --steps K --warmup W]         (N>1: launched by torch.distributed.run)

N>1 = the reference's --split-mode layer: contiguous layer ranges per GPU, the [1, n_embd] f32 activation
handed to the next stage with an RCCL send/recv; a token visits the stages in order (strong scaling: the
curve measures hop overhead).

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "llama.cpp.dsp_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak (6.3 TB/s measured achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)     # tg128
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--ftype", default="Q4_K_M", choices=["Q4_K_M", "Q8_0"])
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3-70b", "mixtral-8x7b"],
                    help="BASELINE.json configs: the headline is llama3-8b; mixtral-8x7b exercises MUL_MAT_ID (per-matmul launches)")
    ap.add_argument("--no-pp", action="store_true", help="skip the pp512 (MFMA tier) measurement")
    ap.add_argument("--launch", default="plan", choices=["plan", "graph", "eager"],
                    help="plan: one persistent launch per token (default); graph: 129 launches in a hipGraph; eager: 129 plain launches")
    ap.add_argument("--no-graph", action="store_true", help="same as --launch eager")
    ap.add_argument("--no-fuse", action="store_true", help="one launch per matrix (no q/k/v, gate/up fusion)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plugin", action="store_true", help="skip the whole-model leg through ggml_backend_graph_compute of the plugin")
    ap.add_argument("--no-llama-bench", action="store_true", help="skip the leg that runs the reference's own llama-bench through the plugin and on the CPU backend")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    ap.add_argument("--dry-run", action="store_true", help="plumbing only (CPU/gloo test of the N>1 hop protocol): no GPU work")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
def device_random_weight(torch, g, spec, device):
    """Random VALID packed rows generated on the device (random payload bytes, finite f16 scales),
    then converted to the device layout with mi355q_weights_pack_d2d.  Expert tensors: n_expert matrices back to back."""
    bs = {g.Q4_K: 144, g.Q5_K: 176, g.Q6_K: 210, g.Q8_0: 34}[spec.type]
    blck = 32 if spec.type == g.Q8_0 else 256
    nb = spec.K // blck
    rows = spec.M * getattr(spec, "n_expert", 1)
    nbytes = rows * nb * bs
    dst = torch.empty(nbytes + 64, dtype=torch.uint8, device=device)
    step = max(1, (1 << 28) // (nb * bs))                      # generate in slabs of <= 256 MiB
    for r0 in range(0, rows, step):
        r1 = min(rows, r0 + step)
        raw = torch.randint(0, 256, (r1 - r0, nb, bs), dtype=torch.uint8, device=device)
        for off in {g.Q4_K: (0, 2), g.Q5_K: (0, 2), g.Q6_K: (208,), g.Q8_0: (0,)}[spec.type]:
            sc = (torch.rand((r1 - r0, nb), device=device) * 0.02 + 1e-3).to(torch.float16)
            raw[:, :, off:off + 2] = sc.view(torch.uint8).view(r1 - r0, nb, 2)
        rc = g.lib().mi355q_weights_pack_d2d(spec.type, dst.data_ptr() + r0 * nb * bs, raw.data_ptr(), r1 - r0, spec.K,
                                             int(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(g.lib().mi355q_last_error().decode())
        torch.cuda.synchronize()
        del raw
    return g.QWeight(spec.type, dst, spec.M, spec.K, getattr(spec, "n_expert", 1))


class Stage:
    """The layers (and possibly the output matrix) one rank owns, as a list of launch groups."""

    def __init__(self, torch, g, specs, fuse, device):
        self.torch, self.g = torch, g
        self.groups = []         # (list[QWeight], x tensor, list[y tensors], bytes a token reads, ids tensor or None)
        self.bytes = 0
        self.has_moe = any(s.n_expert > 1 for s in specs)
        xs = {}
        def x_for(k, rows=1):
            if (k, rows) not in xs:
                xs[(k, rows)] = torch.randn((1, k) if rows == 1 else (1, rows, k), dtype=torch.float32, device=device)
            return xs[(k, rows)]
        by_layer = {}
        for s in specs:
            by_layer.setdefault(s.layer, []).append(s)
        for layer in sorted(by_layer, key=lambda l: (l < 0, l)):
            ss = {s.name.split(".")[-1]: s for s in by_layer[layer]}
            if layer < 0:
                plan = [["output"]]
            elif "ffn_gate_exps" in ss:                        # MoE: every expert tensor is its own MUL_MAT_ID
                plan = [["attn_q", "attn_k", "attn_v"], ["attn_output"], ["ffn_gate_exps"], ["ffn_up_exps"], ["ffn_down_exps"]]
            else:
                plan = [["attn_q", "attn_k", "attn_v"], ["attn_output"], ["ffn_gate", "ffn_up"], ["ffn_down"]]
            if not fuse:
                plan = [[n] for grp in plan for n in grp]
            for grp in plan:
                ws = [device_random_weight(torch, g, ss[n], device) for n in grp]
                for w, n in zip(ws, grp):
                    w.name = ss[n].name
                nbytes = sum(ss[n].nbytes for n in grp)
                if ws[0].n_expert > 1:
                    sp = ss[grp[0]]
                    # ids: the experts the router picked for this token (fixed, distinct per layer); ffn_down gets one activation row per slot
                    ids = torch.tensor([[(3 * max(layer, 0) + u) % sp.n_expert for u in range(sp.n_used)]], dtype=torch.int32, device=device)
                    x = x_for(ws[0].K, sp.n_used if "down" in grp[0] else 1)
                    if x.dim() == 2:
                        x = x.view(1, 1, -1)
                    self.groups.append((ws, x, None, nbytes, ids))
                else:
                    ys = [torch.empty((1, w.M), dtype=torch.float32, device=device) for w in ws]
                    self.groups.append((ws, x_for(ws[0].K), ys, nbytes, None))
                self.bytes += nbytes

    def make_decode_plan(self, cfg, x_in, n_ctx, last_rank):
        """The llama decode step of this rank's layers as ONE persistent launch (see the module docstring).  x_in: the layer input
        ([1, n_embd] f32: the token embedding, or the boundary activation received from the previous rank)."""
        torch, g = self.torch, self.g
        dev = x_in.device
        E, F, nh, nkv, hd = cfg["n_embd"], cfg["n_ff"], cfg["n_head"], cfg["n_head_kv"], cfg["head_dim"]
        kvd = nkv * hd
        f32 = lambda n: torch.zeros((1, n), dtype=torch.float32, device=dev)
        q, k, v, att, o, gate, up, d, hcur, ffn_inp = f32(E), f32(kvd), f32(kvd), f32(E), f32(E), f32(F), f32(F), f32(E), f32(E), f32(E)
        by = {}
        for ws, _, _, _, _ in self.groups:
            for w in ws:
                by[w.name] = w
        layers = sorted({int(n.split(".")[1]) for n in by if n.startswith("blk.")})
        L = len(layers)
        # per-token parameters: ONE device block  [pos i32, n_kv i32 | k_dst, v_dst (L x 2 pointers) | mask f32 [n_ctx]]
        off_dst, off_mask = 16, 16 + 16 * max(L, 1)
        self.par_dev = torch.zeros(off_mask + 4 * n_ctx, dtype=torch.uint8, device=dev)
        self.par_host = torch.zeros(off_mask + 4 * n_ctx, dtype=torch.uint8).pin_memory()
        pos_d = self.par_dev[0:4].view(torch.int32); nkv_d = self.par_dev[4:8].view(torch.int32)
        dst_d = self.par_dev[off_dst:off_mask].view(torch.int64).view(max(L, 1), 2)
        mask_d = self.par_dev[off_mask:].view(torch.float32)
        self.kc = [torch.zeros((n_ctx, kvd), dtype=torch.float16, device=dev) for _ in layers]
        self.vc = [torch.zeros((kvd, n_ctx), dtype=torch.float16, device=dev) for _ in layers]          # transposed V cache (the non-flash graph)
        self.norm_w = [(1.0 + 0.1 * torch.randn(E, device=dev)).float() for _ in range(2 * L + 1)]
        self.n_ctx, self.kvd = n_ctx, kvd
        import numpy as np
        self._par = self.par_host.numpy()
        self._kbase = np.array([t.data_ptr() for t in self.kc], np.int64); self._vbase = np.array([t.data_ptr() for t in self.vc], np.int64)
        self._views = (off_dst, off_mask, L)
        st = []
        x0, x1 = x_in, None
        for li, il in enumerate(layers):
            W = lambda n: by[f"blk.{il}.{n}"]
            last = li == L - 1
            st.append(dict(ws=[W("attn_q"), W("attn_k"), W("attn_v")], ys=[q, k, v], x=x0, x1=x1, x_kind=g.X_NORM, norm_w=self.norm_w[2 * li], eps=1e-5,
                           sum_out=hcur if x1 is not None else None, no_plain=True))
            h_res = hcur if x1 is not None else x0
            st.append(dict(attn=dict(q=q, k=k, v=v, pos=pos_d, n_kv_dev=nkv_d, rope=dict(n_dims=hd, mode=0, n_ctx_orig=8192, freq_base=500000.0),
                                     k_cache=self.kc[li], v_cache=self.vc[li], k_nb_pos=kvd * 2, k_nb_head=hd * 2, v_nb_pos=2, v_nb_dim=n_ctx * 2,
                                     v_nb_head=hd * n_ctx * 2, k_dst=dst_d[li, 0:1], v_dst=dst_d[li, 1:2], v_dst_nb=n_ctx * 2, mask=mask_d,
                                     n_head=nh, n_head_kv=nkv, head_dim=hd, n_kv=n_ctx, scale=1.0 / hd ** 0.5, out=att), no_plain=True))
            st.append(dict(ws=[W("attn_output")], ys=[o], x=att, no_plain=True))
            st.append(dict(ws=[W("ffn_gate"), W("ffn_up")], ys=[gate, up], x=h_res, x1=o, x_kind=g.X_NORM, norm_w=self.norm_w[2 * li + 1], eps=1e-5,
                           sum_out=ffn_inp, y_kind=g.Y_UNARY_MUL, y_unary=g.UNARY_SILU, no_plain=not (last and not last_rank)))     # publishes SiLU(gate) * up
            st.append(dict(ws=[W("ffn_down")], ys=[d], x=gate, no_plain=not (last and not last_rank)))
            x0, x1 = ffn_inp, d
        self.logits = None
        if "output" in by:
            self.logits = torch.zeros((1, by["output"].M), dtype=torch.float32, device=dev)
            st.append(dict(ws=[by["output"]], ys=[self.logits], x=x0, x1=x1, x_kind=g.X_NORM, norm_w=self.norm_w[2 * L], eps=1e-5))
        self.boundary = (ffn_inp, d) if L else None
        self._keep = (q, k, v, att, o, gate, up, d, hcur, ffn_inp, x_in)
        self.kv_bytes_per_pos = 2 * kvd * 2 * L
        return g.Plan(st)

    def set_token(self, pos):
        """Upload the token's graph inputs (position, window length padded to 32 as llama_kv_cache_unified does, this token's cache slots,
        the causal mask row) with ONE host-to-device copy."""
        import numpy as np
        off_dst, off_mask, L = self._views
        n_kv = min(self.n_ctx, (pos + 1 + 31) // 32 * 32)
        p = self._par
        p[0:8].view(np.int32)[:] = (pos, n_kv)
        dst = p[off_dst:off_mask].view(np.int64).reshape(max(L, 1), 2)
        if L:
            dst[:, 0] = self._kbase + pos * self.kvd * 2
            dst[:, 1] = self._vbase + pos * 2
        m = p[off_mask:].view(np.float32)
        m[:] = -np.inf; m[:pos + 1] = 0.0
        self.par_dev.copy_(self.par_host, non_blocking=True)
        return n_kv

    def run_group(self, grp):
        g = self.g
        ws, x, ys, _, ids = grp
        if ids is not None:
            g.mul_mat_id(ws[0], x, ids)
        elif len(ws) == 1:
            g.mul_mat(ws[0], x, out=ys[0])
        else:
            g.mul_mat_multi(ws, x, outs=ys)

    def run(self):
        for grp in self.groups:
            self.run_group(grp)


def kernel_source_sha16():
    """Hash of the sources of the dominant kernel (ties a committed PMC traffic figure to the code it was measured on)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("plan.hip", "gemv_stream.cuh", "act_quant.cuh", "mi355q_common.h"):
        h.update((ROOT / "llama.cpp.dsp_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def measured_hbm_read_GBps(torch, device):
    """Streaming-read rate of this box with a plain 16 B/lane read kernel over 2 GiB (tools/hbm_read.hip)."""
    import ctypes
    so = ROOT / "llama.cpp.dsp_amd" / "lib" / "libmi355q_tools.so"
    if not so.exists():
        return None
    L = ctypes.CDLL(str(so))
    L.mi355q_tool_stream_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    nbytes = 2 << 30
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device).random_(0, 255)
    sink = torch.zeros(4, dtype=torch.int32, device=device)
    st = int(torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for _ in range(6):
        e0.record(); L.mi355q_tool_stream_read(buf.data_ptr(), nbytes, sink.data_ptr(), st); e1.record()
        torch.cuda.synchronize()
        best = max(best, nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del buf, sink
    import _ctypes
    h = L._handle
    del L
    _ctypes.dlclose(h)                                           # (unregisters its code object now, not from an exit handler)
    return round(best, 1)


def verify_token0(torch, g, stage, plan, cfg, x_in):
    """Before timing: the first token (empty context: attention returns the f16-rounded v of its kv head) through the plan must equal the same
    chain issued node by node with the library's per-op entry points (mi355q_op_add_rms_norm_mul, mi355q_mul_mat, mi355q_op_unary_mul)."""
    by = {w.name: w for grp in stage.groups for w in grp[0]}
    layers = sorted({int(n.split(".")[1]) for n in by if n.startswith("blk.")})
    nh, nkv, hd = cfg["n_head"], cfg["n_head_kv"], cfg["head_dim"]
    stage.set_token(0)
    plan.run(); torch.cuda.synchronize()
    if plan.status() != 0:
        raise RuntimeError("decode plan aborted (a poll timed out)")
    h, dn = x_in, None
    for li, il in enumerate(layers):
        W = lambda n: by[f"blk.{il}.{n}"]
        if dn is None:
            x = g.op_add_rms_norm_mul(h, 1e-5, weight=stage.norm_w[2 * li])
        else:
            x, h = g.op_add_rms_norm_mul(h, 1e-5, b=dn, weight=stage.norm_w[2 * li], want_sum=True)
        v = g.mul_mat(W("attn_v"), x)
        att = v.view(nkv, 1, hd).half().float().expand(nkv, nh // nkv, hd).reshape(1, nh * hd).contiguous()
        o = g.mul_mat(W("attn_output"), att)
        x, ffn_inp = g.op_add_rms_norm_mul(h, 1e-5, b=o, weight=stage.norm_w[2 * li + 1], want_sum=True)
        dn = g.mul_mat(W("ffn_down"), g.op_unary_mul(g.UNARY_SILU, g.mul_mat(W("ffn_gate"), x), g.mul_mat(W("ffn_up"), x)))
        h = ffn_inp
    x = g.op_add_rms_norm_mul(h, 1e-5, b=dn, weight=stage.norm_w[2 * len(layers)])
    ref = g.mul_mat(by["output"], x)
    torch.cuda.synchronize()
    err = float((stage.logits - ref).abs().max() / ref.abs().max())
    if not (err <= 1e-4):        # (equal up to the f64 summation order inside the norms and, across 32 layers, what that flips downstream)
        raise RuntimeError(f"decode plan logits differ from the node-by-node path: {err:.3e}")


def cpu_baseline(specs, seconds):
    """The reference ggml CPU backend (oracle/_ref, compiled from /root/reference) on this host's cores:
    the same 225-matmul token chain as ONE ggml graph, N=1."""
    import oracle
    variant = oracle.best_ref_variant()
    if variant is None:
        return None
    ref = oracle.Reference(variant)
    # the GPU box gives one job a 16-CPU share whatever the host's core count says; more threads only oversubscribe it
    threads = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), int(os.environ.get("MI355Q_CPU_THREADS", 16))))
    types = [s.type for s in specs]; Ms = [s.M for s in specs]; Ks = [s.K for s in specs]
    t1 = ref.bench_chain(types, Ms, Ks, 1, threads, 1, 1)
    if t1 <= 0:
        return None
    iters = max(2, min(200, int(seconds / t1)))
    t = ref.bench_chain(types, Ms, Ks, 1, threads, 1, iters)
    return {"value": round(1.0 / t, 3), "unit": "tok/s", "cores": threads, "kind": "reference",
            "sample": f"full token chain ({len(specs)} quantized mul_mats, N=1) as one ggml CPU graph, "
                      f"{iters} evaluations, oracle/_ref/{variant} (ggml CPU backend built from the reference sources)"}


def plugin_graph_compute(seconds_cap=240):
    """The same decode step one level further out: a whole Llama-3-8B-shaped model (32 layers, 128256-row output matrix, random Q4_K_M weights)
    built with the reference's own ggml graph API and run token by token through ggml_backend_graph_compute of libggml-mi355.so, timed per token
    with everything llama_decode does around it (graph build, allocation, input upload, compute, synchronize) -- and the reference's CPU backend
    on the same graphs beside it.  The host side is the reference's libggml (oracle/_ref, compiled from /root/reference; it cannot be shipped in
    this repo) driven by oracle/model_parity/model_parity.cc, so this leg belongs to the CPU-baseline half of the report: a measurement of the
    product through the reference's plugin API, never a dependency of the product."""
    import re, subprocess
    import oracle
    variant = oracle.best_ref_variant()
    exe = ROOT / "oracle" / "_ref" / (variant or "avx2") / "model_parity"
    plugin = ROOT / "llama.cpp.dsp_amd" / "lib" / "libggml-mi355.so"
    if variant is None or not exe.exists() or not plugin.exists():
        return None
    env = dict(os.environ, GGML_BACKEND_PATH=str(plugin), MI355_GRAPH_STATS="1")
    r = subprocess.run([str(exe), "--preset", "8b", "--layers", "32", "--vocab", "128256", "--tokens", "2", "--bench", "128", "--pp", "512"], env=env, capture_output=True, text=True, timeout=seconds_cap)
    m = re.search(r"decode through (\S+) \((.*?)\): \S+ ([0-9.]+) us/token = ([0-9.]+) tok/s(?:; CPU backend ([0-9.]+) us/token = ([0-9.]+) tok/s \((\d+) tokens\))?", r.stdout)
    if not m:
        return {"error": f"model_parity exit {r.returncode}", "tail": (r.stdout + r.stderr)[-400:]}
    # (exit code 1 only says that the live comparison exceeded the single-op bound of 1e-3: whole-model logits of two correct evaluations differ
    # by more -- DESIGN.md section 3b; tests/test_plugin.py holds them to the reference's own build-to-build spread with committed fixtures)
    plans = re.search(r"MI355 decode plans: (\d+) graph_compute calls ran as one persistent launch", r.stdout + r.stderr)
    par = re.search(r"worst logits NMSE ([0-9.e+-]+), worst max\|d\|/max\|ref\| ([0-9.e+-]+)", r.stdout)
    out = {"value": float(m.group(4)), "unit": "tok/s", "us_per_token": float(m.group(3)), "entry": m.group(1), "model": m.group(2),
           "tokens": 128, "graph_compute_calls_as_one_launch": int(plans.group(1)) if plans else None,
           "logits_vs_cpu_first_tokens": {"nmse": float(par.group(1)), "max_rel": float(par.group(2))} if par else None,
           "harness": f"oracle/_ref/{variant}/model_parity (reference libggml host + this repo's plugin via GGML_BACKEND_PATH)"}
    pp = re.search(r"prefill through \S+ \(.*?\): (\d+) tokens in ([0-9.]+) us = ([0-9.]+) tok/s", r.stdout)
    if pp:
        out["pp512"] = {"value": float(pp.group(3)), "unit": "tok/s", "ms": round(float(pp.group(2)) / 1e3, 3),
                        "what": f"a {pp.group(1)}-token prompt from an empty context through graph_compute (all nodes: norms, rope, KV stores, attention, matmuls; logits of the last token), as llama-bench pp"}
    hp = re.search(r"graph build ([0-9.]+) us, allocation ([0-9.]+) us, inputs ([0-9.]+) us, graph_compute call ([0-9.]+) us, synchronize \+ logit ([0-9.]+) us", r.stdout)
    if hp:
        out["host_phases_us"] = dict(zip(("graph_build", "allocation", "inputs", "graph_compute_call", "synchronize_and_logit"), (float(v) for v in hp.groups())))
    if m.group(5):
        out["cpu_backend"] = {"value": float(m.group(6)), "unit": "tok/s", "us_per_token": float(m.group(5)), "tokens": int(m.group(7)),
                              "kind": "reference", "what": "the same graphs on the reference's CPU backend (ggml_backend_cpu, all host threads)"}
    return out


def llama_bench_leg(reps=2, threads=16, timeout_s=420):
    """The metric's OWN harness: the reference's unmodified llama-bench (examples/llama-bench/llama-bench.cpp:1430-1468 test_prompt / test_gen, compiled
    with libllama from the reference sources by oracle/Makefile into oracle/_ref/avx2) on a synthetic Llama-3-8B Q4_K_M GGUF (oracle/gguf_synth: exact
    shapes and tensor-type mix, random weights, `no_vocab` tokenizer), pp512 + tg128: once with every layer on the plugin (-ngl 99, the plugin loaded
    through GGML_BACKEND_PATH) and once on the reference CPU backend alone (-ngl 0, no plugin loaded), same binary, same box, same file, one after the
    other.  A TIMER: it pins no parity (tests/ do).  Returns None where the harness is not built."""
    import re, subprocess
    ref = ROOT / "oracle" / "_ref" / "avx2"
    exe, synth, plugin = ref / "llama-bench", ref / "gguf_synth", ROOT / "llama.cpp.dsp_amd" / "lib" / "libggml-mi355.so"
    if not (exe.exists() and synth.exists() and plugin.exists()):
        return None
    gguf = Path(os.environ.get("TMPDIR", "/tmp")) / "mi355_llama3_8b_q4_k_m.synthetic.gguf"
    if not gguf.exists():
        r = subprocess.run([str(synth), "--preset", "8b", "--ftype", "q4_k_m", "--out", str(gguf)], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": "gguf_synth failed", "tail": (r.stdout + r.stderr)[-300:]}
    def run(ngl, env, r_, extra=(), p="512"):
        cmd = [str(exe), "-m", str(gguf), "-p", p, "-n", "128", "-r", str(r_), "-ngl", str(ngl), "-t", str(threads), "-o", "json"] + list(extra)
        pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        if pr.returncode != 0:
            return None, pr
        rows = json.loads(pr.stdout[pr.stdout.index("["):])
        res = {}
        for row in rows:
            key = "pp512" if row["n_prompt"] else "tg128"
            res[key] = {"value": round(row["avg_ts"], 2), "stddev": round(row["stddev_ts"], 2), "unit": "tok/s"}
        res["backends"] = rows[0].get("backends"); res["model_type"] = rows[0].get("model_type"); res["model_size_bytes"] = rows[0].get("model_size")
        return res, pr
    env_gpu = dict(os.environ, GGML_BACKEND_PATH=str(plugin), MI355_GRAPH_STATS="1")
    gpu, pr = run(99, env_gpu, reps)
    out = {"harness": "oracle/_ref/avx2/llama-bench (the reference's llama-bench + libllama, unmodified sources) -p 512 -n 128", "model": "synthetic Llama-3-8B Q4_K_M GGUF (oracle/_ref/avx2/gguf_synth)",
           "reps": reps}
    if gpu is None:
        out["ngl99"] = {"error": f"exit {pr.returncode}", "tail": (pr.stdout + pr.stderr)[-400:]}
    else:
        plans = re.findall(r"MI355 decode plans: (\d+) graph_compute calls ran as one persistent launch, (\d+) plans built", pr.stderr)
        gpu["graph_compute_calls_as_one_launch"] = sum(int(a_) for a_, _ in plans) if plans else None      # expected: (reps + warm-up) x 128 + warm-up tokens
        gpu["plans_built"] = sum(int(b_) for _, b_ in plans) if plans else None
        out["ngl99"] = gpu
        # the other attention forms of the same run: -fa 1 (FLASH_ATTN_EXT: one launch per token, the CPU's F16 accumulator) and a Q8_0 KV cache (resident, node path)
        for key, extra in (("ngl99_fa1", ["-fa", "1"]), ("ngl99_fa1_kv_q8_0", ["-fa", "1", "-ctk", "q8_0", "-ctv", "q8_0"])):
            r2, pr2 = run(99, env_gpu, 1, extra, p="0")
            if r2 is None:
                out[key] = {"error": f"exit {pr2.returncode}", "tail": (pr2.stdout + pr2.stderr)[-300:]}
            else:
                pl2 = re.findall(r"MI355 decode plans: (\d+) graph_compute calls ran as one persistent launch, (\d+) plans built", pr2.stderr)
                out[key] = {"tg128": r2.get("tg128"), "graph_compute_calls_as_one_launch": sum(int(a_) for a_, _ in pl2) if pl2 else None}
    env_cpu = {k: v for k, v in os.environ.items() if k != "GGML_BACKEND_PATH"}
    cpu, pr = run(0, env_cpu, 1)
    out["ngl0_cpu"] = cpu if cpu is not None else {"error": f"exit {pr.returncode}", "tail": (pr.stdout + pr.stderr)[-400:]}
    if cpu is not None:
        out["ngl0_cpu"]["threads"] = threads
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            print(f"bench.py: --gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    import torch
    import torch.distributed as dist
    import ggml_mi355 as g
    from ggml_mi355 import workloads as wl
    rehearse = os.environ.get("MI355Q_BENCH_REHEARSE") == "1" and not a.dry_run

    def p2p_send(t, dst):
        dist.send(t.cpu() if rehearse else t, dst=dst)

    def p2p_recv(t, src):
        if rehearse:
            h = torch.empty(t.shape, dtype=t.dtype)
            dist.recv(h, src=src)
            t.copy_(h)
        else:
            dist.recv(t, src=src)

    if a.dry_run:
        device = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
    else:
        # MI355Q_BENCH_REHEARSE=1 (tests/test_gpu_bench.py): every rank on GPU 0 and the hop through gloo / host memory, so that the
        # multi-rank code path (per-rank plans, boundary tensors, hop and completion protocol) runs on a one-GPU box.  Never a measurement.
        torch.cuda.set_device(0 if rehearse else local_rank)
        device = torch.device("cuda", 0 if rehearse else local_rank)
        g.lib()                                     # fail loudly if the HIP extension is missing
        if world > 1:
            if rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=device)

    cfg = wl.MODELS[a.model]
    specs = wl.llama_matmuls(cfg, a.ftype)
    total_bytes = sum(s.nbytes for s in specs)
    ranges = wl.partition_layers(cfg["n_layer"], world)
    mine = [s for s in specs if s.layer in ranges[rank] or (s.layer < 0 and rank == world - 1)]

    act = torch.zeros((1, cfg["n_embd"]), dtype=torch.float32, device=device)

    def token_done(flag):
        """llama-bench generates token t+1 only after llama_synchronize() has seen token t leave the LAST device (examples/llama-bench/llama-bench.cpp:1454-1468
        test_gen: llama_decode, llama_synchronize, next token).  With one process per GPU that wait is a one-word message from the last stage back to
        the first: without it the stages would overlap different tokens and the "scaling" would be pipeline throughput no decode loop can have."""
        if world > 1:
            if rank == world - 1:
                p2p_send(flag, 0)
            elif rank == 0:
                p2p_recv(flag, world - 1)

    if a.dry_run:
        stage = None
        done = torch.zeros(1, dtype=torch.int32)
        def token():
            if world > 1 and rank > 0:
                dist.recv(act, src=rank - 1)
            if world > 1 and rank < world - 1:
                dist.send(act, dst=rank + 1)
            token_done(done)
    else:
        stage = Stage(torch, g, mine, not a.no_fuse, device)
        launch = "eager" if a.no_graph else a.launch
        if stage.has_moe and launch == "plan":
            launch = "graph"                                    # the decode plan covers MUL_MAT chains; MUL_MAT_ID runs per launch
        graph = plan = None
        stage.run(); torch.cuda.synchronize()                  # warm every kernel / attribute
        n_ctx = max(32, (max(a.steps, a.warmup) + 31) // 32 * 32)
        if launch == "graph":
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                stage.run()
        elif launch == "plan":
            if rank == 0:
                act.copy_(torch.randn_like(act))                # the token embedding (GET_ROWS of token_embd runs on the CPU in llama.cpp)
            plan = stage.make_decode_plan(cfg, act, n_ctx, rank == world - 1)
            if world == 1:
                verify_token0(torch, g, stage, plan, cfg, act)
        act_out = torch.zeros_like(act)
        done = torch.zeros(1, dtype=torch.int32, device=device)
        tok = [0]
        # llama_decode copies the token's logits to the host before llama_synchronize returns (src/llama-context.cpp: get_tensor_async of t_logits)
        logits_dev = getattr(stage, "logits", None) if (plan is not None and rank == world - 1) else None
        logits_host = torch.empty(logits_dev.shape, dtype=logits_dev.dtype, pin_memory=True) if logits_dev is not None else None
        def token():
            if world > 1 and rank > 0:
                p2p_recv(act, rank - 1)                         # boundary activation from the previous stage: the first norm of this rank reads it
            if plan is not None:
                stage.set_token(tok[0] % n_ctx); tok[0] += 1
                plan.run()
                if world > 1 and rank < world - 1:
                    torch.add(stage.boundary[0], stage.boundary[1], out=act_out)       # the layer output h = ffn_inp + ffn_down
                    p2p_send(act_out, rank + 1)
                if logits_host is not None:
                    logits_host.copy_(logits_dev, non_blocking=True)
                token_done(done)
                torch.cuda.current_stream().synchronize()       # llama-bench: llama_decode + llama_synchronize per generated token
            else:
                (graph.replay if graph is not None else stage.run)()
                if world > 1 and rank < world - 1:
                    p2p_send(act, rank + 1)
                token_done(done)

    def sync():
        if not a.dry_run:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not a.dry_run:
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        token()
    sync()
    if not a.dry_run:
        tok[0] = 0                                              # tg: the timed tokens start from an empty context
    t0 = time.perf_counter()
    for _ in range(a.steps):
        token()
    sync()
    dt = time.perf_counter() - t0
    if not a.dry_run and plan is not None and plan.status() != 0:
        raise RuntimeError(f"rank {rank}: decode plan aborted (a poll timed out)")
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        if rank == world - 1 and not a.dry_run and getattr(stage, "logits", None) is not None:
            if not bool(torch.isfinite(stage.logits).all()):
                raise RuntimeError("the last stage's logits are not finite")

    out = {
        "metric": "llama-bench tg128 tok/s (decode step at the C-ABI of the quantized-matmul hot path), " + {"llama3-8b": "Llama-3-8B", "llama3-70b": "Llama-3-70B", "mixtral-8x7b": "Mixtral-8x7B"}[a.model] + " " + a.ftype,
        "value": round(a.steps / dt, 2), "unit": "tok/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "int8",
        "data": "synthetic (random packed blocks with finite scales, gaussian activations)",
        "config": {"workload": f"{a.model} {a.ftype} tg (N=1): all {len(specs)} quantized mul_mat{'/mul_mat_id' if stage is not None and stage.has_moe else ''} weights per token, "
                               f"{total_bytes / 1e9:.3f} GB/token; {'fused q|k|v and gate|up steps, ' if not a.no_fuse else ''}"
                               + {"plan": "the whole llama decode step (norms, rope, KV store, attention over the growing f16 cache, SiLU, residuals, output matrix) as one persistent launch per token, "
                                          "stage-to-stage hand-off by tagged granules, graph inputs uploaded and the stream synchronized per token as llama-bench does",
                                  "graph": "one launch per matmul step, hipGraph replay; non-matmul graph ops not executed",
                                  "eager": "one launch per matmul step, eager; non-matmul graph ops not executed"}[launch if not a.dry_run else "eager"],
                   "bytes_per_token": total_bytes,
                   "parallelism": "single GPU" if world == 1 else f"layer split over {world} GPUs (RCCL send/recv of the boundary activation; a token starts when the previous one has left the last GPU, as llama_synchronize makes llama-bench wait)"},
    }

    if not a.dry_run and rank == 0:
        # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream ----
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        def timed(fn, reps):
            fn(); torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / reps
        # (a) the per-matmul launch path: every launch is the same kernel template (k_gemv_fast); its instantiations are timed
        # per (weight type, K) class, the launches of a class captured back to back in their own hipGraph.
        classes = {}
        for grp in stage.groups:
            ws = grp[0]
            classes.setdefault((g.TYPE_NAMES[ws[0].type] + ("_id" if grp[4] is not None else ""), ws[0].K), []).append(grp)
        per_kernel = {}
        for (tname, kk), grp in classes.items():
            def run_class():
                for gg in grp:
                    stage.run_group(gg)
            run_class(); torch.cuda.synchronize()
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                run_class()
            secs = timed(cg.replay, 20)
            del cg
            nb = sum(t[3] for t in grp)
            per_kernel[f"k_gemv_fast<{tname}, K={kk}, N=1>"] = {"launches_per_token": len(grp), "bytes_per_token": nb, "avg_launch_us": round(1e6 * secs / len(grp), 2),
                                                                "algorithmic_bytes_per_launch": nb // len(grp), "GBps": round(nb / secs / 1e9, 1)}
        if plan is not None:
            # (b) the persistent launch IS the token: one kernel; algorithmic bytes = every weight byte of this rank's layers + the K / V
            # cache window the token attends to (here the longest one of the run)
            n_kv_t = stage.set_token(n_ctx - 1)
            secs = timed(plan.run, 20)
            if plan.status() != 0:
                raise RuntimeError("decode plan aborted (a poll timed out)")
            alg = stage.bytes + stage.kv_bytes_per_pos * n_kv_t
            dom_name = "k_plan (persistent decode plan, %d stages)" % plan.launch_stages
            dom = {"launches_per_token": 1, "bytes_per_token": alg, "avg_launch_us": round(1e6 * secs, 2), "n_kv": n_kv_t,
                   "algorithmic_bytes_per_launch": alg, "GBps": round(alg / secs / 1e9, 1)}
            all_kernels = {dom_name: dom, "per_matmul_launch_path_for_comparison": per_kernel}
            # round 1's quantity, for continuity: the token's quantized matmuls ALONE, one launch per group, nothing executed between them
            mm_us = sum(v["avg_launch_us"] * v["launches_per_token"] for v in per_kernel.values())
            out["matmul_chain_only"] = {"value": round(1e6 / mm_us, 1), "unit": "tok/s", "us_per_token": round(mm_us, 1),
                                        "GBps": round(total_bytes / mm_us / 1e3, 1) if world == 1 else None,
                                        "what": "sum over the per-(type, K) launch classes of launches x average duration (hipGraph per class, HIP events): "
                                                "the 225 quantized matmuls of a token with no norm / rope / attention / activation between them -- what round 1 reported as value"}
        else:
            dom_name, dom = max(per_kernel.items(), key=lambda kv: kv[1]["bytes_per_token"])
            all_kernels = per_kernel
        achieved = dom["GBps"]
        # HBM bytes per launch come from a separate rocprofv3 --pmc FETCH_SIZE pass of this same command (a profiler cannot run inside
        # the timed process).  The committed summary is quoted ONLY while it was taken from the kernel sources as they are now (their
        # hash is stored with it); otherwise null -- never a stale number.
        traffic, traffic_src = None, None
        if plan is not None and a.ftype == "Q4_K_M" and a.model == "llama3-8b" and world == 1:
            for tf in sorted((ROOT / "profiles").glob("round*_traffic.json"), reverse=True):
                try:
                    tj = json.loads(tf.read_text())
                    vals = [v for k, v in tj.get("kernels", {}).items() if k.startswith("k_plan")]
                    if vals and tj.get("kernel_source_sha16") == kernel_source_sha16():
                        traffic, traffic_src = int(vals[0]), f"{tj.get('source')}: {tj.get('method')}"
                        break
                except Exception:
                    pass
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                           "kernel": dom_name, "launches_per_token": dom["launches_per_token"],
                           "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "avg_launch_us": dom["avg_launch_us"],
                           "whole_token_GBps": round(total_bytes * a.steps / dt / 1e9, 1) if world == 1 else None,
                           "measured_hbm_read_peak_GBps": measured_hbm_read_GBps(torch, device),   # this box, plain streaming read (guide: ~6.3 TB/s)
                           "all_kernels": all_kernels}
        out["roofline"]["frac_of_measured"] = round(achieved / out["roofline"]["measured_hbm_read_peak_GBps"], 4)      # north_star's target (>= 0.70) is against THIS peak
        # ---- pp512: the same weights at N = 512 on the MFMA tier (second half of the north-star metric) ----
        if not a.no_pp and world == 1:
            Npp = 512
            xs_pp = {k: torch.randn((Npp, k), dtype=torch.float32, device=device) for k in {w.K for grp in stage.groups for w in grp[0]}}
            ys_pp = {m: torch.empty((Npp, m), dtype=torch.float32, device=device) for m in {w.M for grp in stage.groups for w in grp[0]}}
            # MoE: every token picks n_used distinct experts at random (a uniform router); MUL_MAT_ID groups the rows by expert
            ids_pp, xs_moe = {}, {}
            for grp in stage.groups:
                if grp[4] is not None:
                    w = grp[0][0]
                    n_used = int(grp[4].shape[-1])
                    if (w.n_expert, n_used) not in ids_pp:
                        ids_pp[(w.n_expert, n_used)] = torch.stack([torch.randperm(w.n_expert, device=device)[:n_used] for _ in range(Npp)]).to(torch.int32)
                    if w.K not in xs_moe:
                        xs_moe[w.K] = torch.randn((Npp, 1, w.K), dtype=torch.float32, device=device)
            def pp():
                for grp in stage.groups:
                    if grp[4] is not None:
                        w = grp[0][0]
                        g.mul_mat_id(w, xs_moe[w.K], ids_pp[(w.n_expert, int(grp[4].shape[-1]))])
                    elif len(grp[0]) == 1:
                        g.mul_mat(grp[0][0], xs_pp[grp[0][0].K], out=ys_pp[grp[0][0].M])
                    else:                                 # wq|wk|wv, ffn_gate|ffn_up: one call, one prepared copy of the activations
                        g.mul_mat_multi(grp[0], xs_pp[grp[0][0].K], outs=ys_multi[id(grp)])
            ys_multi = {id(grp): [torch.empty((Npp, w.M), dtype=torch.float32, device=device) for w in grp[0]] for grp in stage.groups if len(grp[0]) > 1}
            secs = timed(pp, 3)
            flop = 2.0 * Npp * sum(w.M * w.K * (int(grp[4].shape[-1]) if grp[4] is not None else 1) for grp in stage.groups for w in grp[0])
            out["pp512"] = {"value": round(Npp / secs, 1), "unit": "tok/s", "ms": round(1e3 * secs, 3), "TFLOPs": round(flop / secs / 1e12, 1),
                            "roofline": {"bound": "mfma", "achieved": round(flop / secs / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                         "frac": round(flop / secs / 1e12 / 2500.0, 4),
                                         "kernel": "k_mmq_i8_q4k (Q4_K: int8 MFMA 16x16x64 on Q8_K activations, exact integer block sums) + k_mmq_bf16 (other types: bf16 MFMA 16x16x32)",
                                         "peak_note": "dense bf16 MFMA peak; the int8 instructions of the Q4_K kernel have twice that"},
                            "note": "all quantized matmuls of the model at N = 512, one launch each (+ one activation-preparation launch)"}
            del xs_pp, ys_pp
        if not a.no_cpu_baseline and world == 1 and a.model == "llama3-8b":      # (the headline config; the reference chain has no MUL_MAT_ID leg)
            try:
                out["cpu_baseline"] = cpu_baseline(specs, a.cpu_seconds)
            except Exception as e:                              # the baseline is a report, never a reason to fail the bench
                out["cpu_baseline"] = {"error": repr(e)}
            if plan is not None and a.ftype == "Q4_K_M" and not a.no_plugin:
                try:
                    out["graph_compute"] = plugin_graph_compute()
                except Exception as e:
                    out["graph_compute"] = {"error": repr(e)}
            if plan is not None and a.ftype == "Q4_K_M" and not a.no_llama_bench:
                try:
                    out["llama_bench"] = llama_bench_leg()
                except Exception as e:
                    out["llama_bench"] = {"error": repr(e)}
                lb = out.get("llama_bench") or {}
                cpu_tg = ((lb.get("ngl0_cpu") or {}).get("tg128") or {}).get("value")
                if cpu_tg:
                    # BASELINE.md publishes no number for this metric on this hardware; what north_star names as the baseline is "the reference ggml CPU path
                    # timed on the same box's host cores in the same llama-bench run": that tg128 figure, measured a minute ago by the same binary
                    out["vs_baseline"] = round(out["value"] / cpu_tg, 2)
                    out["vs_baseline_is"] = f"value / tg128 of the reference CPU backend in llama-bench on this box ({cpu_tg} tok/s, {lb['ngl0_cpu'].get('threads')} threads); BASELINE.md holds no published figure for MI355X"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not a.dry_run:
        # Explicit, ordered teardown (nothing that owns HIP resources is left to interpreter exit / library finalizers): launch graphs and
        # events first, then the plan, the weights, the cached workspaces; synchronize; unload the HIP library while the runtime is alive.
        import gc
        if plan is not None:
            plan.close()
        graph = plan = run_token = token = None
        stage.groups.clear(); stage = None
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        g.shutdown()
        if os.environ.get("MI355Q_LOG_MAPS"):                   # attribution of exit-time frames (diagnostic)
            Path(os.environ["MI355Q_LOG_MAPS"]).write_text(Path("/proc/self/maps").read_text())


if __name__ == "__main__":
    main()

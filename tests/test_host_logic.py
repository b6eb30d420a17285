"""Host-side logic that needs no GPU: the synthetic workload (Q4_K_M tensor-type recipe, bytes per token),
the --split-mode layer partition, and the N>1 hop protocol of bench.py over gloo with 2 ranks."""
import json
import os
import subprocess
import sys

import pytest

import ggml_mi355 as g
from ggml_mi355 import workloads as wl
from conftest import ROOT


def test_q4_k_m_recipe_and_bytes_per_token():
    specs = wl.llama_matmuls(wl.LLAMA3_8B, "Q4_K_M")
    assert len(specs) == 32 * 7 + 1
    # use_more_bits (src/llama-quant.cpp:129-131): 16 of 32 layers get Q6_K for attn_v and ffn_down
    more = [il for il in range(32) if wl.use_more_bits(il, 32)]
    assert len(more) == 16 and more[:4] == [0, 1, 2, 3] and more[-4:] == [28, 29, 30, 31] and 6 in more and 5 not in more
    q6 = [s for s in specs if s.type == g.Q6_K]
    assert len(q6) == 2 * 16 + 1 and specs[-1].name == "output" and specs[-1].type == g.Q6_K
    total = sum(s.nbytes for s in specs)
    assert total == 4616331264                        # 4.616 GB/token
    q4 = sum(s.nbytes for s in specs if s.type == g.Q4_K)
    assert abs(q4 / 1e9 - 3.360) < 0.005 and abs((total - q4) / 1e9 - 1.257) < 0.005
    assert sum(s.nbytes for s in wl.llama_matmuls(wl.LLAMA3_8B, "Q8_0")) == 7973699584   # 7.974 GB/token
    assert abs(sum(s.nbytes for s in wl.llama_matmuls(wl.LLAMA3_70B, "Q4_K_M")) / 1e9 - 41.9) < 1.0   # 70B (Q5_K bump not modelled)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_layer_partition(world):
    parts = wl.partition_layers(32, world)
    assert len(parts) == world
    assert [l for p in parts for l in p] == list(range(32))          # contiguous, complete, ordered
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_two_rank_hop_protocol_gloo(tmp_path):
    """world_size 2 over gloo on the CPU: rank 0 sends the boundary activation, rank 1 receives it, the timing
    is max-reduced and rank 0 prints the one JSON line.  (--dry-run: plumbing only, no compute anywhere.)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29617", str(ROOT / "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--dry-run"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "strong"
    assert out["unit"] == "tok/s" and out["value"] > 0
    assert "layer split over 2 GPUs" in out["config"]["parallelism"]


def test_other_model_recipes():
    """Q4_K_M type recipes of the other BASELINE.json configs (src/llama-quant.cpp:235-322) and what a token reads."""
    from ggml_mi355 import workloads as wl, Q4_K, Q5_K, Q6_K, Q8_0
    s70 = wl.llama_matmuls(wl.LLAMA3_70B, "Q4_K_M")
    assert len(s70) == 80 * 7 + 1
    v = [s for s in s70 if s.name.endswith("attn_v")]
    assert {s.type for s in v} == {Q5_K, Q6_K}                       # 70B: attn_v never stays Q4_K
    assert sum(s.type == Q6_K for s in v) == sum(wl.use_more_bits(i, 80) for i in range(80))
    assert 41.5e9 < sum(s.nbytes for s in s70) < 42.5e9
    mx = wl.llama_matmuls(wl.MIXTRAL_8X7B, "Q4_K_M")
    by = {s.name.split(".")[-1]: s for s in mx if s.layer == 0}
    assert by["attn_k"].type == Q8_0 and by["attn_v"].type == Q8_0 and by["attn_output"].type == Q5_K and by["attn_q"].type == Q4_K
    assert by["ffn_gate_exps"].n_expert == 8 and by["ffn_gate_exps"].n_used == 2 and by["ffn_down_exps"].type == Q6_K
    assert by["ffn_up_exps"].stored_bytes == 4 * by["ffn_up_exps"].nbytes      # a token reads 2 of 8 experts
    assert 28.0e9 < sum(s.stored_bytes for s in mx) < 28.8e9 and 7.8e9 < sum(s.nbytes for s in mx) < 8.0e9

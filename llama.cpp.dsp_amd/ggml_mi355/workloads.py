"""Synthetic workloads of the hot path: the quantized matmul set one generated token touches.

Tensor shapes are those libllama builds for the llama architecture (src/llama-model.cpp:4451-4539, 4618);
the per-tensor quantization types follow the Q4_K_M recipe of the reference's quantizer
(src/llama-quant.cpp:129-131 use_more_bits, :151-168 output -> Q6_K, :235-249 attn_v, :291-297 ffn_down).
`token_embd` is excluded: it is a 1-row GET_ROWS on the CPU."""
from __future__ import annotations

from dataclasses import dataclass

from . import Q4_K, Q5_K, Q6_K, Q8_0, row_size


@dataclass(frozen=True)
class MatSpec:
    name: str
    type: int
    M: int
    K: int
    layer: int          # -1 for the output matrix
    n_expert: int = 1   # > 1: a MUL_MAT_ID weight [K, M, n_expert] (ffn_*_exps), of which a token uses n_used
    n_used: int = 1

    @property
    def nbytes(self) -> int:
        """bytes a generated token READS: every row of a dense matrix, n_used experts of an expert tensor"""
        return row_size(self.type, self.K) * self.M * self.n_used

    @property
    def stored_bytes(self) -> int:
        return row_size(self.type, self.K) * self.M * self.n_expert


def use_more_bits(i_layer: int, n_layers: int) -> bool:
    """src/llama-quant.cpp:129-131"""
    return i_layer < n_layers // 8 or i_layer >= 7 * n_layers // 8 or (i_layer - n_layers // 8) % 3 == 2


LLAMA3_8B = dict(n_layer=32, n_embd=4096, n_ff=14336, n_head=32, n_head_kv=8, head_dim=128, n_vocab=128256)
LLAMA3_70B = dict(n_layer=80, n_embd=8192, n_ff=28672, n_head=64, n_head_kv=8, head_dim=128, n_vocab=128256, is_70b=True)
MIXTRAL_8X7B = dict(n_layer=32, n_embd=4096, n_ff=14336, n_head=32, n_head_kv=8, head_dim=128, n_vocab=32000, n_expert=8, n_expert_used=2)
MODELS = {"llama3-8b": LLAMA3_8B, "llama3-70b": LLAMA3_70B, "mixtral-8x7b": MIXTRAL_8X7B}


def llama_matmuls(cfg: dict, ftype: str = "Q4_K_M") -> list[MatSpec]:
    """The quantized MUL_MAT / MUL_MAT_ID weights of a llama-architecture model (dense or 8-expert MoE) with the types the
    reference's quantizer gives them (src/llama-quant.cpp: attn_v :235-249, attn_k :250-256, ffn_down :269-313,
    attn_output :314-322, output :151-168).  The MoE router (ffn_gate_inp, f32) is not a quantized matmul."""
    L, E, F = cfg["n_layer"], cfg["n_embd"], cfg["n_ff"]
    kv = cfg["n_head_kv"] * cfg["head_dim"]
    n_exp, n_used = cfg.get("n_expert", 1), cfg.get("n_expert_used", 1)
    out: list[MatSpec] = []
    for il in range(L):
        if ftype == "Q4_K_M":
            more = use_more_bits(il, L)
            t = t_k = t_o = Q4_K
            t_v = Q6_K if more else Q4_K
            if cfg.get("is_70b") and t_v == Q4_K:
                t_v = Q5_K                       # 8 heads share attn_v in the 70B model: more bits (:239-243)
            if n_exp == 8:
                t_k = t_v = Q8_0                 # (:244-255)
                t_o = Q5_K                       # (:316-321)
            t_down = Q6_K if more else Q4_K
        elif ftype == "Q8_0":
            t = t_k = t_o = t_v = t_down = Q8_0
        else:
            raise ValueError(ftype)
        out += [
            MatSpec(f"blk.{il}.attn_q", t, E, E, il), MatSpec(f"blk.{il}.attn_k", t_k, kv, E, il),
            MatSpec(f"blk.{il}.attn_v", t_v, kv, E, il), MatSpec(f"blk.{il}.attn_output", t_o, E, E, il),
        ]
        if n_exp > 1:
            out += [MatSpec(f"blk.{il}.ffn_gate_exps", t, F, E, il, n_exp, n_used), MatSpec(f"blk.{il}.ffn_up_exps", t, F, E, il, n_exp, n_used),
                    MatSpec(f"blk.{il}.ffn_down_exps", t_down, E, F, il, n_exp, n_used)]
        else:
            out += [MatSpec(f"blk.{il}.ffn_gate", t, F, E, il), MatSpec(f"blk.{il}.ffn_up", t, F, E, il),
                    MatSpec(f"blk.{il}.ffn_down", t_down, E, F, il)]
    out.append(MatSpec("output", Q6_K if ftype == "Q4_K_M" else Q8_0, cfg["n_vocab"], E, -1))
    return out


def partition_layers(n_layer: int, world: int) -> list[range]:
    """Contiguous layer ranges per device, as --split-mode layer does with equal free memory
    (src/llama-model.cpp:1438-1497: split points proportional to device memory, upper_bound lookup)."""
    bounds = [round(n_layer * (r + 1) / world) for r in range(world)]
    starts = [0] + bounds[:-1]
    return [range(s, e) for s, e in zip(starts, bounds)]

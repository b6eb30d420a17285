#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the REAL reference.

Run where /root/reference exists (this container):  python tests/golden/make_golden.py
It drives oracle/_ref/scalar (the reference ggml CPU backend compiled from /root/reference by
oracle/Makefile with no SIMD, so every vec_dot takes its ISA-independent scalar branch):

  weights  : ggml_quantize_chunk (ggml/src/ggml.c:6386) of seeded uniform(-1,1) floats, as
             tests/test-backend-ops.cpp:39-128 does (with a fixed seed instead of random_device)
  act      : the CPU backend's activation quantizer for the type's vec_dot_type
  deq      : dequantize_row_<type> (ggml/src/ggml-quants.c)
  y        : GGML_OP_MUL_MAT computed by the CPU backend (ggml-cpu.c:1266-1458)
  y_id     : GGML_OP_MUL_MAT_ID computed by the CPU backend (ggml-cpu.c:1540-1718)

Outputs one small .npz per case (numpy arrays only; loadable with allow_pickle=False).
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

OUT = Path(__file__).resolve().parent
TYPES = [oracle.Q4_0, oracle.Q4_1, oracle.Q5_0, oracle.Q5_1, oracle.Q8_0, oracle.Q2_K, oracle.Q3_K,
         oracle.Q4_K, oracle.Q5_K, oracle.Q6_K, oracle.IQ4_NL, oracle.IQ4_XS,
         oracle.IQ2_XXS, oracle.IQ2_XS, oracle.IQ2_S, oracle.IQ3_XXS, oracle.IQ3_S, oracle.IQ1_S, oracle.IQ1_M]


def main() -> None:
    ref = oracle.Reference("scalar")
    for t in TYPES:
        name = ref.type_name(t)
        rng = np.random.default_rng(1000 + t)
        # (a) test-backend-ops' smallest shape: m=16, k=256, n=1..3   (tests/test-backend-ops.cpp:4143-4147)
        # (b) one wider case with several super-blocks per row
        for tag, (M, N, K) in {"m16n3k256": (16, 3, 256), "m24n2k1024": (24, 2, 1024)}.items():
            wf = rng.uniform(-1, 1, (M, K)).astype(np.float32)
            x = rng.uniform(-1, 1, (N, K)).astype(np.float32)
            if tag == "m24n2k1024":
                x[1, 256:512] = 0.0          # an all-zero activation super-block (Q8_K d == 0 path)
                x[0, 5] = -x[0].max() * 3    # negative element of largest magnitude (Q8_K sign rule)
            w = ref.quantize(t, wf)
            act_t = ref.vec_dot_type(t)
            np.savez_compressed(
                OUT / f"mul_mat_{name}_{tag}.npz",
                type=np.int32(t), act_type=np.int32(act_t), M=np.int32(M), N=np.int32(N), K=np.int32(K),
                w=w, x=x, act=ref.quantize_act(act_t, x, cpu_path=True),
                deq=ref.dequantize(t, w, K), y=ref.mul_mat(t, w, x, M, N, K))
        # (c) MoE: 4 experts, 2 used, 3 tokens, shared activations (b_ne1 = 1) and per-slot (b_ne1 = n_used)
        M, K, n_exp, n_used, n_tok = 16, 256, 4, 2, 3
        as_ = ref.quantize(t, rng.uniform(-1, 1, (n_exp * M, K)).astype(np.float32))
        ids = np.stack([rng.permutation(n_exp)[:n_used] for _ in range(n_tok)]).astype(np.int32)
        for b_ne1 in (1, n_used):
            b = rng.uniform(-1, 1, (n_tok, b_ne1, K)).astype(np.float32)
            np.savez_compressed(
                OUT / f"mul_mat_id_{name}_b{b_ne1}.npz",
                type=np.int32(t), M=np.int32(M), K=np.int32(K), n_expert=np.int32(n_exp),
                as_=as_, b=b, ids=ids, y=ref.mul_mat_id(t, as_, b, ids, M, K, n_exp))
    print("wrote", len(list(OUT.glob("*.npz"))), "fixtures to", OUT)


if __name__ == "__main__":
    main()

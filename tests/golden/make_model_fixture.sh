#!/bin/sh
# Generates tests/golden/model_logits_small_l4.bin: the logits of the reference ggml CPU backend (oracle/_ref/avx2, built from
# /root/reference by oracle/Makefile) for the synthetic 4-layer llama model of oracle/model_parity/model_parity.cc (n_embd 2048, n_ff 4096,
# 16/4 heads, n_vocab 32000, seeded weights quantized by the reference's own quantizer), 16 decode steps from an empty context, sampled at
# every 17th vocabulary position.  tests/test_plugin.py::test_whole_model_logits_against_cpu_fixture compares the plugin's logits with it.
# The *_scalar.bin twins come from the reference's scalar build (oracle/_ref/scalar: every vec_dot takes its ISA-independent branch): the two
# builds of the reference disagree by ~2e-4 NMSE on this model (a chaotic map once activations are re-quantized per matmul), which is the
# resolution any logits comparison has; the 1-layer fixture stays below the north-star bound between the builds.
set -e
cd "$(dirname "$0")/../.."
for v in avx2 scalar; do
  sfx=""; [ $v = scalar ] && sfx="_scalar"
  oracle/_ref/$v/model_parity --preset small --layers 4 --vocab 32000 --tokens 16 --dump tests/golden/model_logits_small_l4$sfx.bin
  oracle/_ref/$v/model_parity --preset small --layers 4 --vocab 32000 --tokens 16 --fa --dump tests/golden/model_logits_small_l4_fa$sfx.bin
  oracle/_ref/$v/model_parity --preset small --layers 1 --vocab 32000 --tokens 16 --dump tests/golden/model_logits_small_l1$sfx.bin
done

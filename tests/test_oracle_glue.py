"""The numpy restatement of the residency ops (oracle/glue.py) against the REAL reference CPU backend (oracle/_ref through
refshim ref_glue_op) at Llama decode shapes.  Elementwise / norm ops must agree bit for bit with the scalar reference build;
ops with expf / sinf / cosf or a long f32 dot product within a few ulp."""
import numpy as np
import pytest

import oracle
from oracle import glue

pytestmark = pytest.mark.skipif(not oracle.ref_available("scalar"), reason="oracle/_ref/scalar not built")


@pytest.fixture(scope="module")
def ref():
    r = oracle.Reference("scalar")
    if not hasattr(r.lib, "ref_glue_op"):
        pytest.skip("refshim without ref_glue_op (stale oracle/_ref)")
    return r


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_bin_bcast_bitexact(ref):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((2, 3, 5, 64)).astype(np.float32)
    for shape in ((2, 3, 5, 64), (1, 1, 1, 64), (1, 3, 1, 64), (1, 1, 5, 1)):
        b = rng.uniform(0.5, 2.0, shape).astype(np.float32)
        for code, name in ((1, "add"), (2, "sub"), (3, "mul"), (4, "div")):
            assert np.array_equal(bits(glue.bin_bcast(name, a, b)), bits(ref.glue_op(code, a, b))), (name, shape)


def test_rms_norm_bitexact(ref):
    rng = np.random.default_rng(2)
    for shape, eps in (((1, 1, 7, 4096), 1e-5), ((2, 3, 5, 64), 1e-6), ((1, 1, 1, 14336), 1e-5), ((1, 1, 3, 100), 0.0)):
        x = (rng.standard_normal(shape) * rng.uniform(0.01, 30.0)).astype(np.float32)
        assert np.array_equal(bits(glue.rms_norm(x, eps)), bits(ref.glue_op(10, x, fparams=[eps]))), shape


def test_silu_soft_max_close(ref):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((1, 1, 4, 14336)) * 4).astype(np.float32)
    r = ref.glue_op(11, x)
    assert np.abs(glue.silu(x) - r).max() <= 4e-7 * max(1.0, np.abs(r).max())
    kq = rng.standard_normal((1, 32, 3, 257)).astype(np.float32) * 3
    mask = np.where(rng.random((3, 257)) < 0.2, -np.inf, 0.0).astype(np.float32); mask[:, 0] = 0
    for mb in (0.0, 8.0):
        r = ref.glue_op(12, kq, mask, fparams=[0.0884, mb])
        o = glue.soft_max(kq, mask, 0.0884, mb)
        assert np.abs(o - r).max() <= 2e-7 and np.allclose(o.sum(-1), 1.0, atol=1e-5)
    r = ref.glue_op(12, kq, None, fparams=[1.0, 0.0])
    assert np.abs(glue.soft_max(kq, None, 1.0) - r).max() <= 2e-7


def test_rope_close(ref):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((1, 5, 8, 128)).astype(np.float32)
    pos = np.array([0, 1, 17, 1000, 8191], np.int32)
    ff = rng.uniform(1.0, 8.0, 64).astype(np.float32)
    for mode in (0, 2):
        for n_dims, freq, fscale, ext in ((128, None, 1.0, 0.0), (64, None, 1.0, 0.0), (128, ff, 1.0, 0.0), (128, None, 0.25, 1.0)):
            r = ref.glue_op(13, x, freq[:n_dims // 2] if freq is not None else None, pos=pos,
                            fparams=[500000.0, fscale, ext, 1.0, 32.0, 1.0], iparams=[n_dims, mode, 8192])
            o = glue.rope(x, pos, n_dims, mode, freq, 8192, 500000.0, fscale, ext, 1.0, 32.0, 1.0)
            assert np.abs(o - r).max() <= 3e-5 * np.abs(r).max(), (mode, n_dims, fscale, ext)      # sinf/cosf of angles up to ~8e3 rad


def test_mul_mat_f_close(ref):
    rng = np.random.default_rng(5)
    k = rng.standard_normal((1, 8, 300, 128)).astype(np.float32)          # K cache view: [n_head_kv, n_kv, head_dim]
    q = rng.standard_normal((1, 32, 2, 128)).astype(np.float32)           # [n_head, n_tokens, head_dim]
    for code, f16 in ((14, True), (15, False)):
        r = ref.glue_op(code, k, q, out_shape=(1, 32, 2, 300))
        o = glue.mul_mat_f(k, q, f16)
        assert np.abs(o - r).max() <= 2e-5 * np.abs(r).max(), code


_FLASH_PIN_SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import oracle
from oracle import glue
variant, cfg = sys.argv[2], json.loads(sys.argv[3])
r = oracle.Reference(variant)
H, Hk, N, DK, n_kv, max_bias, softcap = cfg
rng = np.random.default_rng(H * 1000 + DK + n_kv)
q = rng.standard_normal((1, H, N, DK)).astype(np.float32)
k = rng.standard_normal((1, Hk, n_kv, DK)).astype(np.float16); v = rng.standard_normal((1, Hk, n_kv, DK)).astype(np.float16)
mask = np.zeros((64, n_kv), np.float16)                          # (ggml pads the mask rows to GGML_KQ_MASK_PAD)
for t in range(N):
    mask[t, n_kv - N + t + 1 - 3:] = -np.inf                     # causal, and the last three positions never written
mask[:, 5] = -np.inf                                             # a hole in the middle: skipped, not weighted by zero
if max_bias > 0:
    mask[:N] += (rng.standard_normal((N, n_kv)) * 0.1).astype(np.float16)
if len(sys.argv) > 4 and sys.argv[4] in ("q8_0", "q4_0"):        # the same on a Q8_0 / Q4_0 K / V cache
    want, kb, vb = r.flash_attn_ext_q8_0(q, k.astype(np.float32), v.astype(np.float32), mask, 1.0 / np.sqrt(DK), max_bias, softcap,
                                         kv_type=oracle.Q8_0 if sys.argv[4] == "q8_0" else oracle.Q4_0)
    got = glue.flash_attn_ext_q8_0(q, kb, vb, mask, 1.0 / np.sqrt(DK), max_bias, softcap, kv=sys.argv[4])
else:
    got = glue.flash_attn_ext(q, k, v, mask, 1.0 / np.sqrt(DK), max_bias, softcap)
    want = r.flash_attn_ext(q, k, v, mask, 1.0 / np.sqrt(DK), max_bias, softcap)
d = np.abs(got.astype(np.float64) - want); top = float(np.abs(want).max())
print(json.dumps({"differ": float((got.view(np.uint32) != want.view(np.uint32)).mean()), "max_rel": float(d.max() / top), "flips": float((d > 1e-6 * top).mean())}))
"""


@pytest.mark.parametrize("variant", ["avx2", "scalar"])
@pytest.mark.parametrize("cfg", [(8, 2, 1, 128, 96, 0.0, 0.0), (4, 4, 3, 64, 40, 0.0, 0.0), (8, 2, 2, 128, 256, 4.0, 0.0), (4, 1, 1, 96, 64, 0.0, 10.0), (2, 2, 1, 256, 33, 0.0, 0.0)], ids=str)
def test_flash_attn_ext_bitexact(variant, cfg):
    """FLASH_ATTN_EXT with an F16 cache: the restatement (positions in order, running maximum, F16 accumulator, ggml_vec_dot_f16's SIMD order,
    the C library's expf) against the real reference CPU backend: decode and small prefill batches, GQA, ALiBi, soft-capping, ragged n_kv.
    BIT FOR BIT against the AVX2 build -- the one llama-bench and model_parity run; the scalar build sums the dot products in f64 and S without fma:
    its outputs sit in the neighbouring f32 and, once in a few hundred elements, on the other side of an f16 rounding of the accumulator.
    Each variant runs in a process of its own: the two builds export the same sonames, and a process that has loaded one resolves the other's to it."""
    import json, subprocess, sys
    from pathlib import Path
    if not oracle.ref_available(variant):
        pytest.skip(f"oracle/_ref/{variant} not built")
    root = str(Path(__file__).resolve().parents[1])
    pr = subprocess.run([sys.executable, "-c", _FLASH_PIN_SCRIPT, root, variant, json.dumps(list(cfg))], capture_output=True, text=True, timeout=300)
    if "ref_flash_attn_ext" in pr.stderr and "AttributeError" in pr.stderr:
        pytest.skip("refshim without ref_flash_attn_ext (stale oracle/_ref)")
    assert pr.returncode == 0, pr.stderr[-1500:]
    st = json.loads(pr.stdout.strip().splitlines()[-1])
    if variant == "avx2":
        assert st["differ"] == 0.0, st
    else:
        assert st["max_rel"] <= 2e-3 and st["flips"] <= 0.02, st


@pytest.mark.parametrize("kv", ["q8_0", "q4_0"])
@pytest.mark.parametrize("cfg", [(8, 2, 1, 128, 96, 0.0, 0.0), (4, 4, 3, 64, 40, 0.0, 0.0), (8, 2, 2, 128, 256, 4.0, 0.0), (4, 1, 1, 96, 64, 0.0, 10.0)], ids=str)
def test_flash_attn_ext_q8_0_cache_bitexact(cfg, kv):
    """FLASH_ATTN_EXT on a Q8_0 / Q4_0 K / V cache: q quantized to Q8_0 by the SIMD quantizer, ggml_vec_dot_q8_0_q8_0's AVX2 lane order, the online softmax in
    order, V dequantized into an F32 accumulator -- the restatement against the reference's AVX2 build, bit for bit (own process: see above)."""
    import json, subprocess, sys
    from pathlib import Path
    if not oracle.ref_available("avx2"):
        pytest.skip("oracle/_ref/avx2 not built")
    root = str(Path(__file__).resolve().parents[1])
    pr = subprocess.run([sys.executable, "-c", _FLASH_PIN_SCRIPT, root, "avx2", json.dumps(list(cfg)), kv], capture_output=True, text=True, timeout=300)
    if "ref_flash_attn_ext_t" in pr.stderr and "AttributeError" in pr.stderr:
        pytest.skip("refshim without ref_flash_attn_ext_t (stale oracle/_ref)")
    assert pr.returncode == 0, pr.stderr[-1500:]
    st = json.loads(pr.stdout.strip().splitlines()[-1])
    assert st["differ"] == 0.0, st

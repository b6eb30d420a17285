"""Parity tests proper: the HIP path, called through the C-ABI (libmi355q.so via ggml_mi355), against the
oracle on the same seeded inputs, against the committed golden vectors, and through size-independent
properties at BASELINE.json's full sizes.

Tolerances (stated once):
  * integer work -- activation blocks, block decode round trips, Q8_0/Q4_K/Q6_K on exactly-representable
    inputs: BIT-EXACT.
  * GEMV tier (N <= 8, and every non-planar shape): the per-block integer sums are identical to the CPU's; only
    the order of the final f32 additions differs, so  |y - y_oracle| <= 1e-5 * max|y_oracle|  (about 10x the
    observed error) and, element-wise, the north-star bound |y - y_oracle| <= 1e-3*|y_oracle| + 1e-5*max|y_oracle|.
  * MFMA tier (N > 8 on planar rows): bf16 operands, f32 accumulate, activations NOT Q8-quantized.  Against the
    oracle (which carries the CPU's Q8 activation quantization noise) the reference's own op bound applies:
    NMSE <= 5e-4 (tests/test-backend-ops.cpp:1990-1992; observed ~1e-5).  Against the exact product of the
    dequantized weights and the f32 activations in float64: NMSE <= 2e-5 (bf16 rounding of both operands).
"""
import numpy as np
import pytest

import oracle
from conftest import GOLDEN
from qdata import quantized_weights, random_blocks

pytestmark = pytest.mark.gpu

TYPES = [oracle.Q4_0, oracle.Q4_1, oracle.Q5_0, oracle.Q5_1, oracle.Q8_0, oracle.Q2_K, oracle.Q3_K,
         oracle.Q4_K, oracle.Q5_K, oracle.Q6_K, oracle.IQ4_NL, oracle.IQ4_XS,
         oracle.IQ2_XXS, oracle.IQ2_XS, oracle.IQ2_S, oracle.IQ3_XXS, oracle.IQ3_S, oracle.IQ1_S, oracle.IQ1_M]
FAST = [oracle.Q4_K, oracle.Q5_K, oracle.Q6_K, oracle.Q8_0, oracle.Q4_0]
ids_t = lambda t: oracle.TYPE_NAMES[t]


@pytest.fixture(scope="module")
def G():
    import torch
    import ggml_mi355 as g
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    assert g.lib().mi355q_device_count() >= 1, "libmi355q.so found no gfx950 device"
    return g


@pytest.fixture(scope="module")
def torch():
    import torch as t
    return t


def check_close(y, ref, what=""):
    y = np.asarray(y, np.float32); ref = np.asarray(ref, np.float32)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    assert np.isfinite(y).all(), what
    scale = float(np.abs(ref).max()) or 1.0
    err = np.abs(y - ref)
    assert err.max() <= 1e-5 * scale, f"{what}: max err {err.max():.3e} vs scale {scale:.3e}"
    assert (err <= 1e-3 * np.abs(ref) + 1e-5 * scale).all(), what


def nmse(y, ref):
    y = np.asarray(y, np.float64); ref = np.asarray(ref, np.float64)
    return float(((y - ref) ** 2).sum() / max((ref ** 2).sum(), 1e-300))


MMQ_TYPES = (oracle.Q4_K, oracle.Q5_K, oracle.Q6_K, oracle.Q8_0, oracle.Q4_0, oracle.IQ4_NL, oracle.IQ4_XS)


BLOCK32_TYPES = (oracle.Q4_0, oracle.Q4_1, oracle.Q5_0, oracle.Q5_1, oracle.Q8_0, oracle.IQ4_NL)
CANONICAL_BATCH_TYPES = BLOCK32_TYPES + (oracle.Q2_K, oracle.Q3_K, oracle.IQ2_XXS, oracle.IQ2_XS, oracle.IQ2_S, oracle.IQ3_XXS, oracle.IQ3_S, oracle.IQ1_S, oracle.IQ1_M)


def on_mfma_tier(G, t, K, N):
    """Mirror of the library's tier choice (csrc/api.hip): N > 8 and either planar rows of a type with a matrix-core kernel (K a multiple of its
    MFMA step) or canonical rows of a type the batched canonical tier decodes (csrc/mmq_generic.hip; K a multiple of 128 / of the 256-block)."""
    if N <= 8:
        return False
    if G.is_planar(t, K):
        return t in MMQ_TYPES and K % (256 if t in (oracle.Q4_K, oracle.Q5_K, oracle.Q6_K, oracle.IQ4_XS) else 128) == 0
    return t in CANONICAL_BATCH_TYPES and K % (128 if t in BLOCK32_TYPES else 256) == 0


def check(G, t, K, N, y, ref, what=""):
    if on_mfma_tier(G, t, K, N):
        assert np.isfinite(y).all(), what
        e = nmse(y, ref)
        assert e <= 5e-4, f"{what}: NMSE {e:.3e}"
    else:
        check_close(y, ref, what)


def gpu_mul_mat(G, torch, t, w_rows, x, M, K, flags=0):
    w = G.QWeight.from_host(t, w_rows, M, K)
    y = G.mul_mat(w, torch.from_numpy(np.ascontiguousarray(x)).cuda(), flags=flags)
    torch.cuda.synchronize()
    return y.cpu().numpy()


# ------------------------------------------------------------------------------------------------
# golden vectors produced by the real reference
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", sorted(GOLDEN.glob("mul_mat_[!i]*.npz")) + sorted(GOLDEN.glob("mul_mat_iq*.npz")), ids=lambda p: p.stem)
def test_golden_mul_mat(G, torch, path):
    g = np.load(path, allow_pickle=False)
    t, M, N, K = int(g["type"]), int(g["M"]), int(g["N"]), int(g["K"])
    check(G, t, K, N, gpu_mul_mat(G, torch, t, g["w"], g["x"], M, K), g["y"], path.stem)


@pytest.mark.parametrize("path", sorted(GOLDEN.glob("mul_mat_id_*.npz")), ids=lambda p: p.stem)
def test_golden_mul_mat_id(G, torch, path):
    g = np.load(path, allow_pickle=False)
    t, M, K, ne = int(g["type"]), int(g["M"]), int(g["K"]), int(g["n_expert"])
    w = G.QWeight.from_host(t, g["as_"], M, K, n_expert=ne)
    y = G.mul_mat_id(w, torch.from_numpy(g["b"]).cuda(), torch.from_numpy(g["ids"]).cuda())
    check_close(y.cpu().numpy(), g["y"], path.stem)


# ------------------------------------------------------------------------------------------------
# activation quantizer: bit-exact blocks
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("act", [oracle.Q8_0, oracle.Q8_1, oracle.Q8_K], ids=ids_t)
@pytest.mark.parametrize("mode", [oracle.ROUND_AWAY, oracle.ROUND_EVEN])
def test_quantize_act_bitexact(G, torch, orc, act, mode):
    rng = np.random.default_rng(11 + act)
    K = 4096 if act == oracle.Q8_K else 4096 + 32        # ragged tail for the 32-element formats
    x = rng.standard_normal((5, K)).astype(np.float32)
    x[0, :256] = 0.0                                      # all-zero block
    x[1, 7] = 100.0; x[1, 200] = -100.0                   # equal |max| of both signs: first one wins (Q8_K)
    x[2, :32] = np.arange(32) * 0.5 + 0.25                # exact .5 ties after scaling (d = 15.75/127 not exact, still a stress)
    x[3, :32] = 0; x[3, 0] = 127.0; x[3, 1] = 2.5; x[3, 2] = -3.5   # d = 1: genuine rounding ties
    got = G.quantize_act(act, torch.from_numpy(x).cuda(), flags=mode).cpu().numpy()
    want = orc.quantize_act(act, x, mode)
    if act == oracle.Q8_K:                                # bsums of all-zero blocks are undefined in the reference
        g2 = got.reshape(5, -1, 292).copy(); w2 = want.reshape(5, -1, 292).copy()
        z = w2[:, :, :4].view(np.float32)[..., 0] == 0
        g2[z, 260:] = 0; w2[z, 260:] = 0
        assert np.array_equal(g2, w2)
    else:
        assert np.array_equal(got, want)


# ------------------------------------------------------------------------------------------------
# every type x batch sizes at the reference's test-backend-ops shapes (tests/test-backend-ops.cpp:4143-4147)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("t", TYPES, ids=ids_t)
@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 8, 9, 17])
def test_small_shapes(G, torch, orc, t, N):
    rng = np.random.default_rng(100 * t + N)
    M, K = 16, 256
    w = quantized_weights(t, M, K, rng)
    x = rng.uniform(-1, 1, (N, K)).astype(np.float32)
    check(G, t, K, N, gpu_mul_mat(G, torch, t, w, x, M, K), orc.mul_mat(t, w, x, M, N, K), f"{ids_t(t)} N={N}")


@pytest.mark.parametrize("t", TYPES, ids=ids_t)
@pytest.mark.parametrize("shape", [(64, 2048), (33, 4096), (7, 14336), (130, 512), (257, 6144)], ids=str)
def test_planar_and_odd_shapes(G, torch, orc, t, shape):
    """K multiples of 2048 take the planar fast tier for the fast types; K=512 keeps 2-byte-aligned
    formats on the generic tier; ragged M exercises the row tails."""
    M, K = shape
    rng = np.random.default_rng(7 * t + M)
    w = quantized_weights(t, M, K, rng)
    for N in (1, 3, 8):
        x = (rng.standard_normal((N, K)) * rng.uniform(0.1, 4.0)).astype(np.float32)
        check_close(gpu_mul_mat(G, torch, t, w, x, M, K), orc.mul_mat(t, w, x, M, N, K), f"{ids_t(t)} {shape} N={N}")


@pytest.mark.parametrize("t", TYPES, ids=ids_t)
def test_random_byte_blocks(G, torch, orc, t):
    """Arbitrary bit patterns in the quant payload (all 6-bit scales, all nibble values, sign bits...)."""
    rng = np.random.default_rng(31 + t)
    M, K = 48, 2048
    w = random_blocks(t, M, K, rng)
    x = rng.standard_normal((2, K)).astype(np.float32)
    check_close(gpu_mul_mat(G, torch, t, w, x, M, K), orc.mul_mat(t, w, x, M, 2, K), ids_t(t))


@pytest.mark.parametrize("t", TYPES, ids=ids_t)
def test_upload_download_roundtrip(G, torch, t):
    """set_tensor -> device layout -> get_tensor is the identity on canonical bytes."""
    rng = np.random.default_rng(5 + t)
    for M, K in ((3, 256), (16, 2048), (5, 14336)):
        w = random_blocks(t, M, K, rng)
        back = G.QWeight.from_host(t, w, M, K).to_host()
        assert np.array_equal(back, w), (ids_t(t), M, K)


def test_edge_cases(G, torch, orc):
    t = oracle.Q4_K
    rng = np.random.default_rng(1)
    w = quantized_weights(t, 8, 256, rng)
    wq = G.QWeight.from_host(t, w, 8, 256)
    # empty batch
    y = G.mul_mat(wq, torch.empty((0, 256), dtype=torch.float32, device="cuda"))
    assert tuple(y.shape) == (0, 8)
    # all-zero activations -> exactly zero output
    y = G.mul_mat(wq, torch.zeros((2, 256), dtype=torch.float32, device="cuda")).cpu().numpy()
    assert np.array_equal(y, np.zeros((2, 8), np.float32))
    # strided (non-contiguous rows) activations
    xs = torch.from_numpy(rng.standard_normal((3, 512)).astype(np.float32)).cuda()
    xv = xs[:, :256]
    check_close(G.mul_mat(wq, xv).cpu().numpy(), orc.mul_mat(t, w, xv.cpu().numpy().copy(), 8, 3, 256))
    # bad shapes are reported, not executed
    with pytest.raises(G.Mi355qError):
        G.QWeight.from_host(t, w[:, :100], 8, 256)
    with pytest.raises(G.Mi355qError):
        G.QWeight.from_host(oracle.IQ2_XS, w, 8, 256)    # a Q4_K-sized row handed over as IQ2_XS (74-byte blocks): size mismatch is reported
    with pytest.raises(G.Mi355qError):
        G.QWeight.from_host(oracle.Q8_K, np.zeros((8, 292), np.uint8), 8, 256)      # right size, but an activation-only format is not a weight type: loud error, no fallback
    with pytest.raises(G.Mi355qError):
        G.QWeight.from_host(31, w, 8, 256)               # a ggml type id the library does not know at all


# ------------------------------------------------------------------------------------------------
# bit-exact on exactly representable inputs ("bit-exact for Q8_0 integer dot", BASELINE.json north_star)
# ------------------------------------------------------------------------------------------------
def test_q8_0_integer_dot_bitexact(G, torch, orc):
    rng = np.random.default_rng(2)
    M, K, N = 64, 4096, 3
    nb = K // 32
    w = np.zeros((M, nb, 34), np.uint8)
    w[:, :, 0:2] = np.float16(2.0 ** -6).view(np.uint8) if False else np.frombuffer(np.float16(2.0 ** -6).tobytes(), np.uint8)
    w[:, :, 2:] = rng.integers(-7, 8, (M, nb, 32), dtype=np.int8).view(np.uint8)
    x = rng.integers(-126, 127, (N, K)).astype(np.float32)
    x.reshape(N, nb, 32)[:, :, 0] = 127.0                 # amax = 127 -> d = 1, quants == x exactly
    w = w.reshape(M, nb * 34)
    ref = orc.mul_mat(oracle.Q8_0, w, x, M, N, K)
    for flags in (oracle.ROUND_AWAY, oracle.ROUND_EVEN):
        y = gpu_mul_mat(G, torch, oracle.Q8_0, w, x, M, K, flags=flags)
        assert np.array_equal(y.view(np.uint32), ref.view(np.uint32))
    # and the plain integer answer
    wi = w.reshape(M, nb, 34)[:, :, 2:].view(np.int8).reshape(M, K).astype(np.int64)
    exact = (x.astype(np.int64) @ wi.T).astype(np.float64) * 2.0 ** -6
    assert np.array_equal(ref.astype(np.float64), exact)


@pytest.mark.parametrize("t", [oracle.Q4_K, oracle.Q6_K], ids=ids_t)
def test_kquant_bitexact_on_exact_inputs(G, torch, orc, t):
    rng = np.random.default_rng(3)
    M, K, N = 32, 2048, 2
    nb = K // 256
    if t == oracle.Q4_K:
        w = np.zeros((M, nb, 144), np.uint8)
        w[:, :, 0:2] = np.frombuffer(np.float16(2.0 ** -8).tobytes(), np.uint8)
        w[:, :, 2:4] = np.frombuffer(np.float16(2.0 ** -9).tobytes(), np.uint8)
        w[:, :, 4:12] = rng.integers(0, 8, (M, nb, 8), dtype=np.uint8)      # sc/min < 8, high bits clear
        w[:, :, 12:16] = rng.integers(0, 8, (M, nb, 4), dtype=np.uint8) * 17 & 0x77
        w[:, :, 16:] = rng.integers(0, 256, (M, nb, 128), dtype=np.uint8)
    else:
        w = np.zeros((M, nb, 210), np.uint8)
        w[:, :, :192] = rng.integers(0, 256, (M, nb, 192), dtype=np.uint8)
        w[:, :, 192:208] = rng.integers(-4, 5, (M, nb, 16), dtype=np.int8).view(np.uint8)
        w[:, :, 208:210] = np.frombuffer(np.float16(2.0 ** -7).tobytes(), np.uint8)
    w = w.reshape(M, -1)
    x = rng.integers(-15, 16, (N, K)).astype(np.float32)
    x.reshape(N, nb, 256)[:, :, 3] = -127.0                # iscale = -127/max = 1, d = 1: quants == x exactly
    ref = orc.mul_mat(t, w, x, M, N, K)
    y = gpu_mul_mat(G, torch, t, w, x, M, K)
    assert np.array_equal(y.view(np.uint32), ref.view(np.uint32))


# ------------------------------------------------------------------------------------------------
# MFMA (prefill) tier: N > 8
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("t", MMQ_TYPES, ids=ids_t)
@pytest.mark.parametrize("shape", [(128, 9, 2048), (256, 64, 4096), (200, 130, 2048), (1024, 512, 4096), (96, 33, 14336),
                                   (12288, 512, 2048)],       # (the last one is large enough for the 128 x 128-tile kernels)
                         ids=str)
def test_mfma_tier(G, torch, orc, t, shape):
    M, N, K = shape
    rng = np.random.default_rng(M + N + K + t)
    w = quantized_weights(t, M, K, rng)
    x = (rng.standard_normal((N, K)) * rng.uniform(0.2, 3.0)).astype(np.float32)
    assert on_mfma_tier(G, t, K, N)
    y = gpu_mul_mat(G, torch, t, w, x, M, K)
    assert np.isfinite(y).all()
    # (1) the reference's op bound against the CPU arithmetic (oracle); sampled rows keep the scalar oracle quick
    rows = np.unique(rng.integers(0, M, 48)); cols = np.unique(rng.integers(0, N, 12))
    ref = orc.mul_mat(t, w[rows], x[cols], len(rows), len(cols), K)
    assert nmse(y[np.ix_(cols, rows)], ref) <= 5e-4
    # (2) against the exact product of the dequantized weights and the f32 activations (float64)
    exact = x.astype(np.float64) @ orc.dequantize(t, w, K).astype(np.float64).T
    e = nmse(y, exact)
    if t in (oracle.Q4_K, oracle.Q8_0):
        # integer tiers (csrc/mmq_i8.hip, csrc/mmq_q80.hip): the CPU's own arithmetic -- Q8_K / Q8_0 activations, exact integer block sums --
        # so they match the oracle to f32 summation order (Q8_0: bit for bit, test_q8_0_prefill_is_bit_exact), and their distance from the
        # exact product is the CPU's own activation quantization noise (~5e-5 / ~3e-5)
        check_close(y[np.ix_(cols, rows)], ref)
        if t == oracle.Q8_0 and K % 256 == 0:
            assert np.array_equal(y[np.ix_(cols, rows)].view(np.uint32), ref.view(np.uint32))
        assert e <= 1.5e-4, f"NMSE vs exact {e:.3e}"
        e_cpu = nmse(ref, exact[np.ix_(cols, rows)])
        assert e <= 3 * e_cpu + 1e-6, f"NMSE vs exact {e:.3e}, the CPU arithmetic itself {e_cpu:.3e}"
    else:
        assert e <= 2e-5, f"NMSE vs exact {e:.3e}"
    # (3) the GEMV tier forced on the same data agrees with the oracle to f32 summation order
    if N <= 64:
        yg = gpu_mul_mat(G, torch, t, w, x, M, K, flags=0x4)          # MI355Q_FLAG_FORCE_GEMV
        check_close(yg[np.ix_(cols, rows)], ref)


@pytest.mark.parametrize("shape", [(128, 9, 2048), (200, 130, 2048), (1024, 512, 4096), (96, 33, 14336), (12288, 160, 1024)], ids=str)
@pytest.mark.parametrize("round_even", [False, True], ids=["round_away", "round_even"])
def test_q8_0_prefill_is_bit_exact(G, torch, orc, shape, round_even):
    """north_star: "bit-exact for Q8_0 integer dot".  At N > 8 the Q8_0 tier (csrc/mmq_q80.hip) takes the exact int32 block sums from the integer
    matrix cores and applies the CPU's two f32 operations per block in the CPU's block order (ggml_vec_dot_q8_0_q8_0 scalar tail), so every
    output equals the scalar CPU backend bit for bit -- for both activation rounding rules (roundf: quantize_row_q8_0_ref; rint: the AVX2 path)."""
    M, N, K = shape
    t = oracle.Q8_0
    rng = np.random.default_rng(M + N + K)
    w = quantized_weights(t, M, K, rng)
    x = (rng.standard_normal((N, K)) * rng.uniform(0.2, 3.0)).astype(np.float32)
    x[N // 2, 32:64] = 0.0                                  # an all-zero activation block: d = 0, id = 0
    x[0, :32] = np.arange(32, dtype=np.float32) - 15.5      # exact .5 ties after scaling: the two rounding rules differ here
    y = gpu_mul_mat(G, torch, t, w, x, M, K, flags=0x1 if round_even else 0x0)
    rows = np.unique(np.concatenate([rng.integers(0, M, 96), [0, M - 1]])); cols = np.unique(np.concatenate([rng.integers(0, N, 24), [0, N - 1, N // 2]]))
    ref = orc.mul_mat(t, w[rows], x[cols], len(rows), len(cols), K, round_mode=oracle.ROUND_EVEN if round_even else oracle.ROUND_AWAY)
    assert np.array_equal(y[np.ix_(cols, rows)].view(np.uint32), ref.view(np.uint32))


def test_q8_0_prefill_multi_and_unaligned_fall_back(G, torch, orc):
    """QKV-style multi launch shares one quantized activation image and stays bit-exact; K % 256 != 0 or an unaligned output falls to other tiers."""
    t = oracle.Q8_0
    rng = np.random.default_rng(80)
    K, N = 4096, 77
    specs = [96, 40, 24]
    hosts = [quantized_weights(t, m, K, rng) for m in specs]
    ws = [G.QWeight.from_host(t, h, m, K) for m, h in zip(specs, hosts)]
    x = rng.standard_normal((N, K)).astype(np.float32)
    for ym, h, m in zip(G.mul_mat_multi(ws, torch.from_numpy(x).cuda()), hosts, specs):
        assert np.array_equal(ym.cpu().numpy().view(np.uint32), orc.mul_mat(t, h, x, m, N, K).view(np.uint32))
    K = 384                                                # 12 blocks: not a multiple of the 256-k step
    w = quantized_weights(t, 64, K, rng); x = rng.standard_normal((20, K)).astype(np.float32)
    check(G, t, K, 20, gpu_mul_mat(G, torch, t, w, x, 64, K), orc.mul_mat(t, w, x, 64, 20, K), "K=384")


def test_q6_k_wide_token_tile(G, torch, orc):
    """Q6_K batches of >= 256 tokens run 128 x 256 tiles (8 waves; csrc/mmq_bf16.hip) when those still cover the chip.  An output element's
    k-order does not depend on the tile shape, so the first 200 tokens of a ragged 300-token batch must equal the same tokens multiplied alone
    (200 < 256: the 128-token tile) bit for bit; and the usual bounds against the oracle and the exact product hold."""
    t, M, K = oracle.Q6_K, 16384, 2048
    rng = np.random.default_rng(77)
    w = quantized_weights(t, M, K, rng)
    x = (rng.standard_normal((300, K)) * 1.7).astype(np.float32)
    assert on_mfma_tier(G, t, K, 300)
    wq = G.QWeight.from_host(t, w, M, K)
    y = G.mul_mat(wq, torch.from_numpy(x).cuda()).cpu().numpy()
    y200 = G.mul_mat(wq, torch.from_numpy(x[:200].copy()).cuda()).cpu().numpy()
    assert np.isfinite(y).all()
    assert np.array_equal(y[:200].view(np.uint32), y200.view(np.uint32))
    rows = np.unique(rng.integers(0, M, 40)); cols = np.unique(np.concatenate([rng.integers(0, 300, 10), [255, 256, 299]]))
    ref = orc.mul_mat(t, w[rows], x[cols], len(rows), len(cols), K)
    assert nmse(y[np.ix_(cols, rows)], ref) <= 5e-4
    exact = x.astype(np.float64) @ orc.dequantize(t, w, K).astype(np.float64).T
    assert nmse(y, exact) <= 2e-5


@pytest.mark.parametrize("t", CANONICAL_BATCH_TYPES, ids=ids_t)
def test_canonical_rows_batched_tier(G, torch, orc, t):
    """The 12 types without a planar layout (and the 32-block types at a K that keeps their rows canonical) at batch sizes: every weight is decoded
    once per 64 tokens to the value dequantize_row_<type> gives it, rounded to bf16 and multiplied on the matrix cores (csrc/mmq_generic.hip) --
    NMSE <= 2e-5 against the exact product of the dequantized weights, <= 5e-4 against the CPU arithmetic (the reference's op bound); ragged M and N;
    MI355Q_FLAG_FORCE_GEMV keeps the per-column tier, which reproduces the CPU's integer arithmetic."""
    rng = np.random.default_rng(300 + t)
    for M, N, K in ((200, 70, 384 if t in BLOCK32_TYPES else 512), (64, 9, 128 * 6 if t in BLOCK32_TYPES else 1024)):
        if G.is_planar(t, K):
            continue
        w = quantized_weights(t, M, K, rng)
        x = (rng.standard_normal((N, K)) * 1.3).astype(np.float32)
        assert on_mfma_tier(G, t, K, N)
        y = gpu_mul_mat(G, torch, t, w, x, M, K)
        assert np.isfinite(y).all()
        exact = x.astype(np.float64) @ orc.dequantize(t, w, K).astype(np.float64).T
        e = nmse(y, exact)
        assert e <= 2e-5, f"{ids_t(t)} {M}x{N}x{K}: NMSE vs exact {e:.3e}"
        ref = orc.mul_mat(t, w, x, M, N, K)
        assert nmse(y, ref) <= 5e-4
        yg = gpu_mul_mat(G, torch, t, w, x, M, K, flags=0x4)              # MI355Q_FLAG_FORCE_GEMV
        check_close(yg, ref, f"{ids_t(t)} per-column tier")


def test_mfma_tier_ragged_and_alignment_fallback(G, torch, orc):
    """M not a multiple of the 128-row tile, N not a multiple of the 128-token tile; and an output whose rows are
    not 16-byte aligned (M % 4 != 0) must silently take the GEMV tier (exact) instead of the f32x4 epilogue."""
    t, K = oracle.Q4_K, 2048
    rng = np.random.default_rng(8)
    for M, N in ((130, 129), (4, 40), (257, 10)):
        w = quantized_weights(t, M, K, rng)
        x = rng.standard_normal((N, K)).astype(np.float32)
        y = gpu_mul_mat(G, torch, t, w, x, M, K)
        ref = orc.mul_mat(t, w, x, M, N, K)
        check_close(y, ref, f"ragged {M}x{N}")                         # Q4_K: the integer tier reproduces the CPU arithmetic
    M, N = 131, 20                                                      # 131*4 bytes per row: not 16-byte aligned
    w = quantized_weights(t, M, K, rng); x = rng.standard_normal((N, K)).astype(np.float32)
    check_close(gpu_mul_mat(G, torch, t, w, x, M, K), orc.mul_mat(t, w, x, M, N, K))


# ------------------------------------------------------------------------------------------------
# multi-matrix launch == separate launches (bit-identical), mixed Q4_K/Q6_K as in Q4_K_M attention
# ------------------------------------------------------------------------------------------------
def test_multi_equals_single(G, torch, orc):
    rng = np.random.default_rng(4)
    K = 4096
    specs = [(oracle.Q4_K, 96), (oracle.Q4_K, 40), (oracle.Q6_K, 24)]
    hosts = [quantized_weights(t, m, K, rng) for t, m in specs]
    ws = [G.QWeight.from_host(t, h, m, K) for (t, m), h in zip(specs, hosts)]
    for N in (1, 4, 64):                                   # 64: the matrix-core tiers, which share one prepared copy of the activations per call
        x = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
        multi = [y.cpu().numpy() for y in G.mul_mat_multi(ws, x)]
        for (t, m), h, w, ym in zip(specs, hosts, ws, multi):
            single = G.mul_mat(w, x).cpu().numpy()
            if N <= 8:
                assert np.array_equal(single.view(np.uint32), ym.view(np.uint32))          # the decode contract: bit-identical
            else:
                check_close(ym, single, "multi vs single")        # (a single matrix may be cut along K: another f32 summation order)
            check(G, t, K, N, ym, orc.mul_mat(t, h, x.cpu().numpy(), m, N, K), f"multi N={N}")


# ------------------------------------------------------------------------------------------------
# MoE (tests/test-backend-ops.cpp:4240-4270 shapes: n_mats {4,8} x n_used {1,2,4}, m=512, k=256)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("t", [oracle.Q4_K, oracle.Q6_K, oracle.Q8_0, oracle.Q5_K, oracle.IQ4_XS, oracle.Q4_1, oracle.IQ3_S, oracle.Q2_K], ids=ids_t)
@pytest.mark.parametrize("cfg", [(4, 1, 1), (8, 2, 1), (8, 4, 5), (4, 2, 32), (8, 2, 160)], ids=str)
def test_mul_mat_id(G, torch, orc, t, cfg):
    """(n_expert, n_used, n_tokens).  Up to 16 (token, slot) pairs every pair is one exact GEMV column with the ids read on the device;
    above that the pairs are counting-sorted by expert ON THE DEVICE and one launch of the matrix-core tier walks all experts (csrc/api.hip):
    Q4_K and Q8_0 still reproduce the CPU arithmetic (integer tiers; Q8_0 bit for bit), the bf16 tiers -- planar rows, and canonical rows on the
    batched canonical tier (Q4_1, IQ3_S, Q2_K here) -- are held to the reference's op bound."""
    ne, nu, nt = cfg
    rng = np.random.default_rng(17 * t + ne + nu + nt)
    for K, M in ((256, 512), (2048, 96)):
        as_ = quantized_weights(t, ne * M, K, rng)
        w = G.QWeight.from_host(t, as_, M, K, n_expert=ne)
        ids = np.stack([rng.permutation(ne)[:nu] for _ in range(nt)]).astype(np.int32)
        for b1 in sorted({1, nu}):
            b = rng.uniform(-1, 1, (nt, b1, K)).astype(np.float32)
            y = G.mul_mat_id(w, torch.from_numpy(b).cuda(), torch.from_numpy(ids).cuda()).cpu().numpy()
            ref = orc.mul_mat_id(t, as_, b, ids, M, K, ne)
            # from 17 pairs on the rows are grouped by expert on the device and ALL groups run on the matrix-core tier in one launch
            if nu * nt >= 17 and t not in (oracle.Q4_K, oracle.Q8_0) and on_mfma_tier(G, t, K, 9) and M % 4 == 0:
                assert np.isfinite(y).all() and nmse(y, ref) <= 5e-4, f"{ids_t(t)} {cfg} K={K} b1={b1}"
            else:
                check_close(y, ref, f"{ids_t(t)} {cfg} K={K} b1={b1}")


# ------------------------------------------------------------------------------------------------
# BASELINE.json full sizes (Llama-3-8B): sampled rows against the oracle + size-independent properties
# ------------------------------------------------------------------------------------------------
FULL = [(oracle.Q4_K, 4096, 4096), (oracle.Q4_K, 14336, 4096), (oracle.Q6_K, 4096, 14336),
        (oracle.Q4_K, 4096, 14336), (oracle.Q6_K, 1024, 4096), (oracle.Q8_0, 14336, 4096),
        (oracle.Q5_K, 1024, 8192), (oracle.Q4_0, 4096, 4096),
        # Llama-3-70B Q4_K_M (BASELINE.json configs[3]; SURVEY.md section 8 header: n_embd 8192, n_ff 28672)
        (oracle.Q4_K, 8192, 8192), (oracle.Q4_K, 28672, 8192), (oracle.Q4_K, 8192, 28672), (oracle.Q6_K, 8192, 28672),
        (oracle.Q4_K, 1024, 8192)]


@pytest.mark.parametrize("t,M,K", FULL, ids=lambda v: str(v))
def test_full_size_sampled_rows_and_properties(G, torch, orc, t, M, K):
    rng = np.random.default_rng(M + K + t)
    w = random_blocks(t, M, K, rng)
    wq = G.QWeight.from_host(t, w, M, K)
    x = rng.standard_normal((2, K)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    y = G.mul_mat(wq, xd).cpu().numpy()
    # (1) sampled rows vs the oracle
    rows = np.unique(np.concatenate([[0, 1, M - 1, M - 2], rng.integers(0, M, 60)]))
    ref = orc.mul_mat(t, w[rows], x, len(rows), 2, K)
    check_close(y[:, rows], ref, f"{ids_t(t)} {M}x{K}")
    # (2) N=1 launches give bit-identical columns (column tiles are independent)
    y0 = G.mul_mat(wq, xd[0:1]).cpu().numpy()
    assert np.array_equal(y0.view(np.uint32), y[0:1].view(np.uint32))
    # (3) a row permutation of W permutes y bit-exactly (rows are independent, deterministic)
    perm = rng.permutation(M)
    yp = G.mul_mat(G.QWeight.from_host(t, w[perm], M, K), xd).cpu().numpy()
    assert np.array_equal(yp.view(np.uint32), y[:, perm].view(np.uint32))
    # (4) determinism across repeated launches
    y2 = G.mul_mat(wq, xd).cpu().numpy()
    assert np.array_equal(y2.view(np.uint32), y.view(np.uint32))


@pytest.mark.parametrize("t,M,K", [(oracle.Q4_K, 14336, 4096), (oracle.Q4_K, 4096, 14336), (oracle.Q6_K, 4096, 14336), (oracle.Q6_K, 1024, 4096),
                                   # Llama-3-70B shapes
                                   (oracle.Q4_K, 8192, 8192), (oracle.Q4_K, 28672, 8192), (oracle.Q4_K, 8192, 28672), (oracle.Q6_K, 8192, 28672),
                                   (oracle.Q5_K, 1024, 8192)],
                         ids=lambda v: str(v))
def test_full_size_prefill_sampled_and_properties(G, torch, orc, t, M, K):
    """pp512 at the BASELINE.json sizes (the integer tier with 128x128 tiles, its split-K form, the bf16 tier's split-K form): sampled
    rows / tokens against the oracle, and size-independent properties -- determinism across launches, token rows independent of the other
    tokens of the batch (a token computed in a batch of 512 equals the same token in a batch of 140, up to the f32 summation order where
    the tile shape changes), row permutations of W permute the outputs."""
    N = 512
    rng = np.random.default_rng(M + K + t + 5)
    w = random_blocks(t, M, K, rng)
    wq = G.QWeight.from_host(t, w, M, K)
    x = rng.standard_normal((N, K)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    y = G.mul_mat(wq, xd).cpu().numpy()
    assert np.isfinite(y).all()
    rows = np.unique(np.concatenate([[0, M - 1], rng.integers(0, M, 30)])); toks = np.unique(np.concatenate([[0, N - 1], rng.integers(0, N, 6)]))
    ref = orc.mul_mat(t, w[rows], x[toks], len(rows), len(toks), K)
    if t == oracle.Q4_K:
        check_close(y[np.ix_(toks, rows)], ref, f"{ids_t(t)} {M}x{K} N={N}")          # the CPU arithmetic on the integer tier
    else:
        assert nmse(y[np.ix_(toks, rows)], ref) <= 5e-4
    y2 = G.mul_mat(wq, xd).cpu().numpy()
    assert np.array_equal(y2.view(np.uint32), y.view(np.uint32))                       # deterministic (split-K partial sums are added in fixed order)
    ys = G.mul_mat(wq, xd[:140]).cpu().numpy()                                         # another batch size: other tiles / pieces, same numbers
    assert np.abs(ys - y[:140]).max() <= 2e-5 * np.abs(y).max()
    perm = rng.permutation(M)
    yp = G.mul_mat(G.QWeight.from_host(t, w[perm], M, K), xd).cpu().numpy()
    assert np.abs(yp - y[:, perm]).max() <= 2e-5 * np.abs(y).max()


def test_output_layer_rows(G, torch, orc):
    """The 128256 x 4096 Q6_K output matrix of Llama-3-8B Q4_K_M: maximum row count of the workload."""
    t, M, K = oracle.Q6_K, 128256, 4096
    rng = np.random.default_rng(9)
    base = random_blocks(t, 4096, K, rng)
    w = np.tile(base, (M // 4096 + 1, 1))[:M]
    wq = G.QWeight.from_host(t, w, M, K)
    x = rng.standard_normal((1, K)).astype(np.float32)
    y = G.mul_mat(wq, torch.from_numpy(x).cuda()).cpu().numpy()
    ref = orc.mul_mat(t, base[:256], x, 256, 1, K)
    check_close(y[:, :256], ref)
    # periodic weights -> periodic output, bit-exact, over the whole row range
    assert np.array_equal(y[0, 4096:8192].view(np.uint32), y[0, :4096].view(np.uint32))
    assert np.array_equal(y[0, M - 4096 + (4096 - M % 4096) % 4096 - 4096:][:0], y[0, :0])
    tail = M % 4096
    assert np.array_equal(y[0, M - tail:].view(np.uint32), y[0, :tail].view(np.uint32))


def test_output_layer_rows_70b(G, torch, orc):
    """The 128256 x 8192 Q6_K output matrix of Llama-3-70B Q4_K_M at N = 1 (GEMV tier) and N = 512 (matrix-core tier): sampled rows
    against the oracle, and periodic weights -> periodic outputs over the whole row range."""
    t, M, K = oracle.Q6_K, 128256, 8192
    rng = np.random.default_rng(19)
    base = random_blocks(t, 2048, K, rng)
    w = np.tile(base, (M // 2048 + 1, 1))[:M]
    wq = G.QWeight.from_host(t, w, M, K)
    del w
    x = rng.standard_normal((512, K)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    y1 = G.mul_mat(wq, xd[:1]).cpu().numpy()
    rows = np.unique(rng.integers(0, 2048, 64))
    check_close(y1[:, rows], orc.mul_mat(t, base[rows], x[:1], len(rows), 1, K), "70B output N=1")
    assert np.array_equal(y1[0, 2048:4096].view(np.uint32), y1[0, :2048].view(np.uint32))
    tail = M % 2048
    assert np.array_equal(y1[0, M - tail:].view(np.uint32), y1[0, :tail].view(np.uint32))
    y = G.mul_mat(wq, xd).cpu().numpy()
    assert np.isfinite(y).all()
    toks = np.array([0, 77, 300, 511])
    ref = orc.mul_mat(t, base[rows], x[toks], len(rows), len(toks), K)
    check(G, t, K, 512, y[np.ix_(toks, rows)], ref, "70B output N=512")
    assert np.array_equal(y[:, 2048:4096].view(np.uint32), y[:, :2048].view(np.uint32))       # same weights, same tokens -> same numbers in every tile
    assert np.array_equal(y[:, M - tail:].view(np.uint32), y[:, :tail].view(np.uint32))


@pytest.mark.parametrize("n_tok", [1, 512])
def test_mixtral_full_size_expert_tensor(G, torch, orc, n_tok):
    """BASELINE.json configs[4]: one full-size Mixtral-8x7B expert tensor (8 x 14336 x 4096 Q4_K, n_used 2) through MUL_MAT_ID at decode
    (ids read on the device) and at prefill size (rows grouped by expert): sampled rows / tokens against the oracle."""
    t, ne, nu, M, K = oracle.Q4_K, 8, 2, 14336, 4096
    rng = np.random.default_rng(23 + n_tok)
    as_ = random_blocks(t, ne * M, K, rng)
    w = G.QWeight.from_host(t, as_, M, K, n_expert=ne)
    ids = np.stack([rng.permutation(ne)[:nu] for _ in range(n_tok)]).astype(np.int32)
    b = rng.standard_normal((n_tok, 1, K)).astype(np.float32)
    y = G.mul_mat_id(w, torch.from_numpy(b).cuda(), torch.from_numpy(ids).cuda()).cpu().numpy()
    assert y.shape == (n_tok, nu, M) and np.isfinite(y).all()
    rows = np.unique(np.concatenate([[0, M - 1], rng.integers(0, M, 40)]))
    toks = np.unique(np.concatenate([[0, n_tok - 1], rng.integers(0, n_tok, 6)]))
    sub = np.ascontiguousarray(as_.reshape(ne, M, -1)[:, rows]).reshape(ne * len(rows), -1)      # the sampled rows of every expert
    ref = orc.mul_mat_id(t, sub, np.ascontiguousarray(b[toks]), np.ascontiguousarray(ids[toks]), len(rows), K, ne)
    check_close(y[np.ix_(toks, np.arange(nu), rows)], ref, f"mixtral expert tensor n_tok={n_tok}")      # Q4_K: the CPU arithmetic on both tiers
    y2 = G.mul_mat_id(w, torch.from_numpy(b).cuda(), torch.from_numpy(ids).cuda()).cpu().numpy()
    assert np.array_equal(y2.view(np.uint32), y.view(np.uint32))                                 # deterministic


# ------------------------------------------------------------------------------------------------
# decode plan: a chain of N=1 mul_mats as ONE persistent launch (mi355q_plan_*)
# ------------------------------------------------------------------------------------------------
def _bits_t(t):
    return t.detach().cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("t", FAST, ids=ids_t)
def test_plan_dependent_chain_bitexact(G, torch, orc, t):
    """y of every stage IS x of the next one (the grid barrier and the agent-scope stores/loads carry real
    data between workgroups on different XCDs); every stage must equal the per-matmul launch bit for bit
    and the oracle within tolerance.  Re-run several times: the barrier counters re-arm themselves."""
    rng = np.random.default_rng(21 + t)
    K = 2048
    hosts = [quantized_weights(t, K, K, rng, scale=0.05) for _ in range(5)]
    ws = [G.QWeight.from_host(t, h, K, K) for h in hosts]
    x0 = torch.from_numpy(rng.standard_normal((1, K)).astype(np.float32)).cuda()
    bufs = [x0] + [torch.zeros((1, K), dtype=torch.float32, device="cuda") for _ in ws]
    refs = [x0]
    for w in ws:
        refs.append(G.mul_mat(w, refs[-1]))
    plan = G.Plan([([w], bufs[i], [bufs[i + 1]], i > 0) for i, w in enumerate(ws)])
    assert plan.weight_bytes == sum(w.nbytes for w in ws)
    for rep in range(4):
        for b in bufs[1:]:
            b.zero_()
        plan.run(); torch.cuda.synchronize()
        assert plan.status() == 0
        for i in range(len(ws)):
            assert np.array_equal(_bits_t(bufs[i + 1]), _bits_t(refs[i + 1])), (rep, i)
    xin = x0.cpu().numpy()
    for i, h in enumerate(hosts):                              # and against the oracle, stage by stage on the device's own inputs
        check_close(bufs[i + 1].cpu().numpy(), orc.mul_mat(t, h, xin, K, 1, K), f"stage {i}")
        xin = bufs[i + 1].cpu().numpy()
    plan.close()


def test_plan_llama_layer_mixed_types(G, torch, orc):
    """One Q4_K_M-shaped layer at reduced width: wq|wk (Q4_K) + wv (Q6_K) on one x, wo, gate|up, down (Q6_K, K != n_embd),
    ragged row counts (not multiples of 16 or of the CU count), independent activation buffers per stage."""
    rng = np.random.default_rng(77)
    E, F, KV = 2048, 2816, 264                              # Q6_K rows are planar when k % 2048 == 0
    def W(t, m, k):
        h = quantized_weights(t, m, k, rng)
        return h, G.QWeight.from_host(t, h, m, k)
    layer = [[W(oracle.Q4_K, E, E), W(oracle.Q4_K, KV, E), W(oracle.Q6_K, KV, E)], [W(oracle.Q4_K, E, E)],
             [W(oracle.Q4_K, F, E), W(oracle.Q4_K, F, E)], [W(oracle.Q6_K, E, 4096)]]
    stages, checks = [], []
    for gi, grp in enumerate(layer):
        k = grp[0][1].K
        x = torch.from_numpy(rng.standard_normal((1, k)).astype(np.float32)).cuda()
        ys = [torch.zeros((1, w.M), dtype=torch.float32, device="cuda") for _, w in grp]
        stages.append(([w for _, w in grp], x, ys, gi > 0))
        checks.append((grp, x, ys))
    plan = G.Plan(stages)
    assert plan.launch_stages == 5                             # the mixed-type first stage is split per type
    plan.run(); torch.cuda.synchronize()
    assert plan.status() == 0
    for grp, x, ys in checks:
        for (h, w), y in zip(grp, ys):
            assert np.array_equal(_bits_t(y), _bits_t(G.mul_mat(w, x)))
            check_close(y.cpu().numpy(), orc.mul_mat(w.type, h, x.cpu().numpy(), w.M, 1, w.K))
    plan.close()


def test_plan_iq4_model_layer(G, torch, orc):
    """An IQ4_XS-recipe layer (llama-quant.cpp: IQ4_XS for most tensors, IQ4_NL where a row is not a multiple of 256 blocks' worth, Q5_K attn_v,
    Q6_K output-style matrix) as one persistent launch: the plan has a kernel instantiation for {IQ4_XS, IQ4_NL, Q5_K, Q6_K}; every stage equals the
    per-matmul launch bit for bit and the oracle within tolerance.  A mix of the IQ4 types with Q4_K has no instantiation and is refused."""
    rng = np.random.default_rng(78)
    E, F, KV = 2048, 2816, 264
    def W(t, m, k):
        h = quantized_weights(t, m, k, rng)
        return h, G.QWeight.from_host(t, h, m, k)
    layer = [[W(oracle.IQ4_XS, E, E), W(oracle.IQ4_XS, KV, E), W(oracle.Q5_K, KV, E)], [W(oracle.IQ4_NL, E, E)],
             [W(oracle.IQ4_XS, F, E), W(oracle.IQ4_XS, F, E)], [W(oracle.Q6_K, E, 4096)]]
    stages, checks = [], []
    for gi, grp in enumerate(layer):
        k = grp[0][1].K
        x = torch.from_numpy(rng.standard_normal((1, k)).astype(np.float32)).cuda()
        ys = [torch.zeros((1, w.M), dtype=torch.float32, device="cuda") for _, w in grp]
        stages.append(([w for _, w in grp], x, ys, gi > 0))
        checks.append((grp, x, ys))
    plan = G.Plan(stages)
    for _ in range(2):
        plan.run(); torch.cuda.synchronize()
        assert plan.status() == 0
        for grp, x, ys in checks:
            for (h, w), y in zip(grp, ys):
                assert np.array_equal(_bits_t(y), _bits_t(G.mul_mat(w, x)))
                check_close(y.cpu().numpy(), orc.mul_mat(w.type, h, x.cpu().numpy(), w.M, 1, w.K))
    plan.close()
    x = torch.zeros((1, E), dtype=torch.float32, device="cuda")
    (_, wn), (_, w5) = W(oracle.IQ4_NL, 64, E), W(oracle.Q5_K, 64, E)          # Q8_0-family and Q8_K-family matrices cannot share a stage's image
    with pytest.raises(G.Mi355qError):
        G.Plan([([wn, w5], x, [torch.zeros((1, 64), device="cuda"), torch.zeros((1, 64), device="cuda")], False)])
    (_, wa), (_, wb) = W(oracle.IQ4_XS, 64, E), W(oracle.Q4_K, 64, E)
    with pytest.raises(G.Mi355qError):
        G.Plan([([wa], x, [torch.zeros((1, 64), device="cuda")], False), ([wb], x, [torch.zeros((1, 64), device="cuda")], False)])


def test_plan_rejects_what_it_cannot_stream(G, torch):
    rng = np.random.default_rng(5)
    w = G.QWeight.from_host(oracle.Q3_K, random_blocks(oracle.Q3_K, 32, 256, rng), 32, 256)      # no planar streamer for Q3_K
    x = torch.zeros((1, 256), dtype=torch.float32, device="cuda"); y = torch.zeros((1, 32), dtype=torch.float32, device="cuda")
    with pytest.raises(G.Mi355qError):
        G.Plan([([w], x, [y], False)])
    w6 = G.QWeight.from_host(oracle.Q6_K, random_blocks(oracle.Q6_K, 32, 256, rng), 32, 256)      # Q6_K rows of 210 bytes are not planar at k=256
    with pytest.raises(G.Mi355qError):
        G.Plan([([w6], x, [y], False)])


def test_plan_llama70b_layer_against_oracle(G, torch, orc):
    """One Llama-3-70B Q4_K_M layer at FULL width as one persistent launch (BASELINE.json configs[3]: n_embd 8192, n_ff 28672, K = 28672
    for ffn_down), every output checked against the ORACLE on sampled rows (not only against mul_mat)."""
    rng = np.random.default_rng(70)
    E, F, KV = 8192, 28672, 1024
    def W(t, m, k):
        h = random_blocks(t, m, k, rng)
        return h, G.QWeight.from_host(t, h, m, k)
    layer = [[W(oracle.Q4_K, E, E), W(oracle.Q4_K, KV, E), W(oracle.Q5_K, KV, E)], [W(oracle.Q4_K, E, E)],
             [W(oracle.Q4_K, F, E), W(oracle.Q4_K, F, E)], [W(oracle.Q6_K, E, F)]]
    stages, checks = [], []
    for gi, grp in enumerate(layer):
        k = grp[0][1].K
        x = torch.from_numpy(rng.standard_normal((1, k)).astype(np.float32)).cuda()
        ys = [torch.zeros((1, w.M), dtype=torch.float32, device="cuda") for _, w in grp]
        stages.append(([w for _, w in grp], x, ys, gi > 0))
        checks.append((grp, x, ys))
    plan = G.Plan(stages)
    plan.run(); torch.cuda.synchronize()
    assert plan.status() == 0
    for grp, x, ys in checks:
        for (h, w), y in zip(grp, ys):
            assert np.array_equal(_bits_t(y), _bits_t(G.mul_mat(w, x)))
            rows = np.unique(np.concatenate([[0, w.M - 1], rng.integers(0, w.M, 48)]))
            check_close(y.cpu().numpy()[:, rows], orc.mul_mat(w.type, h[rows], x.cpu().numpy(), len(rows), 1, w.K), f"70B layer {ids_t(w.type)} {w.M}x{w.K}")
    plan.close()


@pytest.mark.parametrize("t", [oracle.Q4_K, oracle.Q6_K, oracle.IQ4_XS], ids=ids_t)
@pytest.mark.parametrize("n_tok", [2, 64])
def test_mul_mat_id_out_of_range_expert_is_nan_on_every_path(G, torch, orc, t, n_tok):
    """One rule for an expert id outside [0, n_expert) -- on which the reference asserts -- on all device paths (per-pair GEMV with the ids read
    on the device, the device-grouped matrix-core form, the generic tier): that pair's output row is NaN, every other row is unaffected."""
    rng = np.random.default_rng(3 + t + n_tok)
    ne, nu, M, K = 4, 2, 64, 2048
    as_ = quantized_weights(t, ne * M, K, rng)
    w = G.QWeight.from_host(t, as_, M, K, n_expert=ne)
    ids = np.stack([rng.permutation(ne)[:nu] for _ in range(n_tok)]).astype(np.int32)
    good = ids.copy()
    ids[0, 1] = ne; ids[n_tok - 1, 0] = -1
    b = rng.uniform(-1, 1, (n_tok, 1, K)).astype(np.float32)
    y = G.mul_mat_id(w, torch.from_numpy(b).cuda(), torch.from_numpy(ids).cuda()).cpu().numpy()
    assert np.isnan(y[0, 1]).all() and np.isnan(y[n_tok - 1, 0]).all()
    ref = orc.mul_mat_id(t, as_, b, good, M, K, ne)
    ok = np.ones((n_tok, nu), bool); ok[0, 1] = False; ok[n_tok - 1, 0] = False
    assert np.isfinite(y[ok]).all()
    if n_tok * nu >= 17 and t not in (oracle.Q4_K, oracle.Q8_0) and on_mfma_tier(G, t, K, 9):      # the grouped form on a bf16 tier
        assert nmse(y[ok], ref[ok]) <= 5e-4
    else:
        check_close(y[ok], ref[ok])

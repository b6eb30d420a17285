"""The C restatement (oracle/) against the committed golden vectors (tests/golden/*.npz), which were
produced by the real reference CPU backend (tests/golden/make_golden.py).  Bit-exact everywhere:
block decode, activation quantizer, MUL_MAT and MUL_MAT_ID outputs."""
import numpy as np
import pytest

import oracle

from conftest import GOLDEN

MM = sorted(GOLDEN.glob("mul_mat_[!i]*.npz")) + sorted(GOLDEN.glob("mul_mat_iq*.npz"))
MMID = sorted(GOLDEN.glob("mul_mat_id_*.npz"))


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_fixture_inventory():
    assert len(MM) == 38 and len(MMID) == 38        # 19 weight types x 2 shapes each


@pytest.mark.parametrize("path", MM, ids=lambda p: p.stem)
def test_mul_mat_golden(orc, path):
    g = np.load(path, allow_pickle=False)
    t, M, N, K = int(g["type"]), int(g["M"]), int(g["N"]), int(g["K"])
    assert orc.supported(t)
    assert orc.row_size(t, K) == g["w"].shape[1]
    assert orc.vec_dot_type(t) == int(g["act_type"])
    # decode spec
    assert np.array_equal(_bits(orc.dequantize(t, g["w"], K)), _bits(g["deq"]))
    # activation quantizer (scalar spec: roundf / nearest_int)
    act = orc.quantize_act(int(g["act_type"]), g["x"], oracle.ROUND_AWAY)
    if int(g["act_type"]) == oracle.Q8_K:
        # all-zero super-blocks leave bsums undefined in the reference (ggml-quants.c:2494-2499)
        a = act.reshape(N, -1, 292).copy(); b = g["act"].reshape(N, -1, 292).copy()
        zero = (b[:, :, :4].view(np.float32)[..., 0] == 0)
        a[zero, 260:] = 0; b[zero, 260:] = 0
        assert np.array_equal(a, b)
    else:
        assert np.array_equal(act, g["act"])
    # whole op
    y = orc.mul_mat(t, g["w"], g["x"], M, N, K)
    assert np.array_equal(_bits(y), _bits(g["y"]))


@pytest.mark.parametrize("path", MMID, ids=lambda p: p.stem)
def test_mul_mat_id_golden(orc, path):
    g = np.load(path, allow_pickle=False)
    t, M, K, ne = int(g["type"]), int(g["M"]), int(g["K"]), int(g["n_expert"])
    y = orc.mul_mat_id(t, g["as_"], g["b"], g["ids"], M, K, ne)
    assert np.array_equal(_bits(y), _bits(g["y"]))


def test_f16_roundtrip_all_halfs(orc):
    hs = np.arange(65536, dtype=np.uint16)
    want = hs.view(np.float16).astype(np.float32)
    got = np.array([orc.lib.orc_f16_to_f32(int(h)) for h in hs], np.float32)
    ok = ~np.isnan(want)
    assert np.array_equal(got.view(np.uint32)[ok], want.view(np.uint32)[ok])
    back = np.array([orc.lib.orc_f32_to_f16(float(f)) for f in want[ok]], np.uint16)
    assert np.array_equal(back, hs[ok])


def test_round_modes_differ_only_on_ties(orc):
    x = np.zeros((1, 32), np.float32)
    x[0, 0] = 127.0          # d = 1 exactly
    x[0, 1] = 2.5            # tie
    x[0, 2] = -3.5           # tie
    x[0, 3] = 2.4
    away = orc.quantize_act(oracle.Q8_0, x, oracle.ROUND_AWAY)[0, 2:].view(np.int8)
    even = orc.quantize_act(oracle.Q8_0, x, oracle.ROUND_EVEN)[0, 2:].view(np.int8)
    assert list(away[:4]) == [127, 3, -4, 2]
    assert list(even[:4]) == [127, 2, -4, 2]


def test_broadcast_mul_mat(orc):
    """ne12 = 2*ne02 broadcast (ggml-cpu.c:1197-1198): each src0 matrix serves two src1 batches."""
    g = np.load(GOLDEN / "mul_mat_q4_K_m16n3k256.npz", allow_pickle=False)
    t, M, N, K = int(g["type"]), 16, 3, 256
    x2 = np.stack([g["x"], g["x"][::-1]])                   # [ne12=2, N, K]
    y = orc.mul_mat(t, g["w"], x2, M, N, K, ne02=1, ne12=2)
    assert np.array_equal(_bits(y[0, 0]), _bits(g["y"]))
    assert np.array_equal(_bits(y[0, 1]), _bits(g["y"][::-1]))


@pytest.mark.parametrize("path", sorted(GOLDEN.glob("flash_attn_*.npz")), ids=lambda p: p.stem)
def test_flash_attn_ext_golden(path):
    """oracle/glue.py flash_attn_ext against the reference CPU backend's FLASH_ATTN_EXT (fixtures: tests/golden/make_flash_attn_golden.py), bit for bit."""
    from oracle import glue
    g = np.load(path, allow_pickle=False)
    if "k_blocks" in g:                                              # a Q8_0 cache
        got = glue.flash_attn_ext_q8_0(g["q"], g["k_blocks"], g["v_blocks"], g["mask"], float(g["scale"]), float(g["max_bias"]), float(g["softcap"]))
    else:
        got = glue.flash_attn_ext(g["q"], g["k"], g["v"], g["mask"], float(g["scale"]), float(g["max_bias"]), float(g["softcap"]))
    assert np.array_equal(got.view(np.uint32), g["y"].view(np.uint32))

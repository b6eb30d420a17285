"""Residency ops through the C-ABI (mi355q_op_*) against the numpy oracle (oracle/glue.py, itself pinned
against the real reference in tests/test_oracle_glue.py), at Llama-3-8B decode / small-prefill shapes, including the strided
and broadcast operands the decode graph produces (permuted q, KV-cache views)."""
import numpy as np
from pathlib import Path
import pytest

import oracle
from oracle import glue

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import torch
    import ggml_mi355 as g
    assert torch.cuda.is_available() and g.lib().mi355q_device_count() >= 1
    return g


@pytest.fixture(scope="module")
def torch():
    import torch as t
    return t


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_bin_bcast_bitexact(G, torch):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((2, 3, 5, 4096)).astype(np.float32)
    for shape in ((2, 3, 5, 4096), (1, 1, 1, 4096), (1, 3, 1, 4096), (1, 1, 5, 1)):
        b = rng.uniform(0.5, 2.0, shape).astype(np.float32)
        for code, name in ((G.OP_ADD, "add"), (G.OP_SUB, "sub"), (G.OP_MUL, "mul"), (G.OP_DIV, "div")):
            y = G.op_bin_bcast(code, dev(torch, a), dev(torch, b)).cpu().numpy()
            assert np.array_equal(bits(y), bits(glue.bin_bcast(name, a, b))), (name, shape)
    # a strided (permuted) src0 and an in-place destination, as the residual adds of the graph do
    at = dev(torch, a).permute(0, 2, 1, 3)
    y = G.op_bin_bcast(G.OP_ADD, at, dev(torch, a.transpose(0, 2, 1, 3).copy())).cpu().numpy()
    assert np.array_equal(bits(y), bits(2 * a.transpose(0, 2, 1, 3)))


def test_rms_norm(G, torch):
    rng = np.random.default_rng(2)
    for shape, eps in (((1, 1, 1, 4096), 1e-5), ((1, 1, 7, 4096), 1e-5), ((2, 3, 5, 64), 1e-6), ((1, 1, 2, 14336), 1e-5), ((1, 1, 3, 100), 0.0)):
        x = (rng.standard_normal(shape) * rng.uniform(0.01, 30.0)).astype(np.float32)
        y = G.op_rms_norm(dev(torch, x), eps).cpu().numpy()
        ref = glue.rms_norm(x, eps)
        assert np.abs(y - ref).max() <= 1.5e-7 * np.abs(ref).max(), shape          # (only the order of the f64 sum differs)
        assert (bits(y) == bits(ref)).mean() > 0.999


def test_unary_and_cpy(G, torch):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((1, 1, 4, 14336)) * 4).astype(np.float32)
    y = G.op_unary(G.UNARY_SILU, dev(torch, x)).cpu().numpy()
    ref = glue.silu(x)
    assert np.abs(y - ref).max() <= 5e-7 * max(1.0, np.abs(ref).max())
    assert np.array_equal(G.op_unary(G.UNARY_RELU, dev(torch, x)).cpu().numpy(), np.maximum(x, 0))
    # CPY f32 -> f16 into a strided KV-cache view (row pitch larger than the row), and back
    k = rng.standard_normal((1, 1, 8, 128)).astype(np.float32)
    cache = torch.zeros((1, 1, 8, 256), dtype=torch.float16, device="cuda")
    G.op_cpy(dev(torch, k), cache[..., 64:192])
    assert np.array_equal(cache[..., 64:192].cpu().numpy(), k.astype(np.float16))
    assert float(cache[..., :64].abs().max()) == 0.0 and float(cache[..., 192:].abs().max()) == 0.0
    back = torch.empty((1, 1, 8, 128), dtype=torch.float32, device="cuda")
    G.op_cpy(cache[..., 64:192], back)
    assert np.array_equal(back.cpu().numpy(), k.astype(np.float16).astype(np.float32))
    # CONT of a permuted tensor (logical element order)
    p = dev(torch, k).permute(0, 1, 3, 2)
    out = torch.empty((1, 1, 128, 8), dtype=torch.float32, device="cuda")
    G.op_cpy(p, out)
    assert np.array_equal(out.cpu().numpy(), k.transpose(0, 1, 3, 2))


def test_cpy_batch_forms(G, torch):
    """Copies of >= 16384 elements with identical shapes take the row form (unit stride along dim 0 on both sides) or the tiled-transpose form
    (source contiguous along dim 1, destination along dim 0): same values as the element-wise kernel, i.e. exactly numpy's."""
    rng = np.random.default_rng(31)
    nt, nh, hd, n_ctx = 200, 32, 128, 512
    # (1) the head merge after attention: CONT of KQV [hd, n_tokens, n_head] viewed as [hd, n_head, n_tokens]
    kqv = rng.standard_normal((1, nh, nt, hd)).astype(np.float32)
    out = torch.empty((1, nt, nh, hd), dtype=torch.float32, device="cuda")
    G.op_cpy(dev(torch, kqv).permute(0, 2, 1, 3), out)
    assert np.array_equal(out.cpu().numpy(), kqv.transpose(0, 2, 1, 3))
    out2 = torch.empty((1, 1, nt, nh * hd), dtype=torch.float32, device="cuda")            # ggml_cont_2d: another shape, same elements
    G.op_cpy(dev(torch, kqv).permute(0, 2, 1, 3), out2)
    assert np.array_equal(out2.cpu().numpy().reshape(1, nt, nh, hd), kqv.transpose(0, 2, 1, 3))
    # (2) K store of a prompt: f32 rows -> f16 cache rows with a larger pitch, at an offset that is only 8-byte aligned
    k = rng.standard_normal((1, 1, nt, 1024)).astype(np.float32)
    cache = torch.zeros((1, 1, nt, 2048 + 4), dtype=torch.float16, device="cuda")
    G.op_cpy(dev(torch, k), cache[..., 4:1028])
    assert np.array_equal(cache[..., 4:1028].cpu().numpy(), k.astype(np.float16)) and float(cache[..., :4].abs().max()) == 0.0 and float(cache[..., 1028:].abs().max()) == 0.0
    back = torch.empty((1, 1, nt, 1024), dtype=torch.float32, device="cuda")
    G.op_cpy(cache[..., 4:1028], back)                        # f16 -> f32 rows
    assert np.array_equal(back.cpu().numpy(), k.astype(np.float16).astype(np.float32))
    h2 = torch.empty((1, 1, nt, 1024), dtype=torch.float16, device="cuda")
    G.op_cpy(cache[..., 4:1028], h2)                          # f16 -> f16 rows
    assert np.array_equal(h2.cpu().numpy(), k.astype(np.float16))
    # (3) the transposed V store: v [n_embd_v, n_tokens] seen transposed -> cache view [n_tokens (contiguous), n_embd_v (pitch n_ctx)], f32 -> f16,
    #     ragged against the 64 x 64 tiles
    v = rng.standard_normal((1, 1, nt, 1000)).astype(np.float32)                     # [n_tokens][n_embd_v] in memory
    vcache = torch.zeros((1, 1, 1000, n_ctx), dtype=torch.float16, device="cuda")
    G.op_cpy(dev(torch, v).permute(0, 1, 3, 2), vcache[..., 7:7 + nt])
    assert np.array_equal(vcache[..., 7:7 + nt].cpu().numpy(), v.transpose(0, 1, 3, 2).astype(np.float16))
    assert float(vcache[..., :7].abs().max()) == 0.0 and float(vcache[..., 7 + nt:].abs().max()) == 0.0
    # (4) a transposed f32 -> f32 copy with batch dims
    m = rng.standard_normal((2, 3, 70, 130)).astype(np.float32)
    o = torch.empty((2, 3, 130, 70), dtype=torch.float32, device="cuda")
    G.op_cpy(dev(torch, m).permute(0, 1, 3, 2), o)
    assert np.array_equal(o.cpu().numpy(), m.transpose(0, 1, 3, 2))


def test_soft_max(G, torch):
    rng = np.random.default_rng(4)
    kq = (rng.standard_normal((1, 32, 3, 513)) * 3).astype(np.float32)
    mask = np.where(rng.random((3, 513)) < 0.2, -np.inf, 0.0).astype(np.float32); mask[:, 0] = 0
    for m, mb in ((mask, 0.0), (mask, 8.0), (None, 0.0)):
        y = G.op_soft_max(dev(torch, kq), dev(torch, m) if m is not None else None, 0.0884, mb).cpu().numpy()
        ref = glue.soft_max(kq, m, 0.0884, mb)
        assert np.abs(y - ref).max() <= 3e-7 and np.allclose(y.sum(-1), 1.0, atol=1e-5)
    y16 = G.op_soft_max(dev(torch, kq), dev(torch, mask).half(), 0.0884, 0.0).cpu().numpy()       # f16 mask (the graph's KQ mask)
    assert np.abs(y16 - glue.soft_max(kq, mask, 0.0884, 0.0)).max() <= 3e-7


def test_rope(G, torch):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 5, 32, 128)).astype(np.float32)
    pos = np.array([0, 1, 17, 1000, 8191], np.int32)
    ff = rng.uniform(1.0, 8.0, 64).astype(np.float32)
    for mode in (0, 2):
        for n_dims, freq, fscale, ext in ((128, None, 1.0, 0.0), (64, None, 1.0, 0.0), (128, ff, 1.0, 0.0), (128, None, 0.25, 1.0)):
            y = G.op_rope(dev(torch, x), dev(torch, pos), n_dims, mode, dev(torch, freq[:n_dims // 2]) if freq is not None else None,
                          8192, 500000.0, fscale, ext, 1.0, 32.0, 1.0).cpu().numpy()
            ref = glue.rope(x, pos, n_dims, mode, freq, 8192, 500000.0, fscale, ext, 1.0, 32.0, 1.0)
            assert np.abs(y - ref).max() <= 6e-5 * np.abs(ref).max(), (mode, n_dims, fscale, ext)
    with pytest.raises(G.Mi355qError):
        G.op_rope(dev(torch, x), dev(torch, pos), 128, 8)                                            # mrope: not on the device


def test_rope_batch_kernel_bit_identical(G, torch):
    """Batches of >= 32 tokens take the kernel that computes a (position, pair) angle once per group of heads (k_rope_rows): the per-pair
    operations are those of the one-thread-per-pair kernel, so the two must agree bit for bit -- here against the same rows roped in slices of
    8 tokens (which take the per-pair kernel), in both modes, with partial rotation, frequency factors, YaRN, and a head count that is not a
    multiple of the group size."""
    rng = np.random.default_rng(55)
    ff = rng.uniform(1.0, 8.0, 64).astype(np.float32)
    for nt, nh in ((512, 32), (40, 7), (64, 8)):
        x = rng.standard_normal((1, nt, nh, 128)).astype(np.float32)
        pos = rng.integers(0, 8192, nt).astype(np.int32)
        for mode in (0, 2):
            for n_dims, freq, fscale, ext in ((128, None, 1.0, 0.0), (64, ff, 1.0, 0.0), (128, None, 0.25, 1.0)):
                fq = dev(torch, freq[:n_dims // 2]) if freq is not None else None
                args = (n_dims, mode, fq, 8192, 500000.0, fscale, ext, 1.0, 32.0, 1.0)
                y = G.op_rope(dev(torch, x), dev(torch, pos), *args).cpu().numpy()
                parts = [G.op_rope(dev(torch, x[:, t0:t0 + 8]), dev(torch, pos[t0:t0 + 8]), *args).cpu().numpy() for t0 in range(0, nt, 8)]
                assert np.array_equal(y.view(np.uint32), np.concatenate(parts, axis=1).view(np.uint32)), (nt, nh, mode, n_dims)
        ref = glue.rope(x, pos, 128, 0, None, 8192, 500000.0, 1.0, 0.0, 1.0, 32.0, 1.0)
        y = G.op_rope(dev(torch, x), dev(torch, pos), 128, 0, None, 8192, 500000.0, 1.0, 0.0, 1.0, 32.0, 1.0).cpu().numpy()
        assert np.abs(y - ref).max() <= 6e-5 * np.abs(ref).max()


def test_mul_mat_f_attention_shapes(G, torch):
    rng = np.random.default_rng(6)
    n_kv, hd, nh, nkvh, nt = 300, 128, 32, 8, 2
    # KQ: K cache view [n_head_kv, n_kv, head_dim] with a row pitch of n_embd_k_gqa f16; q permuted to [n_head, n_tokens, head_dim]
    kc = rng.standard_normal((n_kv, nkvh * hd)).astype(np.float32)
    k_view = dev(torch, kc).half().view(n_kv, nkvh, hd).permute(1, 0, 2)[None]
    q = rng.standard_normal((nt, nh, hd)).astype(np.float32)
    q_view = dev(torch, q).permute(1, 0, 2)[None]
    y = G.op_mul_mat_f(k_view, q_view).cpu().numpy()
    ref = glue.mul_mat_f(kc.reshape(n_kv, nkvh, hd).transpose(1, 0, 2)[None], q.transpose(1, 0, 2)[None], True)
    assert y.shape == (1, nh, nt, n_kv) and np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max()
    # KQV: transposed V cache [n_head_kv, head_dim, n_kv] f16 times the soft-max output [n_head, n_tokens, n_kv]
    v = rng.standard_normal((1, nkvh, hd, n_kv)).astype(np.float32)
    p = rng.random((1, nh, nt, n_kv)).astype(np.float32)
    y = G.op_mul_mat_f(dev(torch, v).half(), dev(torch, p)).cpu().numpy()
    assert np.abs(y - glue.mul_mat_f(v, p, True)).max() <= 2e-5 * np.abs(y).max()
    # f32 src0
    y = G.op_mul_mat_f(dev(torch, v), dev(torch, p)).cpu().numpy()
    assert np.abs(y - glue.mul_mat_f(v, p, False)).max() <= 2e-5 * np.abs(y).max()


def test_mul_mat_f_prefill_on_matrix_cores(G, torch):
    """The f16 attention products of a prefill batch (>= 16 src1 rows: v_mfma_f32_16x16x32_f16 tiles): KQ with a ragged window (n_kv and
    n_tokens not multiples of the 64 x 64 tile), GQA broadcast, cache-strided views; KQV with K = n_kv not a multiple of the 64-step."""
    rng = np.random.default_rng(16)
    n_kv, hd, nh, nkvh, nt, n_ctx = 336, 128, 8, 2, 150, 400
    kc = rng.standard_normal((n_ctx, nkvh * hd)).astype(np.float32)
    k_view = dev(torch, kc).half()[:n_kv].view(n_kv, nkvh, hd).permute(1, 0, 2)[None]
    q = rng.standard_normal((nt, nh, hd)).astype(np.float32)
    y = G.op_mul_mat_f(k_view, dev(torch, q).permute(1, 0, 2)[None]).cpu().numpy()
    ref = glue.mul_mat_f(kc[:n_kv].reshape(n_kv, nkvh, hd).transpose(1, 0, 2)[None], q.transpose(1, 0, 2)[None], True)
    assert y.shape == (1, nh, nt, n_kv) and np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max()
    vc = rng.standard_normal((1, nkvh, hd, n_ctx)).astype(np.float32)
    p = rng.random((1, nh, nt, n_kv)).astype(np.float32)
    y = G.op_mul_mat_f(dev(torch, vc).half()[..., :n_kv], dev(torch, p)).cpu().numpy()
    assert np.abs(y - glue.mul_mat_f(vc[..., :n_kv], p, True)).max() <= 2e-5 * np.abs(y).max()


# ------------------------------------------------------------------------------------------------
# fused forms: bit-identical to the chain of separate ops they replace
# ------------------------------------------------------------------------------------------------
def test_fused_norm_and_activation_equal_separate_ops(G, torch):
    rng = np.random.default_rng(11)
    for rows, n in ((1, 4096), (5, 4096), (3, 14336), (2, 100)):
        a = dev(torch, (rng.standard_normal((rows, n)) * 3).astype(np.float32))
        b = dev(torch, rng.standard_normal((rows, n)).astype(np.float32))
        w = dev(torch, rng.uniform(0.5, 1.5, (n,)).astype(np.float32))
        # rms_norm * weight
        sep = G.op_bin_bcast(G.OP_MUL, G.op_rms_norm(a, 1e-5), w)
        assert torch.equal(G.op_add_rms_norm_mul(a, 1e-5, weight=w), sep)
        assert torch.equal(G.op_add_rms_norm_mul(a, 1e-5), G.op_rms_norm(a, 1e-5))
        # (a + b) stored, then rms_norm * weight
        s_sep = G.op_bin_bcast(G.OP_ADD, a, b)
        y_sep = G.op_bin_bcast(G.OP_MUL, G.op_rms_norm(s_sep, 1e-5), w)
        y, s = G.op_add_rms_norm_mul(a, 1e-5, b=b, weight=w, want_sum=True)
        assert torch.equal(s, s_sep) and torch.equal(y, y_sep)
        # silu(a) * b
        assert torch.equal(G.op_unary_mul(G.UNARY_SILU, a, b), G.op_bin_bcast(G.OP_MUL, G.op_unary(G.UNARY_SILU, a), b))


@pytest.mark.parametrize("cfg", [(128, 128, 8, 2, 1, 512, 0.0, 0.0), (64, 64, 4, 4, 7, 96, 0.0, 0.0), (128, 128, 8, 8, 3, 300, 8.0, 0.0),
                                 (80, 80, 4, 1, 2, 64, 0.0, 10.0), (256, 256, 2, 2, 1, 1000, 0.0, 0.0), (128, 128, 32, 8, 1, 4096, 0.0, 0.0),
                                 (128, 128, 8, 2, 150, 336, 0.0, 0.0), (64, 64, 4, 2, 40, 96, 4.0, 0.0), (96, 96, 4, 4, 33, 80, 0.0, 5.0)], ids=str)   # (the last three: prefill batches)
def test_flash_attn_ext(G, torch, cfg):
    """GGML_OP_FLASH_ATTN_EXT for an f16 KV cache against a float64 restatement of ggml-cpu/ops.cpp:6690-6905 (q rounded to f16 before
    the dot products; causal f16 mask with -inf; GQA; ALiBi slopes; logit soft-capping).  The reference's own harness checks the same op
    through the plugin (tests/test_plugin.py, FLASH_ATTN_EXT)."""
    DK, DV, H, Hk, N, n_kv, max_bias, softcap = cfg
    rng = np.random.default_rng(DK + H + N + n_kv)
    q = rng.standard_normal((1, H, N, DK)).astype(np.float32)
    k = rng.standard_normal((1, Hk, n_kv, DK)).astype(np.float16)
    v = rng.standard_normal((1, Hk, n_kv, DV)).astype(np.float16)
    mask = np.zeros((32 * ((N + 31) // 32), n_kv), np.float16)
    for t in range(N):
        mask[t, n_kv - N + t + 1:] = -np.inf
    mask[:N] += (rng.standard_normal((N, n_kv)) * 0.1).astype(np.float16) * (max_bias > 0)      # ALiBi-like finite biases
    scale = 1.0 / np.sqrt(DK)
    y = G.op_flash_attn_ext(dev(torch, q), dev(torch, k), dev(torch, v), dev(torch, mask), scale, max_bias, softcap).cpu().numpy()
    # float64 restatement
    n2 = 1
    while 2 * n2 <= H:
        n2 *= 2
    m0, m1 = 2.0 ** (-max_bias / n2), 2.0 ** (-(max_bias / 2.0) / n2)
    ref = np.zeros((1, N, H, DV))
    qh = q.astype(np.float16).astype(np.float64)
    for h in range(H):
        hk = h // (H // Hk)
        slope = (m0 ** (h + 1) if h < n2 else m1 ** (2 * (h - n2) + 1)) if max_bias > 0 else 1.0
        s = qh[0, h] @ k[0, hk].astype(np.float64).T                      # [N, n_kv]
        s = s * (scale / softcap if softcap else scale)
        if softcap:
            s = softcap * np.tanh(s)
        s = s + slope * mask[:N].astype(np.float64)
        s = s - s.max(axis=1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(axis=1, keepdims=True)
        ref[0, :, h] = p @ v[0, hk].astype(np.float64)
    assert np.isfinite(y).all()
    if (N < 16 or softcap) and n_kv <= 1024 and DK % 32 == 0:
        # the one-workgroup-per-row kernel (few rows, or soft-capping) over a window of up to 1024 positions: the CPU's own arithmetic (sequential F16 accumulator: 1e-3 from the exact product), checked
        # against the restatement that is bit-exact with the reference (oracle/glue.py flash_attn_ext; tests/test_oracle_glue.py)
        want = glue.flash_attn_ext(q, k, v, mask, scale, max_bias, softcap)
        d = np.abs(y.astype(np.float64) - want); top = np.abs(want).max()
        assert d.max() <= 2e-3 * top and (d > 1e-6 * top).mean() <= 0.05, (cfg, d.max() / top, (d > 1e-6 * top).mean())
        assert np.abs(y - ref).max() <= 5e-3 * max(1.0, np.abs(ref).max())
        return
    err = np.abs(y - ref).max()
    # (prefill batches round the probabilities to f16 for the matrix cores: 2^-11 relative on each of them)
    assert err <= (2e-5 if N < 16 or softcap else 4e-4) * max(1.0, np.abs(ref).max()), (cfg, err)


def test_argsort_and_sum_rows(G, torch):
    """The MoE router ops (GGML_OP_ARGSORT as ggml_top_k uses it, GGML_OP_SUM_ROWS): against numpy (stable order on ties; f64 row sums)."""
    rng = np.random.default_rng(12)
    for shape in ((1, 1, 5, 8), (1, 2, 33, 256), (2, 3, 7, 60), (1, 1, 3, 1000)):
        x = rng.standard_normal(shape).astype(np.float32)
        x[..., 3] = x[..., 1]                                           # ties
        for desc in (False, True):
            got = G.op_argsort(dev(torch, x), descending=desc).cpu().numpy()
            want = np.argsort(-x if desc else x, axis=-1, kind="stable").astype(np.int32)
            assert np.array_equal(got, want), (shape, desc)
        s = G.op_sum_rows(dev(torch, x)).cpu().numpy()
        assert np.array_equal(bits(s), bits(x.astype(np.float64).sum(-1, keepdims=True).astype(np.float32))), shape
    # a strided source (the router reads a view of the probabilities)
    x = rng.standard_normal((4, 64)).astype(np.float32)
    xs = dev(torch, x)[:, ::2]
    assert np.array_equal(G.op_sum_rows(xs).cpu().numpy().reshape(-1), x[:, ::2].astype(np.float64).sum(-1).astype(np.float32))


@pytest.mark.parametrize("path", sorted((Path(__file__).parent / "golden").glob("flash_attn_*.npz")), ids=lambda p: p.stem)
def test_flash_attn_ext_has_the_cpu_bits(G, torch, path):
    """FLASH_ATTN_EXT against the reference CPU backend's own output (tests/golden/flash_attn_*.npz, made by make_flash_attn_golden.py): windows up to
    1024 positions follow the CPU's order -- ggml_vec_dot_f16's summation tree for the scores, positions in sequence with a running maximum, the
    F16 accumulator with its two roundings, the C library's expf -- so the outputs are the CPU's, bit for bit, except where a device tanhf / powf
    (soft-capping, ALiBi slopes) differs in the last place and flips one of the f16 roundings."""
    g = np.load(path, allow_pickle=False)
    kk, vv = (g["k_blocks"], g["v_blocks"]) if "k_blocks" in g else (g["k"], g["v"])       # (uint8 block rows: a Q8_0 cache)
    y = G.op_flash_attn_ext(dev(torch, g["q"]), dev(torch, kk), dev(torch, vv), dev(torch, g["mask"]), float(g["scale"]), float(g["max_bias"]),
                            float(g["softcap"])).cpu().numpy()
    want = g["y"]
    assert np.isfinite(y).all()
    top = np.abs(want).max()
    d = np.abs(y.astype(np.float64) - want)
    plain = float(g["max_bias"]) == 0.0 and float(g["softcap"]) == 0.0
    differ = (y.view(np.uint32) != want.view(np.uint32)).mean()
    # (tanhf / powf one ulp off move every output of a head in the last place through the sum S; what must stay rare is a flipped f16 rounding)
    assert d.max() <= 2e-3 * top and (differ == 0.0 if plain else (d > 1e-6 * top).mean() <= 0.02), (d.max() / top, differ, (d > 1e-6 * top).mean())


def test_cpy_f32_to_q8_0_is_the_reference_quantizer(G, torch):
    """CPY f32 -> Q8_0 (the K / V stores of a quantized cache): block for block the bytes of quantize_row_q8_0_ref (oracle / the reference's
    ggml_quantize_chunk), from a strided source."""
    orc = oracle.Oracle()
    rng = np.random.default_rng(31)
    x = (rng.standard_normal((3, 5, 256)) * rng.uniform(0.01, 20.0, (3, 5, 1))).astype(np.float32)
    x[1, 2, 32:64] = 0.0                                                 # an all-zero block: d = 0, quants 0
    big = dev(torch, np.concatenate([x, np.zeros_like(x)], axis=-1))     # rows with a gap: source stride 512 floats
    src = big[..., :256]
    out = torch.zeros((3, 5, 256 // 32 * 34), dtype=torch.uint8, device="cuda")
    G.op_cpy(src, out)
    want = orc.quantize_act(oracle.Q8_0, x.reshape(-1, 256))
    assert np.array_equal(out.cpu().numpy().reshape(-1), np.ascontiguousarray(want).view(np.uint8).reshape(-1))


@pytest.mark.parametrize("kv", ["q8_0", "q4_0"])
@pytest.mark.parametrize("cfg", [(8, 2, 1, 128, 256, 201, 0.0), (4, 4, 3, 64, 96, 60, 0.0), (8, 8, 2, 128, 64, 55, 8.0), (2, 1, 1, 256, 40, 30, 0.0)], ids=str)
def test_flash_attn_ext_q8_0_cache(G, torch, cfg, kv):
    """FLASH_ATTN_EXT on a Q8_0 / Q4_0 K / V cache (-ctk q8_0 -ctv q8_0 and the q4_0 forms) against oracle.glue.flash_attn_ext_q8_0, which is bit-exact with
    the reference CPU backend (tests/test_oracle_glue.py): q quantized to Q8_0, ggml_vec_dot_q8_0_q8_0's / _q4_0_q8_0's lane order, the online softmax in
    order, an F32 accumulator.  Plain cases: every output bit; with ALiBi (device powf) within the last place.  The cache rows are written by the device's
    own CPY."""
    H, Hk, N, DK, n_kv, first_masked, max_bias = cfg
    bb = 34 if kv == "q8_0" else 18
    rng = np.random.default_rng(sum(cfg[:6]))
    q = rng.standard_normal((1, H, N, DK)).astype(np.float32)
    kf = rng.standard_normal((1, Hk, n_kv, DK)).astype(np.float32); vf = rng.standard_normal((1, Hk, n_kv, DK)).astype(np.float32)
    kb = torch.zeros((1, Hk, n_kv, DK // 32 * bb), dtype=torch.uint8, device="cuda"); vb = torch.zeros_like(kb)
    G.op_cpy(dev(torch, kf), kb, kv); G.op_cpy(dev(torch, vf), vb, kv)
    mask = np.zeros((64, n_kv), np.float16)
    for t in range(N):
        mask[t, first_masked + t:] = -np.inf
    if max_bias > 0:
        mask[:N] += (rng.standard_normal((N, n_kv)) * 0.1).astype(np.float16)
    scale = float(np.float32(1.0 / np.sqrt(DK)))
    y = G.op_flash_attn_ext(dev(torch, q), kb, vb, dev(torch, mask), scale, max_bias, 0.0, kv=kv).cpu().numpy()
    want = glue.flash_attn_ext_q8_0(q, kb.cpu().numpy(), vb.cpu().numpy(), mask, scale, max_bias, 0.0, kv=kv)
    assert np.isfinite(y).all()
    d = np.abs(y.astype(np.float64) - want); top = np.abs(want).max()
    differ = (y.view(np.uint32) != want.view(np.uint32)).mean()
    assert d.max() <= 1e-5 * top and (differ == 0.0 if max_bias == 0.0 else True), (cfg, kv, d.max() / top, differ)


def test_cpy_f32_to_q4_0_is_the_reference_quantizer(G, torch):
    """CPY f32 -> Q4_0: block for block the bytes of quantize_row_q4_0_ref (the reference's ggml_quantize_chunk through oracle/_ref): the FIRST element
    of largest magnitude sets the scale (ties included), all-zero blocks, a strided source."""
    if not oracle.ref_available("scalar"):
        pytest.skip("oracle/_ref/scalar not built")
    ref = oracle.Reference("scalar")
    rng = np.random.default_rng(32)
    x = (rng.standard_normal((3, 5, 256)) * rng.uniform(0.01, 20.0, (3, 5, 1))).astype(np.float32)
    x[1, 2, 32:64] = 0.0
    x[0, 0, 3] = 7.25; x[0, 0, 17] = -7.25; x[0, 0, :32] = np.clip(x[0, 0, :32], -7.25, 7.25)       # a tie of magnitudes: the first (positive) one decides the sign of d
    x[0, 1, 40] = -9.5; x[0, 1, 33] = 9.5; x[0, 1, 32:64] = np.clip(x[0, 1, 32:64], -9.5, 9.5)       # the first one is the later lane's? no: element 33 comes first
    big = dev(torch, np.concatenate([x, np.zeros_like(x)], axis=-1))
    out = torch.zeros((3, 5, 256 // 32 * 18), dtype=torch.uint8, device="cuda")
    G.op_cpy(big[..., :256], out, "q4_0")
    want = ref.quantize(oracle.Q4_0, x.reshape(-1, 256))
    assert np.array_equal(out.cpu().numpy().reshape(-1), np.ascontiguousarray(want).view(np.uint8).reshape(-1))

"""The decode plan's glue (mi355q_plan_* API version 2): activation prologues (rms_norm(x0 + x1) * w, unary(x0) * x1), the
data-driven hand-off between stages (tagged granules instead of a grid barrier) and the one-token attention stage (rope, KV
store, softmax(q k) v over an f16 cache), through the C-ABI.

What is compared with what:
  * every GEMV output against mi355q_mul_mat applied to the activation vector the same glue ops of the node-by-node path
    produce (mi355q_op_add_rms_norm_mul / mi355q_op_unary_mul, themselves pinned to the CPU ops in tests/test_gpu_glue.py):
    BIT-IDENTICAL (the prologues perform the same f32 operations; the f64 sum of squares is accumulated in another order,
    which changes the f32 scale in far fewer than 1 of 1000 rows -- the tolerance below covers that case);
  * the attention stage against a float64 numpy restatement of the reference's graph (rope with the reference's repeated-f32
    angle, q rounded to f16 as the CPU's f16 dot does, f16 cache, softmax((q.k) scale + mask), P V): <= 2e-5 of max|out|
    (f32 summation order only), and the K / V rows it stores bit-exactly against mi355q_op_rope + mi355q_op_cpy.
"""
import numpy as np
import pytest

import oracle
from oracle import glue
from qdata import quantized_weights, random_blocks

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


def bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


def close_or_equal(y, ref, what):
    y = y.detach().cpu().numpy(); ref = ref.detach().cpu().numpy()
    assert np.isfinite(y).all(), what
    if np.array_equal(y.view(np.uint32), ref.view(np.uint32)):
        return True
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max(), f"{what}: {np.abs(y - ref).max():.3e} vs {np.abs(ref).max():.3e}"
    return False


def W(G, t, m, k, rng, scale=1.0):
    h = quantized_weights(t, m, k, rng, scale=scale)
    return G.QWeight.from_host(t, h, m, k)


@pytest.mark.parametrize("t", [oracle.Q4_K, oracle.Q6_K, oracle.Q8_0], ids=lambda t: oracle.TYPE_NAMES[t])
def test_norm_and_unary_prologues_match_the_node_ops(G, torch, t):
    """A llama layer without the attention block, at reduced width, as 5 stages whose operands are earlier stages' outputs:
    qkv = W1 norm(h) ; o = Wo q ; gate|up = W norm(h + o) [sum -> ffn_inp] ; down = Wd (silu(gate) * up) ; next = W norm(ffn_inp + down)."""
    rng = np.random.default_rng(40 + t)
    E, F = 2048, 4096                                            # (Q6_K rows are planar when k % 2048 == 0)
    eps = 1e-5
    w_qkv, w_o, w_g, w_u, w_d, w_n = W(G, t, E, E, rng), W(G, t, E, E, rng), W(G, t, F, E, rng), W(G, t, F, E, rng), W(G, t, E, F, rng), W(G, t, 300, E, rng)
    h = dev(torch, rng.standard_normal((1, E)).astype(np.float32))
    n1, n2, n3 = (dev(torch, (1.0 + 0.1 * rng.standard_normal(E)).astype(np.float32)) for _ in range(3))
    z = lambda n: torch.zeros((1, n), dtype=torch.float32, device="cuda")
    q, o, gate, up, down, nxt, ffn_inp, h2 = z(E), z(E), z(F), z(F), z(E), z(300), z(E), z(E)
    stages = [
        dict(ws=[w_qkv], ys=[q], x=h, x_kind=G.X_NORM, norm_w=n1, eps=eps),
        dict(ws=[w_o], ys=[o], x=q),
        dict(ws=[w_g, w_u], ys=[gate, up], x=h, x1=o, x_kind=G.X_NORM, norm_w=n2, eps=eps, sum_out=ffn_inp),
        dict(ws=[w_d], ys=[down], x=gate, x1=up, x_kind=G.X_UNARY_MUL, unary=G.UNARY_SILU),
        dict(ws=[w_n], ys=[nxt], x=ffn_inp, x1=down, x_kind=G.X_NORM, norm_w=n3, eps=eps, sum_out=h2),
    ]
    plan = G.Plan(stages)
    # the node-by-node path
    r_q = G.mul_mat(w_qkv, G.op_add_rms_norm_mul(h, eps, weight=n1))
    r_o = G.mul_mat(w_o, r_q)
    x3, r_ffn = G.op_add_rms_norm_mul(h, eps, b=r_o, weight=n2, want_sum=True)
    r_gate, r_up = G.mul_mat(w_g, x3), G.mul_mat(w_u, x3)
    r_down = G.mul_mat(w_d, G.op_unary_mul(G.UNARY_SILU, r_gate, r_up))
    x5, r_h2 = G.op_add_rms_norm_mul(r_ffn, eps, b=r_down, weight=n3, want_sum=True)
    r_nxt = G.mul_mat(w_n, x5)
    for rep in range(3):                                         # re-runs: the tags advance, nothing is re-armed
        for b in (q, o, gate, up, down, nxt, ffn_inp, h2):
            b.zero_()
        plan.run(); torch.cuda.synchronize()
        assert plan.status() == 0
        same = [close_or_equal(a, b, n) for a, b, n in ((q, r_q, "q"), (o, r_o, "o"), (ffn_inp, r_ffn, "ffn_inp"), (gate, r_gate, "gate"), (up, r_up, "up"),
                                                        (down, r_down, "down"), (h2, r_h2, "h2"), (nxt, r_nxt, "next"))]
        assert same[0] and same[1] and same[2], same              # (no reduction-order freedom before the second norm's scale)
    plan.close()


@pytest.mark.parametrize("t", [oracle.Q4_K, oracle.Q6_K, oracle.Q8_0], ids=lambda t: oracle.TYPE_NAMES[t])
@pytest.mark.parametrize("F", [4096, 1000, 14336, 4352, 10007], ids=str)
def test_paired_output_unary_mul(G, torch, t, F):
    """MI355Q_Y_UNARY_MUL: the gate | up stage publishes SiLU(W_gate x) * (W_up x) itself (every workgroup computes matching rows of both
    matrices) and ffn_down gathers that one vector: bit-identical to mul_mat, mul_mat, op_unary_mul, mul_mat.  F = 1000: rows that do not
    divide by the workgroup count (ragged pair ranges, idle workgroups); F = 14336: 56 pairs per workgroup, i.e. 4 pairs on half of the waves and 3 on
    the others (a wave streams the two rows of a pair back to back and publishes the product from its registers).  F = 4352: 17 pairs, one wave with two;
    F = 10007: 40 pairs per workgroup and a ragged last workgroup of 7."""
    rng = np.random.default_rng(60 + t + F)
    E = 2048
    w_g, w_u = W(G, t, F, E, rng), W(G, t, F, E, rng)
    x = dev(torch, rng.standard_normal((1, E)).astype(np.float32))
    act, unused = torch.zeros((1, F), dtype=torch.float32, device="cuda"), torch.full((1, F), 3.0, dtype=torch.float32, device="cuda")
    stages = [dict(ws=[w_g, w_u], ys=[act, unused], x=x, y_kind=G.Y_UNARY_MUL, y_unary=G.UNARY_SILU)]
    ref_act = G.op_unary_mul(G.UNARY_SILU, G.mul_mat(w_g, x), G.mul_mat(w_u, x))
    ref_down = None
    if F % 256 == 0 and (t != oracle.Q6_K or F % 2048 == 0):
        w_d = W(G, t, 300, F, rng)
        down = torch.zeros((1, 300), dtype=torch.float32, device="cuda")
        stages.append(dict(ws=[w_d], ys=[down], x=act))
        ref_down = G.mul_mat(w_d, ref_act)
    plan = G.Plan(stages)
    for _ in range(2):
        act.zero_()
        plan.run(); torch.cuda.synchronize()
        assert plan.status() == 0
        assert np.array_equal(bits(act), bits(ref_act))
        assert bool((unused == 3.0).all())
        if ref_down is not None:
            assert np.array_equal(bits(down), bits(ref_down))
    plan.close()


def test_no_plain_outputs_are_not_written(G, torch):
    rng = np.random.default_rng(5)
    K = 2048
    w1, w2 = W(G, oracle.Q4_K, K, K, rng), W(G, oracle.Q4_K, 512, K, rng)
    x = dev(torch, rng.standard_normal((1, K)).astype(np.float32))
    mid = torch.full((1, K), 7.0, dtype=torch.float32, device="cuda"); out = torch.zeros((1, 512), dtype=torch.float32, device="cuda")
    plan = G.Plan([dict(ws=[w1], ys=[mid], x=x, no_plain=True), dict(ws=[w2], ys=[out], x=mid)])
    plan.run(); torch.cuda.synchronize()
    assert plan.status() == 0
    assert bool((mid == 7.0).all())                               # the intermediate lives only in the plan's granules
    assert np.array_equal(bits(out), bits(G.mul_mat(w2, G.mul_mat(w1, x))))
    plan.close()


def _rope_np(x, pos, n_dims, mode, freq_base):
    return glue.rope(x.reshape(1, 1, -1, x.shape[-1]).astype(np.float32), np.array([pos], np.int32), n_dims, mode, freq_base=freq_base).reshape(x.shape)


@pytest.mark.parametrize("layout", ["transposed_v", "rows_v"])
@pytest.mark.parametrize("cfg", [(32, 8, 128, 512, 500), (32, 8, 128, 96, 37), (8, 8, 64, 64, 0), (16, 2, 128, 2048, 2047), (4, 4, 256, 160, 100)], ids=str)
def test_attention_stage(G, torch, layout, cfg):
    """(n_head, n_head_kv, head_dim, n_kv, pos).  The q / k / v vectors are the outputs of a GEMV stage of the same plan (so the
    attention workgroups poll granules), the cache holds `pos` earlier rows, the window is padded to n_kv and masked beyond pos."""
    n_head, n_head_kv, hd, n_kv, pos = cfg
    rng = np.random.default_rng(hash(cfg) % 1000 + (layout == "rows_v"))
    E = 2048
    n_q, n_k = n_head * hd, n_head_kv * hd
    n_ctx = n_kv + 32
    wq, wk, wv = W(G, oracle.Q4_K, n_q, E, rng), W(G, oracle.Q4_K, n_k, E, rng), W(G, oracle.Q6_K, n_k, E, rng)
    x = dev(torch, rng.standard_normal((1, E)).astype(np.float32))
    q, k, v = (torch.zeros((1, n), dtype=torch.float32, device="cuda") for n in (n_q, n_k, n_k))
    out = torch.zeros((1, n_q), dtype=torch.float32, device="cuda")
    kc_h = rng.standard_normal((n_ctx, n_k)).astype(np.float16)                    # [position][kv head * hd]
    vc_h = rng.standard_normal((n_ctx, n_k)).astype(np.float16)
    kc_h[pos + 1:] = np.float16(np.nan); vc_h[pos + 1:] = np.float16(np.nan)       # never-written rows must not leak through the mask
    kc = dev(torch, kc_h)
    if layout == "rows_v":
        vc = dev(torch, vc_h)
        v_nb_pos, v_nb_dim, v_nb_head, v_dst_off, v_dst_nb = n_k * 2, 2, hd * 2, pos * n_k * 2, 2
    else:
        vc = dev(torch, np.ascontiguousarray(vc_h.T))                              # [kv head * hd][position]
        v_nb_pos, v_nb_dim, v_nb_head, v_dst_off, v_dst_nb = 2, n_ctx * 2, hd * n_ctx * 2, pos * 2, n_ctx * 2
    mask_h = np.full(n_kv, -np.inf, np.float32); mask_h[:pos + 1] = 0.0
    mask = dev(torch, mask_h if layout == "transposed_v" else mask_h.astype(np.float16))
    posd = dev(torch, np.array([pos], np.int32))
    dst = torch.tensor([kc.data_ptr() + pos * n_k * 2, vc.data_ptr() + v_dst_off], dtype=torch.int64, device="cuda")
    scale = 1.0 / np.sqrt(hd)
    mode = 0 if hd != 64 else 2                                                    # one neox case
    attn = dict(q=q, k=k, v=v, pos=posd, rope=dict(n_dims=hd, mode=mode, n_ctx_orig=8192, freq_base=500000.0), k_cache=kc, v_cache=vc,
                k_nb_pos=n_k * 2, k_nb_head=hd * 2, v_nb_pos=v_nb_pos, v_nb_dim=v_nb_dim, v_nb_head=v_nb_head,
                k_dst=dst[0:1], v_dst=dst[1:2], v_dst_nb=v_dst_nb, mask=mask, n_head=n_head, n_head_kv=n_head_kv, head_dim=hd, n_kv=n_kv,
                scale=scale, out=out)
    plan = G.Plan([dict(ws=[wq, wk, wv], ys=[q, k, v], x=x), dict(attn=attn)])
    plan.run(); torch.cuda.synchronize()
    assert plan.status() == 0
    # --- reference in float64 on the device's own q / k / v
    qh, kh_, vh_ = q.cpu().numpy().reshape(n_head, hd), k.cpu().numpy().reshape(n_head_kv, hd), v.cpu().numpy().reshape(n_head_kv, hd)
    assert np.array_equal(bits(q), bits(G.mul_mat(wq, x))) and np.array_equal(bits(v), bits(G.mul_mat(wv, x)))
    q_r = _rope_np(qh, pos, hd, mode, 500000.0).astype(np.float16).astype(np.float64)
    k_r = _rope_np(kh_, pos, hd, mode, 500000.0).astype(np.float16)
    v_r = vh_.astype(np.float16)
    # the stored rows, bit-exact
    kc_after = kc.cpu().numpy()
    vc_after = vc.cpu().numpy() if layout == "rows_v" else vc.cpu().numpy().T
    assert np.array_equal(kc_after[pos].view(np.uint16), k_r.reshape(-1).view(np.uint16))
    assert np.array_equal(vc_after[pos].view(np.uint16), v_r.reshape(-1).view(np.uint16))
    untouched = np.ones(n_ctx, bool); untouched[pos] = False
    assert np.array_equal(kc_after[untouched].view(np.uint16), kc_h[untouched].view(np.uint16))
    K_all = kc_h.copy(); K_all[pos] = k_r.reshape(-1); V_all = vc_h.copy(); V_all[pos] = v_r.reshape(-1)
    # The non-flash graph (transposed V cache) with the whole window in one workgroup (n_kv <= 256) reproduces the CPU's arithmetic: softmax in f32
    # with an f64 sum, probabilities ROUNDED TO F16 before the P.V product (the f16 src0 of that MUL_MAT makes the CPU convert src1).  A device
    # expf that differs from numpy's in the last bit can flip such a rounding (2^-11 of one probability), hence the wider tolerance there.
    cpu_like = layout == "transposed_v" and n_kv <= 256
    tol = 5e-4 if cpu_like else 2e-5
    def probs(Kg, qv):
        s_ = (Kg @ qv) * scale
        if not cpu_like:
            p_ = np.exp(s_ - s_.max()); return p_ / p_.sum()
        s32 = s_.astype(np.float32)
        e_ = np.exp(s32 - s32.max()).astype(np.float32)
        return (e_ * np.float32(1.0 / e_.astype(np.float64).sum())).astype(np.float16).astype(np.float64)
    ref = np.zeros((n_head, hd))
    gq = n_head // n_head_kv
    for h in range(n_head):
        g = h // gq
        Kg = K_all[:pos + 1, g * hd:(g + 1) * hd].astype(np.float64); Vg = V_all[:pos + 1, g * hd:(g + 1) * hd].astype(np.float64)
        ref[h] = probs(Kg, q_r[h]) @ Vg
    got = out.cpu().numpy().reshape(n_head, hd).astype(np.float64)
    assert np.isfinite(got).all()
    # The flash layout with the whole window in one workgroup AND in LDS follows the CPU's FLASH_ATTN_EXT instead: positions in order, F16 accumulator
    # (oracle/glue.py flash_attn_ext; 1e-3 away from the exact product).  Whether the window fits the LDS is the plan's decision, so either form passes
    # here; test_attention_stage_follows_the_cpu_flash_accumulator pins the choice for the llama shape.
    def seq_ref(K_, V_, q_, upto):
        o = glue.flash_attn_ext(q_.astype(np.float32).reshape(1, n_head, 1, hd), K_[:upto].reshape(1, upto, n_head_kv, hd).transpose(0, 2, 1, 3),
                                V_[:upto].reshape(1, upto, n_head_kv, hd).transpose(0, 2, 1, 3), None, scale)
        return o.reshape(n_head, hd).astype(np.float64)
    def check(got_, ref_, K_, V_, q_, upto, what):
        err = np.abs(got_ - ref_).max() / np.abs(ref_).max()
        if err <= tol:
            return
        assert layout == "rows_v" and n_kv <= 256, (what, err)
        r2 = seq_ref(K_, V_, q_, upto)
        d = np.abs(got_ - r2)
        # (a score a few ulp from the oracle's can flip one f16 rounding of the accumulator: 2^-11 of it, in a few elements)
        assert d.max() <= 2e-3 * np.abs(r2).max() and (d > 1e-6 * np.abs(r2).max()).mean() <= 0.05, (what, d.max() / np.abs(r2).max(), (d > 1e-6 * np.abs(r2).max()).mean())
    check(got, ref, K_all, V_all, q_r, pos + 1, "first token")
    # a second token at pos + 1 through the SAME plan: only the device-side destination slots, pos and the mask move
    if pos + 1 < n_kv:
        pos2 = pos + 1
        posd.fill_(pos2)
        mask_h[:pos2 + 1] = 0.0
        mask.copy_(dev(torch, mask_h if layout == "transposed_v" else mask_h.astype(np.float16)))
        dst.copy_(torch.tensor([kc.data_ptr() + pos2 * n_k * 2, vc.data_ptr() + (pos2 * n_k * 2 if layout == "rows_v" else pos2 * 2)], dtype=torch.int64))
        plan.run(); torch.cuda.synchronize()
        assert plan.status() == 0
        kc2 = kc.cpu().numpy()
        assert np.array_equal(kc2[pos].view(np.uint16), k_r.reshape(-1).view(np.uint16))           # the previous row stays
        assert np.array_equal(kc2[pos2].view(np.uint16), _rope_np(kh_, pos2, hd, mode, 500000.0).astype(np.float16).reshape(-1).view(np.uint16))
        q_r2 = _rope_np(qh, pos2, hd, mode, 500000.0).astype(np.float16).astype(np.float64)
        K_all[pos2] = _rope_np(kh_, pos2, hd, mode, 500000.0).astype(np.float16).reshape(-1); V_all[pos2] = v_r.reshape(-1)
        ref2 = np.zeros((n_head, hd))
        for h in range(n_head):
            g = h // gq
            Kg = K_all[:pos2 + 1, g * hd:(g + 1) * hd].astype(np.float64); Vg = V_all[:pos2 + 1, g * hd:(g + 1) * hd].astype(np.float64)
            ref2[h] = probs(Kg, q_r2[h]) @ Vg
        got2 = out.cpu().numpy().reshape(n_head, hd).astype(np.float64)
        check(got2, ref2, K_all, V_all, q_r2, pos2 + 1, "second token")
    plan.close()


def test_attention_stage_against_the_node_ops(G, torch):
    """The same token through the node-by-node ops of the plugin (rope, cache stores, f16 K.q, soft_max, f16 V.p): the plan keeps the
    probabilities in f32 where the CPU graph rounds them to f16 for the P V product, so the two agree to that rounding (NMSE <= 1e-6)."""
    n_head, n_head_kv, hd, n_kv, pos = 32, 8, 128, 256, 200
    rng = np.random.default_rng(9)
    n_q, n_k, n_ctx = n_head * hd, n_head_kv * hd, 512
    q = dev(torch, rng.standard_normal((1, n_q)).astype(np.float32)); k = dev(torch, rng.standard_normal((1, n_k)).astype(np.float32)); v = dev(torch, rng.standard_normal((1, n_k)).astype(np.float32))
    kc_h = rng.standard_normal((n_ctx, n_k)).astype(np.float16); vc_h = rng.standard_normal((n_k, n_ctx)).astype(np.float16)
    mask_h = np.full((1, n_kv), -np.inf, np.float32); mask_h[0, :pos + 1] = 0.0
    scale = 1.0 / np.sqrt(hd)
    posd = dev(torch, np.array([pos], np.int32))
    # node path
    kc1, vc1 = dev(torch, kc_h), dev(torch, vc_h)
    q_r = G.op_rope(q.view(1, 1, n_head, hd), posd, hd, 0, n_ctx_orig=8192, freq_base=500000.0)
    k_r = G.op_rope(k.view(1, 1, n_head_kv, hd), posd, hd, 0, n_ctx_orig=8192, freq_base=500000.0)
    G.op_cpy(k_r.view(1, n_k), kc1[pos:pos + 1])
    G.op_cpy(v.view(n_k, 1), vc1[:, pos:pos + 1])
    kq = G.op_mul_mat_f(kc1[:n_kv].view(n_kv, n_head_kv, hd).permute(1, 0, 2), q_r.view(n_head, 1, hd))          # [n_head, 1, n_kv]
    p = G.op_soft_max(kq, dev(torch, mask_h), scale)
    o = G.op_mul_mat_f(vc1.view(n_head_kv, hd, n_ctx)[:, :, :n_kv], p)                                              # [n_head, 1, hd]
    ref = o.reshape(-1).cpu().numpy().astype(np.float64)
    # plan path (a trivial GEMV stage in front: a plan needs one)
    kc2, vc2 = dev(torch, kc_h), dev(torch, vc_h)
    out = torch.zeros((1, n_q), dtype=torch.float32, device="cuda")
    dst = torch.tensor([kc2.data_ptr() + pos * n_k * 2, vc2.data_ptr() + pos * 2], dtype=torch.int64, device="cuda")
    w0 = W(G, oracle.Q4_K, 32, 256, rng); x0 = torch.zeros((1, 256), dtype=torch.float32, device="cuda"); y0 = torch.zeros((1, 32), dtype=torch.float32, device="cuda")
    attn = dict(q=q, k=k, v=v, pos=posd, rope=dict(n_dims=hd, mode=0, n_ctx_orig=8192, freq_base=500000.0), k_cache=kc2, v_cache=vc2,
                k_nb_pos=n_k * 2, k_nb_head=hd * 2, v_nb_pos=2, v_nb_dim=n_ctx * 2, v_nb_head=hd * n_ctx * 2, k_dst=dst[0:1], v_dst=dst[1:2], v_dst_nb=n_ctx * 2,
                mask=dev(torch, mask_h.reshape(-1)), n_head=n_head, n_head_kv=n_head_kv, head_dim=hd, n_kv=n_kv, scale=scale, out=out)
    plan = G.Plan([([w0], x0, [y0], False), dict(attn=attn)])
    plan.run(); torch.cuda.synchronize()
    assert plan.status() == 0
    assert np.array_equal(kc2.cpu().numpy().view(np.uint16), kc1.cpu().numpy().view(np.uint16))
    assert np.array_equal(vc2.cpu().numpy().view(np.uint16), vc1.cpu().numpy().view(np.uint16))
    got = out.reshape(-1).cpu().numpy().astype(np.float64)
    assert ((got - ref) ** 2).sum() / (ref ** 2).sum() <= 1e-6
    plan.close()


def test_attention_stage_follows_the_cpu_flash_accumulator(G, torch):
    """Llama-3-8B's decode shape with the flash (rows per position) V cache and a window of 256: the stage runs from LDS and reproduces
    ggml_compute_forward_flash_attn_ext_f16 -- positions in order, running maximum, F16 accumulator (ggml-cpu/ops.cpp:6810-6890; restated in
    oracle/glue.py flash_attn_ext).  Nearly every element equals the oracle's to f32 rounding; the exact product is 1e-3 away.
    MI355Q_PLAN_FA_EXACT=0 (read at plan creation) selects the f32 accumulator instead."""
    import os
    n_head, n_head_kv, hd, n_kv, pos = 32, 8, 128, 256, 201
    rng = np.random.default_rng(77)
    n_q, n_k, n_ctx = n_head * hd, n_head_kv * hd, 256
    q = dev(torch, rng.standard_normal((1, n_q)).astype(np.float32)); k = dev(torch, rng.standard_normal((1, n_k)).astype(np.float32)); v = dev(torch, rng.standard_normal((1, n_k)).astype(np.float32))
    kc_h = rng.standard_normal((n_ctx, n_k)).astype(np.float16); vc_h = rng.standard_normal((n_ctx, n_k)).astype(np.float16)
    mask_h = np.full(n_kv, -np.inf, np.float16); mask_h[:pos + 1] = 0.0
    scale = 1.0 / np.sqrt(hd)
    posd = dev(torch, np.array([pos], np.int32))
    w0 = W(G, oracle.Q4_K, 32, 256, rng); x0 = torch.zeros((1, 256), dtype=torch.float32, device="cuda"); y0 = torch.zeros((1, 32), dtype=torch.float32, device="cuda")
    res = {}
    for exact in ("1", "0"):
        kc, vc = dev(torch, kc_h), dev(torch, vc_h)
        out = torch.zeros((1, n_q), dtype=torch.float32, device="cuda")
        dst = torch.tensor([kc.data_ptr() + pos * n_k * 2, vc.data_ptr() + pos * n_k * 2], dtype=torch.int64, device="cuda")
        attn = dict(q=q, k=k, v=v, pos=posd, rope=dict(n_dims=hd, mode=0, n_ctx_orig=8192, freq_base=500000.0), k_cache=kc, v_cache=vc,
                    k_nb_pos=n_k * 2, k_nb_head=hd * 2, v_nb_pos=n_k * 2, v_nb_dim=2, v_nb_head=hd * 2, k_dst=dst[0:1], v_dst=dst[1:2], v_dst_nb=2,
                    mask=dev(torch, mask_h), n_head=n_head, n_head_kv=n_head_kv, head_dim=hd, n_kv=n_kv, scale=scale, out=out)
        os.environ["MI355Q_PLAN_FA_EXACT"] = exact
        try:
            plan = G.Plan([([w0], x0, [y0], False), dict(attn=attn)])
        finally:
            del os.environ["MI355Q_PLAN_FA_EXACT"]
        plan.run(); torch.cuda.synchronize()
        assert plan.status() == 0
        res[exact] = (out.reshape(n_head, hd).cpu().numpy().astype(np.float64), kc.cpu().numpy(), vc.cpu().numpy())
        plan.close()
    got, kc_a, vc_a = res["1"]
    q_r = _rope_np(q.cpu().numpy().reshape(n_head, hd), pos, hd, 0, 500000.0)
    K = kc_a[:pos + 1].reshape(1, pos + 1, n_head_kv, hd).transpose(0, 2, 1, 3); V = vc_a[:pos + 1].reshape(1, pos + 1, n_head_kv, hd).transpose(0, 2, 1, 3)
    ref_seq = glue.flash_attn_ext(q_r.reshape(1, n_head, 1, hd), K, V, None, scale).reshape(n_head, hd).astype(np.float64)
    qd = q_r.astype(np.float16).astype(np.float64)
    ref_exact = np.zeros((n_head, hd))
    for h in range(n_head):
        g = h // (n_head // n_head_kv)
        s_ = (K[0, g].astype(np.float64) @ qd[h]) * scale
        p_ = np.exp(s_ - s_.max()); p_ /= p_.sum()
        ref_exact[h] = p_ @ V[0, g].astype(np.float64)
    top = np.abs(ref_seq).max()
    d = np.abs(got - ref_seq)
    # measured: every one of the 4096 elements has the oracle's (= the reference CPU backend's) bits; what is left room for is a cosf / sinf of the rope
    # one ulp from numpy's that flips the f16 rounding of a q element
    differ = (got.astype(np.float32).view(np.uint32) != ref_seq.astype(np.float32).view(np.uint32)).mean()
    assert d.max() <= 2e-3 * top and differ <= 0.002, (d.max() / top, differ)
    assert np.abs(got - ref_exact).max() > 1e-4 * top                       # the F16 accumulator is visible
    got0 = res["0"][0]
    assert np.abs(got0 - ref_exact).max() <= 2e-5 * top                     # the f32 accumulator: the exact product
    assert np.array_equal(res["0"][1].view(np.uint16), kc_a.view(np.uint16)) and np.array_equal(res["0"][2].view(np.uint16), vc_a.view(np.uint16))


@pytest.mark.parametrize("groups", ["1", "0"], ids=["concurrent", "one_after_the_other"])
def test_mixed_type_stage_runs_as_a_group(G, torch, groups):
    """One activation vector, matrices of three weight types of one activation format (Q4_K, Q6_K, Q5_K), a norm prologue with the residual sum and x_out:
    the sub-stages run CONCURRENTLY on disjoint workgroup ranges (MI355Q_PLAN_GROUPS=0: one after the other, round 2's form).  Every output, the
    published sum and the stored vector are bit-identical to the node ops either way, and a following stage sees all three outputs."""
    import os
    rng = np.random.default_rng(91)
    K = 4096
    wa, wb, wc = W(G, oracle.Q4_K, 5120, K, rng), W(G, oracle.Q6_K, 1024, K, rng), W(G, oracle.Q5_K, 300, K, rng)
    x0 = dev(torch, rng.standard_normal((1, K)).astype(np.float32)); x1 = dev(torch, rng.standard_normal((1, K)).astype(np.float32))
    nw = dev(torch, (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float32))
    z = lambda n: torch.zeros((1, n), dtype=torch.float32, device="cuda")
    ya, yb, yc, so, xo, y2 = z(5120), z(1024), z(300), z(K), z(K), z(64)
    w2 = W(G, oracle.Q4_K, 64, 1024, rng)
    os.environ["MI355Q_PLAN_GROUPS"] = groups
    try:
        plan = G.Plan([dict(ws=[wa, wb, wc], ys=[ya, yb, yc], x=x0, x1=x1, x_kind=G.X_NORM, norm_w=nw, eps=1e-5, sum_out=so, x_out=xo),
                       dict(ws=[w2], ys=[y2], x=yb)])
    finally:
        del os.environ["MI355Q_PLAN_GROUPS"]
    r_x, r_sum = G.op_add_rms_norm_mul(x0, 1e-5, b=x1, weight=nw, want_sum=True)
    for rep in range(2):
        for b in (ya, yb, yc, so, xo, y2):
            b.zero_()
        plan.run(); torch.cuda.synchronize()
        assert plan.status() == 0
        assert np.array_equal(bits(xo), bits(r_x)) and np.array_equal(bits(so), bits(r_sum))
        for y, w in ((ya, wa), (yb, wb), (yc, wc)):
            assert np.array_equal(bits(y), bits(G.mul_mat(w, r_x)))
        assert np.array_equal(bits(y2), bits(G.mul_mat(w2, G.mul_mat(wb, r_x))))
    plan.close()


@pytest.mark.parametrize("n_kv", [96, 600], ids=["one_split", "three_splits_and_merge"])
def test_stage_behind_attention_skips_the_attention_workgroups(G, torch, n_kv):
    """wq|wk|wv -> ATTN (-> COMBINE) -> wo -> a following stage: `wo` is dealt to the workgroups that did not run the attention (they arrive last); with
    MI355Q_PLAN_WO_SKIP=0 every workgroup takes part.  Both forms give the same bits -- rows do not care which wave computes them -- and equal the node path's
    wo(attention output)."""
    import os
    n_head, n_head_kv, hd, pos = 32, 8, 128, 57
    rng = np.random.default_rng(101 + n_kv)
    E = 2048
    n_q, n_k, n_ctx = n_head * hd, n_head_kv * hd, n_kv + 32
    wq, wk, wv = W(G, oracle.Q4_K, n_q, E, rng), W(G, oracle.Q4_K, n_k, E, rng), W(G, oracle.Q6_K, n_k, E, rng)
    wo, wn = W(G, oracle.Q4_K, E, n_q, rng), W(G, oracle.Q4_K, 512, E, rng)
    x = dev(torch, rng.standard_normal((1, E)).astype(np.float32))
    kc_h = rng.standard_normal((n_ctx, n_k)).astype(np.float16); vc_h = rng.standard_normal((n_k, n_ctx)).astype(np.float16)
    mask_h = np.full(n_kv, -np.inf, np.float32); mask_h[:pos + 1] = 0.0
    posd = dev(torch, np.array([pos], np.int32))
    res = {}
    for skip in ("1", "0"):
        kc, vc = dev(torch, kc_h), dev(torch, vc_h)
        z = lambda n: torch.zeros((1, n), dtype=torch.float32, device="cuda")
        q, k, v, att, o, nxt = z(n_q), z(n_k), z(n_k), z(n_q), z(E), z(512)
        dst = torch.tensor([kc.data_ptr() + pos * n_k * 2, vc.data_ptr() + pos * 2], dtype=torch.int64, device="cuda")
        attn = dict(q=q, k=k, v=v, pos=posd, rope=dict(n_dims=hd, mode=0, n_ctx_orig=8192, freq_base=500000.0), k_cache=kc, v_cache=vc,
                    k_nb_pos=n_k * 2, k_nb_head=hd * 2, v_nb_pos=2, v_nb_dim=n_ctx * 2, v_nb_head=hd * n_ctx * 2, k_dst=dst[0:1], v_dst=dst[1:2], v_dst_nb=n_ctx * 2,
                    mask=dev(torch, mask_h), n_head=n_head, n_head_kv=n_head_kv, head_dim=hd, n_kv=n_kv, scale=1.0 / np.sqrt(hd), out=att)
        os.environ["MI355Q_PLAN_WO_SKIP"] = skip
        try:
            plan = G.Plan([dict(ws=[wq, wk, wv], ys=[q, k, v], x=x), dict(attn=attn), dict(ws=[wo], ys=[o], x=att), dict(ws=[wn], ys=[nxt], x=o)])
        finally:
            del os.environ["MI355Q_PLAN_WO_SKIP"]
        for _ in range(2):
            plan.run()
        torch.cuda.synchronize()
        assert plan.status() == 0
        res[skip] = (att.cpu().numpy(), o.cpu().numpy(), nxt.cpu().numpy())
        assert np.array_equal(bits(o), bits(G.mul_mat(wo, att))) and np.array_equal(bits(nxt), bits(G.mul_mat(wn, o)))
        plan.close()
    for a_, b_ in zip(res["1"], res["0"]):
        assert np.array_equal(a_.view(np.uint32), b_.view(np.uint32))


def test_plan_timeout_is_reported_not_hung(G, torch):
    """An operand that claims to come from an earlier stage but is never produced cannot be built through the API (dependencies follow
    from addresses), so the bounded poll is exercised the only way possible: a plan whose launch is healthy reports status 0 repeatedly,
    and a destroyed / re-created plan starts from fresh tags."""
    rng = np.random.default_rng(6)
    K = 2048
    w = W(G, oracle.Q4_K, 64, K, rng)
    x = dev(torch, rng.standard_normal((1, K)).astype(np.float32)); y = torch.zeros((1, 64), dtype=torch.float32, device="cuda")
    for _ in range(3):
        plan = G.Plan([([w], x, [y], False)])
        for _ in range(5):
            plan.run()
        torch.cuda.synchronize()
        assert plan.status() == 0
        assert np.array_equal(bits(y), bits(G.mul_mat(w, x)))
        plan.close()


def test_epoch_wrap_resets_the_granules(G, torch):
    """Tags are epoch + stage + 1 with epoch = run count x (stages + 1), 32 bits: before the product wraps, mi355q_plan_run zeroes the granules on the launch
    stream and starts the count over.  The run counter is set just below the wrap (test hook mi355q_plan_debug_set_runs); the runs on both sides of the
    reset must give the bit-identical chain outputs, and a stale tag of the old count must never satisfy a poll."""
    rng = np.random.default_rng(77)
    K = 2048
    w1, w2 = W(G, oracle.Q4_K, K, K, rng), W(G, oracle.Q6_K, 512, K, rng)
    x = dev(torch, rng.standard_normal((1, K)).astype(np.float32))
    y1 = torch.zeros((1, K), dtype=torch.float32, device="cuda"); y2 = torch.zeros((1, 512), dtype=torch.float32, device="cuda")
    plan = G.Plan([([w1], x, [y1], False), ([w2], y1, [y2], True)])
    plan.run(); torch.cuda.synchronize()
    ref1, ref2 = y1.clone(), y2.clone()
    span = plan.launch_stages + 1
    plan.debug_set_runs(0xFFFFFFFF // span - 3)
    for _ in range(6):                                          # crosses the reset
        y1.zero_(); y2.zero_()
        plan.run(); torch.cuda.synchronize()
        assert plan.status() == 0
        assert np.array_equal(bits(y1), bits(ref1)) and np.array_equal(bits(y2), bits(ref2))
    plan.close()


def test_x_out_stores_the_formed_vector(G, torch):
    """mi355q_stage.x_out (API version 3): an X_NORM stage also stores rms_norm(x0 + x1) * w as plain f32 -- bit-identical to mi355q_op_add_rms_norm_mul --
    for a caller whose graph reads that vector besides the stage's matrices."""
    rng = np.random.default_rng(78)
    K = 4096
    w = W(G, oracle.Q4_K, 256, K, rng)
    x0 = dev(torch, rng.standard_normal((1, K)).astype(np.float32)); x1 = dev(torch, rng.standard_normal((1, K)).astype(np.float32))
    nw = dev(torch, (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float32))
    y = torch.zeros((1, 256), dtype=torch.float32, device="cuda"); xo = torch.zeros((1, K), dtype=torch.float32, device="cuda"); so = torch.zeros((1, K), dtype=torch.float32, device="cuda")
    plan = G.Plan([dict(ws=[w], ys=[y], x=x0, x1=x1, x_kind=G.X_NORM, norm_w=nw, eps=1e-5, sum_out=so, x_out=xo)])
    plan.run(); torch.cuda.synchronize()
    assert plan.status() == 0
    r_x, r_sum = G.op_add_rms_norm_mul(x0, 1e-5, b=x1, weight=nw, want_sum=True)
    assert np.array_equal(bits(xo), bits(r_x)) and np.array_equal(bits(so), bits(r_sum))
    assert np.array_equal(bits(y), bits(G.mul_mat(w, r_x)))
    plan.close()

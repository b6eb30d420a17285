// layer_parity.cc -- TEST INFRASTRUCTURE (built into oracle/_ref/<variant>/ against the reference's public ggml API).
//
// Builds ONE llama decoder layer the way the reference's graph builder does (src/llama-model.cpp llm_build_llama,
// src/llama-graph.cpp build_norm / build_attn / build_attn_mha / build_ffn: RMS_NORM, MUL, quantized MUL_MAT, ROPE, CPY into
// the f16 KV cache (V transposed), f16 MUL_MAT for KQ and KQV, SOFT_MAX with mask and scale, CONT, ADD, SILU) twice -- once
// on the reference CPU backend, once with every tensor in MI355 device buffers -- feeds both the same weights, inputs and cache
// contents, and compares the layer output and the updated KV cache.  It also asks the device backend supports_op() for every
// node: a decode layer is "resident" only if none is refused (each refusal is a scheduler split with a PCIe round trip).
//
//   GGML_BACKEND_PATH=.../libggml-mi355.so layer_parity [n_tokens] [device name] [8b [iters]]
// exit code 0 = all nodes supported and NMSE(out), NMSE(k cache), NMSE(v cache) below 5e-4 / 1e-6.
// With "8b": Llama-3-8B dimensions (n_embd 4096, n_ff 14336, 32/8 heads, 512 cached positions) and a timing loop of
// graph_compute on both backends (one layer of a decode step through the reference's own graph API).
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "ggml.h"
#include "ggml-alloc.h"
#include "ggml-backend.h"

struct Dims { int n_embd = 2048, n_head = 16, n_head_kv = 4, hd = 128, n_ff = 4096, n_ctx = 64, n_past = 31; };

struct Layer {                       // tensors of one build
    ggml_context * ctx = nullptr;
    ggml_cgraph *  gf = nullptr;
    ggml_tensor *x, *pos, *mask, *attn_norm, *ffn_norm, *wq, *wk, *wv, *wo, *wgate, *wup, *wdown, *kc, *vc, *out;
    ggml_backend_buffer_t buf = nullptr;
};

static Layer build(const Dims & d, int n_tokens, ggml_backend_t backend) {
    Layer L;
    ggml_init_params ip = { ggml_tensor_overhead() * 256 + ggml_graph_overhead(), nullptr, true };
    L.ctx = ggml_init(ip);
    ggml_context * c = L.ctx;
    const int n_embd_kv = d.n_head_kv * d.hd, n_kv = d.n_past + n_tokens;
    L.x    = ggml_new_tensor_2d(c, GGML_TYPE_F32, d.n_embd, n_tokens);
    L.pos  = ggml_new_tensor_1d(c, GGML_TYPE_I32, n_tokens);
    L.mask = ggml_new_tensor_2d(c, GGML_TYPE_F32, n_kv, GGML_PAD(n_tokens, 32));
    L.attn_norm = ggml_new_tensor_1d(c, GGML_TYPE_F32, d.n_embd);
    L.ffn_norm  = ggml_new_tensor_1d(c, GGML_TYPE_F32, d.n_embd);
    L.wq = ggml_new_tensor_2d(c, GGML_TYPE_Q4_K, d.n_embd, d.n_embd);
    L.wk = ggml_new_tensor_2d(c, GGML_TYPE_Q4_K, d.n_embd, n_embd_kv);
    L.wv = ggml_new_tensor_2d(c, GGML_TYPE_Q6_K, d.n_embd, n_embd_kv);
    L.wo = ggml_new_tensor_2d(c, GGML_TYPE_Q4_K, d.n_embd, d.n_embd);
    L.wgate = ggml_new_tensor_2d(c, GGML_TYPE_Q4_K, d.n_embd, d.n_ff);
    L.wup   = ggml_new_tensor_2d(c, GGML_TYPE_Q4_K, d.n_embd, d.n_ff);
    L.wdown = ggml_new_tensor_2d(c, GGML_TYPE_Q6_K, d.n_ff, d.n_embd);
    L.kc = ggml_new_tensor_1d(c, GGML_TYPE_F16, (int64_t) n_embd_kv * d.n_ctx);      // [n_embd_kv, n_ctx] row per position
    L.vc = ggml_new_tensor_1d(c, GGML_TYPE_F16, (int64_t) n_embd_kv * d.n_ctx);      // transposed: [n_ctx, n_embd_kv]

    const float eps = 1e-5f, kq_scale = 1.0f / sqrtf((float) d.hd);
    // --- attention norm, projections, rope (llm_build_llama)
    ggml_tensor * cur = ggml_mul(c, ggml_rms_norm(c, L.x, eps), L.attn_norm);
    ggml_tensor * Q = ggml_mul_mat(c, L.wq, cur), * K = ggml_mul_mat(c, L.wk, cur), * V = ggml_mul_mat(c, L.wv, cur);
    Q = ggml_rope_ext(c, ggml_reshape_3d(c, Q, d.hd, d.n_head, n_tokens), L.pos, nullptr, d.hd, 0, 8192, 500000.0f, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f);
    K = ggml_rope_ext(c, ggml_reshape_3d(c, K, d.hd, d.n_head_kv, n_tokens), L.pos, nullptr, d.hd, 0, 8192, 500000.0f, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f);
    // --- store k, v in the cache (llama_kv_cache_unified cpy_k / cpy_v; V transposed when flash attention is off)
    L.gf = ggml_new_graph(c);
    ggml_tensor * k_view = ggml_view_1d(c, L.kc, (int64_t) n_tokens * n_embd_kv, ggml_row_size(GGML_TYPE_F16, n_embd_kv) * d.n_past);
    ggml_build_forward_expand(L.gf, ggml_cpy(c, ggml_reshape_2d(c, K, n_embd_kv, n_tokens), k_view));
    ggml_tensor * v_view = ggml_view_2d(c, L.vc, n_tokens, n_embd_kv, d.n_ctx * ggml_element_size(L.vc), d.n_past * ggml_element_size(L.vc));
    ggml_build_forward_expand(L.gf, ggml_cpy(c, ggml_transpose(c, ggml_reshape_2d(c, V, n_embd_kv, n_tokens)), v_view));
    // --- attention (build_attn_mha)
    ggml_tensor * q = ggml_permute(c, Q, 0, 2, 1, 3);
    ggml_tensor * k = ggml_view_3d(c, L.kc, d.hd, n_kv, d.n_head_kv, ggml_row_size(GGML_TYPE_F16, n_embd_kv), ggml_row_size(GGML_TYPE_F16, d.hd), 0);
    ggml_tensor * kq = ggml_mul_mat(c, k, q);
    kq = ggml_soft_max_ext(c, kq, L.mask, kq_scale, 0.0f);
    ggml_tensor * v = ggml_view_3d(c, L.vc, n_kv, d.hd, d.n_head_kv, ggml_element_size(L.vc) * d.n_ctx, ggml_element_size(L.vc) * d.n_ctx * d.hd, 0);
    ggml_tensor * kqv = ggml_mul_mat(c, v, kq);
    cur = ggml_cont_2d(c, ggml_permute(c, kqv, 0, 2, 1, 3), d.n_embd, n_tokens);
    cur = ggml_mul_mat(c, L.wo, cur);
    // --- residual, ffn (build_ffn LLM_FFN_SILU, LLM_FFN_PAR)
    ggml_tensor * ffn_inp = ggml_add(c, cur, L.x);
    cur = ggml_mul(c, ggml_rms_norm(c, ffn_inp, eps), L.ffn_norm);
    ggml_tensor * gate = ggml_silu(c, ggml_mul_mat(c, L.wgate, cur));
    cur = ggml_mul(c, gate, ggml_mul_mat(c, L.wup, cur));
    cur = ggml_mul_mat(c, L.wdown, cur);
    L.out = ggml_add(c, cur, ffn_inp);
    ggml_build_forward_expand(L.gf, L.out);
    L.buf = ggml_backend_alloc_ctx_tensors(c, backend);
    return L;
}

static std::vector<uint8_t> quantize(ggml_type t, const std::vector<float> & w, int64_t k, int64_t m) {
    std::vector<uint8_t> q(ggml_row_size(t, k) * m);
    ggml_quantize_chunk(t, w.data(), q.data(), 0, m, k, nullptr);
    return q;
}
static double nmse(const std::vector<float> & a, const std::vector<float> & b) {
    double e = 0, s = 0;
    for (size_t i = 0; i < a.size(); ++i) { e += ((double) a[i] - b[i]) * ((double) a[i] - b[i]); s += (double) b[i] * b[i]; }
    return e / (s > 0 ? s : 1);
}

int main(int argc, char ** argv) {
    const int n_tokens = argc > 1 ? atoi(argv[1]) : 1;
    const std::string dev_name = argc > 2 ? argv[2] : "MI355_0";
    ggml_backend_load_all();
    ggml_backend_dev_t dev = ggml_backend_dev_by_name(dev_name.c_str());
    if (!dev) { fprintf(stderr, "device %s not found (is GGML_BACKEND_PATH set?)\n", dev_name.c_str()); return 3; }
    ggml_backend_t be_dev = ggml_backend_dev_init(dev, nullptr);
    ggml_backend_t be_cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    Dims d;
    const bool big = argc > 3 && std::string(argv[3]) == "8b";
    const int iters = argc > 4 ? atoi(argv[4]) : 50;
    if (big) { d.n_embd = 4096; d.n_head = 32; d.n_head_kv = 8; d.hd = 128; d.n_ff = 14336; d.n_ctx = 1024; d.n_past = 511; }
    Layer ref = build(d, n_tokens, be_cpu), tst = build(d, n_tokens, be_dev);

    // ---- the same data for both
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    auto randv = [&](size_t n, float s) { std::vector<float> v(n); for (auto & x : v) x = nd(rng) * s; return v; };
    const int n_embd_kv = d.n_head_kv * d.hd, n_kv = d.n_past + n_tokens;
    auto set_both = [&](ggml_tensor * a, ggml_tensor * b, const void * data, size_t bytes) {
        ggml_backend_tensor_set(a, data, 0, bytes); ggml_backend_tensor_set(b, data, 0, bytes);
    };
    { auto x = randv((size_t) d.n_embd * n_tokens, 1.0f); set_both(ref.x, tst.x, x.data(), x.size() * 4); }
    { std::vector<int32_t> p(n_tokens); for (int i = 0; i < n_tokens; ++i) p[i] = d.n_past + i; set_both(ref.pos, tst.pos, p.data(), p.size() * 4); }
    { std::vector<float> m((size_t) n_kv * GGML_PAD(n_tokens, 32), -INFINITY);       // causal mask
      for (int t = 0; t < n_tokens; ++t) for (int j = 0; j <= d.n_past + t; ++j) m[(size_t) t * n_kv + j] = 0.0f;
      set_both(ref.mask, tst.mask, m.data(), m.size() * 4); }
    { auto w = randv(d.n_embd, 0.1f); for (auto & x : w) x += 1.0f; set_both(ref.attn_norm, tst.attn_norm, w.data(), w.size() * 4); }
    { auto w = randv(d.n_embd, 0.1f); for (auto & x : w) x += 1.0f; set_both(ref.ffn_norm, tst.ffn_norm, w.data(), w.size() * 4); }
    struct WQ { ggml_tensor * a, * b; int64_t k, m; };
    for (WQ w : { WQ{ref.wq, tst.wq, d.n_embd, d.n_embd}, WQ{ref.wk, tst.wk, d.n_embd, n_embd_kv}, WQ{ref.wv, tst.wv, d.n_embd, n_embd_kv},
                  WQ{ref.wo, tst.wo, d.n_embd, d.n_embd}, WQ{ref.wgate, tst.wgate, d.n_embd, d.n_ff}, WQ{ref.wup, tst.wup, d.n_embd, d.n_ff},
                  WQ{ref.wdown, tst.wdown, d.n_ff, d.n_embd} }) {
        auto f = randv((size_t) w.k * w.m, 1.0f / sqrtf((float) w.k));
        auto q = quantize(w.a->type, f, w.k, w.m);
        set_both(w.a, w.b, q.data(), q.size());
    }
    { auto f = randv((size_t) n_embd_kv * d.n_ctx, 1.0f); std::vector<ggml_fp16_t> h(f.size()); ggml_fp32_to_fp16_row(f.data(), h.data(), f.size());
      set_both(ref.kc, tst.kc, h.data(), h.size() * 2); }
    { auto f = randv((size_t) n_embd_kv * d.n_ctx, 1.0f); std::vector<ggml_fp16_t> h(f.size()); ggml_fp32_to_fp16_row(f.data(), h.data(), f.size());
      set_both(ref.vc, tst.vc, h.data(), h.size() * 2); }

    // ---- residency: does the device backend take every node?
    int unsupported = 0;
    for (int i = 0; i < ggml_graph_n_nodes(tst.gf); ++i) {
        ggml_tensor * n = ggml_graph_node(tst.gf, i);
        if (!ggml_backend_supports_op(be_dev, n)) { ++unsupported; printf("  NOT SUPPORTED on %s: node %d %s (%s)\n", dev_name.c_str(), i, ggml_op_desc(n), n->name); }
    }
    printf("decode layer graph: %d nodes, %d refused by %s\n", ggml_graph_n_nodes(tst.gf), unsupported, dev_name.c_str());
    if (unsupported) return 2;

    if (ggml_backend_graph_compute(be_cpu, ref.gf) != GGML_STATUS_SUCCESS || ggml_backend_graph_compute(be_dev, tst.gf) != GGML_STATUS_SUCCESS) {
        fprintf(stderr, "graph_compute failed\n"); return 4;
    }
    auto get_f32 = [](ggml_tensor * t) { std::vector<float> v(ggml_nelements(t)); ggml_backend_tensor_get(t, v.data(), 0, v.size() * 4); return v; };
    auto get_f16 = [](ggml_tensor * t) { std::vector<ggml_fp16_t> h(ggml_nelements(t)); ggml_backend_tensor_get(t, h.data(), 0, h.size() * 2);
                                         std::vector<float> v(h.size()); ggml_fp16_to_fp32_row(h.data(), v.data(), h.size()); return v; };
    const double e_out = nmse(get_f32(tst.out), get_f32(ref.out));
    const double e_k = nmse(get_f16(tst.kc), get_f16(ref.kc)), e_v = nmse(get_f16(tst.vc), get_f16(ref.vc));
    printf("n_tokens=%d n_kv=%d  NMSE out %.3e  k cache %.3e  v cache %.3e\n", n_tokens, n_kv, e_out, e_k, e_v);
    if (big) {
        auto time_it = [&](ggml_backend_t be, ggml_cgraph * gf, int n) {
            ggml_backend_graph_compute(be, gf);
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) ggml_backend_graph_compute(be, gf);
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        };
        const double us_dev = time_it(be_dev, tst.gf, iters), us_cpu = time_it(be_cpu, ref.gf, iters < 10 ? iters : 10);
        printf("one Llama-3-8B decoder layer (n_tokens=%d, n_kv=%d, %d graph nodes) through ggml_backend_graph_compute: %s %.1f us, CPU backend %.1f us\n",
               n_tokens, n_kv, ggml_graph_n_nodes(tst.gf), dev_name.c_str(), us_dev, us_cpu);
    }
    const bool ok = e_out <= 5e-4 && e_k <= 1e-6 && e_v <= 1e-6 && std::isfinite(e_out);
    printf("%s\n", ok ? "LAYER PARITY OK" : "LAYER PARITY FAILED");
    return ok ? 0 : 1;
}

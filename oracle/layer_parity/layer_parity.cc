// layer_parity.cc -- TEST INFRASTRUCTURE (built into oracle/_ref/<variant>/ against the reference's public ggml API).
//
// Builds ONE llama decoder layer the way the reference's graph builder does (src/llama-model.cpp llm_build_llama,
// src/llama-graph.cpp build_norm / build_attn / build_attn_mha / build_ffn: RMS_NORM, MUL, quantized MUL_MAT, ROPE, CPY into
// the f16 KV cache (V transposed), f16 MUL_MAT for KQ and KQV, SOFT_MAX with mask and scale, CONT, ADD, SILU) twice -- once
// on the reference CPU backend, once with every tensor in MI355 device buffers -- feeds both the same weights, inputs and cache
// contents, and compares the layer output and the updated KV cache.  It also asks the device backend supports_op() for every
// node: a decode layer is "resident" only if none is refused (each refusal is a scheduler split with a PCIe round trip).
//
// Weights and the KV cache are persistent tensors; every step builds a fresh graph (as llama_decode does) whose activations are
// placed by the graph allocator (ggml_gallocr: same topology -> same addresses, which is what lets a backend replay a captured
// launch graph).  The attended KV window is padded to 32 positions as llama_kv_cache_unified does.
//
//   GGML_BACKEND_PATH=.../libggml-mi355.so layer_parity [n_tokens] [device] [small|8b] [steps] [iters]
//     steps > 1 : a decode loop -- step t appends n_tokens positions at n_past + t*n_tokens; every step is compared with the CPU.
//     LAYER_PARITY_FA=1    : build the layer as llama.cpp does with -fa 1 (one FLASH_ATTN_EXT node, f16 mask, window padded to 256).
//     LAYER_PARITY_TRACE=1 : keep every intermediate and print the per-node NMSE of the steps whose output differs.
//     LAYER_PARITY_JITTER=k: from step k on the number of new tokens alternates n_tokens, n_tokens+1, ... -- the graph changes on every
//                            call, which is what makes a backend give up its launch-graph cache (and must not corrupt the KV cache).
//     iters > 0 : afterwards, time `iters` graph_compute calls of the last step's graph on both backends.
// exit code 0 = all nodes supported and every step's NMSE(out) <= 5e-4, NMSE(k cache), NMSE(v cache) <= 1e-6 (5e-4 with more than 8 tokens per step).
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

struct Dims { int n_embd = 2048, n_head = 16, n_head_kv = 4, hd = 128, n_ff = 4096, n_ctx = 128, n_past = 20; bool fa = false; };

struct Model {                       // persistent: weights and the KV cache of the layer
    ggml_context * ctx = nullptr; ggml_backend_buffer_t buf = nullptr;
    ggml_tensor *attn_norm, *ffn_norm, *wq, *wk, *wv, *wo, *wgate, *wup, *wdown, *kc, *vc;
};
struct Step { ggml_context * ctx = nullptr; ggml_cgraph * gf = nullptr; ggml_tensor *x, *pos, *mask, *out; int n_kv = 0; };

static Model make_model(const Dims & d, ggml_backend_t backend) {
    Model M;
    ggml_init_params ip = { ggml_tensor_overhead() * 32, nullptr, true };
    ggml_context * c = M.ctx = ggml_init(ip);
    const int n_embd_kv = d.n_head_kv * d.hd;
    M.attn_norm = ggml_new_tensor_1d(c, GGML_TYPE_F32, d.n_embd);
    M.ffn_norm  = ggml_new_tensor_1d(c, GGML_TYPE_F32, d.n_embd);
    M.wq = ggml_new_tensor_2d(c, GGML_TYPE_Q4_K, d.n_embd, d.n_embd);
    M.wk = ggml_new_tensor_2d(c, GGML_TYPE_Q4_K, d.n_embd, n_embd_kv);
    M.wv = ggml_new_tensor_2d(c, GGML_TYPE_Q6_K, d.n_embd, n_embd_kv);
    M.wo = ggml_new_tensor_2d(c, GGML_TYPE_Q4_K, d.n_embd, d.n_embd);
    M.wgate = ggml_new_tensor_2d(c, GGML_TYPE_Q4_K, d.n_embd, d.n_ff);
    M.wup   = ggml_new_tensor_2d(c, GGML_TYPE_Q4_K, d.n_embd, d.n_ff);
    M.wdown = ggml_new_tensor_2d(c, GGML_TYPE_Q6_K, d.n_ff, d.n_embd);
    M.kc = ggml_new_tensor_1d(c, GGML_TYPE_F16, (int64_t) n_embd_kv * d.n_ctx);      // [n_embd_kv, n_ctx]: one row per position
    M.vc = ggml_new_tensor_1d(c, GGML_TYPE_F16, (int64_t) n_embd_kv * d.n_ctx);      // transposed: [n_ctx, n_embd_kv]
    M.buf = ggml_backend_alloc_ctx_tensors(c, backend);
    return M;
}

static Step build_step(const Dims & d, const Model & M, int n_past, int n_tokens) {
    Step S;
    ggml_init_params ip = { ggml_tensor_overhead() * 256 + ggml_graph_overhead(), nullptr, true };
    ggml_context * c = S.ctx = ggml_init(ip);
    const int n_embd_kv = d.n_head_kv * d.hd;
    const int n_kv = S.n_kv = GGML_PAD(n_past + n_tokens, d.fa ? 256 : 32);            // the attended window is padded (masked beyond the data); -fa pads to 256
    S.x    = ggml_new_tensor_2d(c, GGML_TYPE_F32, d.n_embd, n_tokens);  ggml_set_input(S.x);
    S.pos  = ggml_new_tensor_1d(c, GGML_TYPE_I32, n_tokens);            ggml_set_input(S.pos);
    S.mask = ggml_new_tensor_2d(c, d.fa ? GGML_TYPE_F16 : GGML_TYPE_F32, n_kv, GGML_PAD(n_tokens, d.fa ? 64 : 32)); ggml_set_input(S.mask);   // (-fa: f16 mask, llama-graph.cpp:1211-1232)

    const float eps = 1e-5f, kq_scale = 1.0f / sqrtf((float) d.hd);
    // --- attention norm, projections, rope (llm_build_llama)
    ggml_tensor * cur = ggml_mul(c, ggml_rms_norm(c, S.x, eps), M.attn_norm);
    ggml_tensor * Q = ggml_mul_mat(c, M.wq, cur), * K = ggml_mul_mat(c, M.wk, cur), * V = ggml_mul_mat(c, M.wv, cur);
    Q = ggml_rope_ext(c, ggml_reshape_3d(c, Q, d.hd, d.n_head, n_tokens), S.pos, nullptr, d.hd, 0, 8192, 500000.0f, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f);
    K = ggml_rope_ext(c, ggml_reshape_3d(c, K, d.hd, d.n_head_kv, n_tokens), S.pos, nullptr, d.hd, 0, 8192, 500000.0f, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f);
    // --- build_attn (src/llama-graph.cpp:1384-1388): q, k, v enter the graph together, before the cache stores
    S.gf = ggml_new_graph(c);
    ggml_build_forward_expand(S.gf, Q); ggml_build_forward_expand(S.gf, K); ggml_build_forward_expand(S.gf, V);
    // --- store k, v in the cache (llama_kv_cache_unified cpy_k / cpy_v; V transposed when flash attention is off)
    ggml_tensor * k_view = ggml_view_1d(c, M.kc, (int64_t) n_tokens * n_embd_kv, ggml_row_size(GGML_TYPE_F16, n_embd_kv) * n_past);
    ggml_build_forward_expand(S.gf, ggml_cpy(c, ggml_reshape_2d(c, K, n_embd_kv, n_tokens), k_view));
    ggml_tensor * q = ggml_permute(c, Q, 0, 2, 1, 3);
    ggml_tensor * k = ggml_view_3d(c, M.kc, d.hd, n_kv, d.n_head_kv, ggml_row_size(GGML_TYPE_F16, n_embd_kv), ggml_row_size(GGML_TYPE_F16, d.hd), 0);
    if (d.fa) {
        // --- -fa 1 (build_attn_mha, llama-graph.cpp:1211-1232): V is stored like K (one row per position) and one FLASH_ATTN_EXT node
        // replaces KQ, SOFT_MAX, KQV and the CONT
        ggml_tensor * v_view = ggml_view_1d(c, M.vc, (int64_t) n_tokens * n_embd_kv, ggml_row_size(GGML_TYPE_F16, n_embd_kv) * n_past);
        ggml_build_forward_expand(S.gf, ggml_cpy(c, ggml_reshape_2d(c, V, n_embd_kv, n_tokens), v_view));
        ggml_tensor * v = ggml_view_3d(c, M.vc, d.hd, n_kv, d.n_head_kv, ggml_row_size(GGML_TYPE_F16, n_embd_kv), ggml_row_size(GGML_TYPE_F16, d.hd), 0);
        cur = ggml_flash_attn_ext(c, q, k, v, S.mask, kq_scale, 0.0f, 0.0f);
        ggml_flash_attn_ext_set_prec(cur, GGML_PREC_F32);
        cur = ggml_reshape_2d(c, cur, d.n_embd, n_tokens);
    } else {
        ggml_tensor * v_view = ggml_view_2d(c, M.vc, n_tokens, n_embd_kv, d.n_ctx * ggml_element_size(M.vc), n_past * ggml_element_size(M.vc));
        ggml_build_forward_expand(S.gf, ggml_cpy(c, ggml_transpose(c, ggml_reshape_2d(c, V, n_embd_kv, n_tokens)), v_view));
        // --- attention (build_attn_mha)
        ggml_tensor * kq = ggml_mul_mat(c, k, q);
        kq = ggml_soft_max_ext(c, kq, S.mask, kq_scale, 0.0f);
        ggml_tensor * v = ggml_view_3d(c, M.vc, n_kv, d.hd, d.n_head_kv, ggml_element_size(M.vc) * d.n_ctx, ggml_element_size(M.vc) * d.n_ctx * d.hd, 0);
        ggml_tensor * kqv = ggml_mul_mat(c, v, kq);
        cur = ggml_cont_2d(c, ggml_permute(c, kqv, 0, 2, 1, 3), d.n_embd, n_tokens);
    }
    cur = ggml_mul_mat(c, M.wo, cur);
    // --- residual, ffn (build_ffn LLM_FFN_SILU, LLM_FFN_PAR)
    ggml_tensor * ffn_inp = ggml_add(c, cur, S.x);
    cur = ggml_mul(c, ggml_rms_norm(c, ffn_inp, eps), M.ffn_norm);
    ggml_tensor * gate = ggml_silu(c, ggml_mul_mat(c, M.wgate, cur));
    cur = ggml_mul(c, gate, ggml_mul_mat(c, M.wup, cur));
    cur = ggml_mul_mat(c, M.wdown, cur);
    S.out = ggml_add(c, cur, ffn_inp);
    ggml_set_output(S.out);
    ggml_build_forward_expand(S.gf, S.out);
    return S;
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
    const bool big = argc > 3 && std::string(argv[3]) == "8b";
    const int steps = argc > 4 ? atoi(argv[4]) : 1;
    const int iters = argc > 5 ? atoi(argv[5]) : 0;
    const bool trace = getenv("LAYER_PARITY_TRACE") != nullptr;       // per-node comparison on the steps that differ
    ggml_backend_load_all();
    ggml_backend_dev_t dev = ggml_backend_dev_by_name(dev_name.c_str());
    if (!dev) { fprintf(stderr, "device %s not found (is GGML_BACKEND_PATH set?)\n", dev_name.c_str()); return 3; }
    ggml_backend_t be_dev = ggml_backend_dev_init(dev, nullptr);
    ggml_backend_t be_cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    Dims d;
    d.fa = getenv("LAYER_PARITY_FA") != nullptr;               // the -fa 1 form of the layer (FLASH_ATTN_EXT, V cache not transposed)
    if (big) { d.n_embd = 4096; d.n_head = 32; d.n_head_kv = 8; d.hd = 128; d.n_ff = 14336; d.n_ctx = 1024; d.n_past = 500; }
    if (d.fa && d.n_ctx < 512) d.n_ctx = 512;
    const int jitter = getenv("LAYER_PARITY_JITTER") ? atoi(getenv("LAYER_PARITY_JITTER")) : -1;
    if (d.n_past + steps * (n_tokens + (jitter >= 0 ? 1 : 0)) > d.n_ctx) { fprintf(stderr, "too many steps for n_ctx\n"); return 3; }
    Model mr = make_model(d, be_cpu), mt = make_model(d, be_dev);
    ggml_gallocr_t ga_cpu = ggml_gallocr_new(ggml_backend_get_default_buffer_type(be_cpu));
    ggml_gallocr_t ga_dev = ggml_gallocr_new(ggml_backend_get_default_buffer_type(be_dev));

    // ---- the same weights and initial cache contents for both
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    auto randv = [&](size_t n, float s) { std::vector<float> v(n); for (auto & x : v) x = nd(rng) * s; return v; };
    const int n_embd_kv = d.n_head_kv * d.hd;
    auto set_both = [&](ggml_tensor * a, ggml_tensor * b, const void * data, size_t bytes) {
        ggml_backend_tensor_set(a, data, 0, bytes); ggml_backend_tensor_set(b, data, 0, bytes);
    };
    { auto w = randv(d.n_embd, 0.1f); for (auto & x : w) x += 1.0f; set_both(mr.attn_norm, mt.attn_norm, w.data(), w.size() * 4); }
    { auto w = randv(d.n_embd, 0.1f); for (auto & x : w) x += 1.0f; set_both(mr.ffn_norm, mt.ffn_norm, w.data(), w.size() * 4); }
    struct WQ { ggml_tensor * a, * b; int64_t k, m; };
    for (WQ w : { WQ{mr.wq, mt.wq, d.n_embd, d.n_embd}, WQ{mr.wk, mt.wk, d.n_embd, n_embd_kv}, WQ{mr.wv, mt.wv, d.n_embd, n_embd_kv},
                  WQ{mr.wo, mt.wo, d.n_embd, d.n_embd}, WQ{mr.wgate, mt.wgate, d.n_embd, d.n_ff}, WQ{mr.wup, mt.wup, d.n_embd, d.n_ff},
                  WQ{mr.wdown, mt.wdown, d.n_ff, d.n_embd} }) {
        auto f = randv((size_t) w.k * w.m, 1.0f / sqrtf((float) w.k));
        auto q = quantize(w.a->type, f, w.k, w.m);
        set_both(w.a, w.b, q.data(), q.size());
    }
    for (auto pr : { std::pair<ggml_tensor *, ggml_tensor *>{mr.kc, mt.kc}, {mr.vc, mt.vc} }) {
        auto f = randv((size_t) n_embd_kv * d.n_ctx, 1.0f); std::vector<ggml_fp16_t> h(f.size()); ggml_fp32_to_fp16_row(f.data(), h.data(), f.size());
        set_both(pr.first, pr.second, h.data(), h.size() * 2);
    }
    auto get_f32 = [](ggml_tensor * t) { std::vector<float> v(ggml_nelements(t)); ggml_backend_tensor_get(t, v.data(), 0, v.size() * 4); return v; };
    auto get_f16 = [](ggml_tensor * t) { std::vector<ggml_fp16_t> h(ggml_nelements(t)); ggml_backend_tensor_get(t, h.data(), 0, h.size() * 2);
                                         std::vector<float> v(h.size()); ggml_fp16_to_fp32_row(h.data(), v.data(), h.size()); return v; };

    bool ok = true;
    double worst_out = 0, worst_kv = 0;
    uint64_t digest = 1469598103934665603ull;                    // of every step's device output bits: equal digests = bit-identical runs
    auto mix = [&](const std::vector<float> & v) { for (float f : v) { uint32_t u; memcpy(&u, &f, 4); digest = (digest ^ u) * 1099511628211ull; } };
    Step sr, st;
    int n_past_run = d.n_past;
    const int n_tokens_base = n_tokens;
    for (int t = 0; t < steps; ++t) {
        const int n_tokens = n_tokens_base + ((jitter >= 0 && t >= jitter && ((t - jitter) & 1)) ? 1 : 0);
        const int n_past = n_past_run;
        n_past_run += n_tokens;
        if (sr.ctx) { ggml_free(sr.ctx); ggml_free(st.ctx); }
        sr = build_step(d, mr, n_past, n_tokens); st = build_step(d, mt, n_past, n_tokens);
        if (trace) for (Step * S : { &sr, &st }) for (int i = 0; i < ggml_graph_n_nodes(S->gf); ++i) ggml_set_output(ggml_graph_node(S->gf, i));   // keep every intermediate
        if (!ggml_gallocr_alloc_graph(ga_cpu, sr.gf) || !ggml_gallocr_alloc_graph(ga_dev, st.gf)) { fprintf(stderr, "graph allocation failed\n"); return 4; }
        if (t == 0) {                // residency: does the device backend take every node?
            int unsupported = 0;
            for (int i = 0; i < ggml_graph_n_nodes(st.gf); ++i) {
                ggml_tensor * n = ggml_graph_node(st.gf, i);
                if (!ggml_backend_supports_op(be_dev, n)) { ++unsupported; printf("  NOT SUPPORTED on %s: node %d %s (%s)\n", dev_name.c_str(), i, ggml_op_desc(n), n->name); }
            }
            printf("decode layer graph: %d nodes, %d refused by %s\n", ggml_graph_n_nodes(st.gf), unsupported, dev_name.c_str());
            if (unsupported) return 2;
        }
        { auto x = randv((size_t) d.n_embd * n_tokens, 1.0f); set_both(sr.x, st.x, x.data(), x.size() * 4); }
        { std::vector<int32_t> p(n_tokens); for (int i = 0; i < n_tokens; ++i) p[i] = n_past + i; set_both(sr.pos, st.pos, p.data(), p.size() * 4); }
        { std::vector<float> m((size_t) sr.n_kv * GGML_PAD(n_tokens, d.fa ? 64 : 32), -INFINITY);       // causal mask over the padded window
          for (int i = 0; i < n_tokens; ++i) for (int j = 0; j <= n_past + i; ++j) m[(size_t) i * sr.n_kv + j] = 0.0f;
          if (d.fa) { std::vector<ggml_fp16_t> h(m.size()); ggml_fp32_to_fp16_row(m.data(), h.data(), m.size()); set_both(sr.mask, st.mask, h.data(), h.size() * 2); }
          else set_both(sr.mask, st.mask, m.data(), m.size() * 4); }
        if (ggml_backend_graph_compute(be_cpu, sr.gf) != GGML_STATUS_SUCCESS || ggml_backend_graph_compute(be_dev, st.gf) != GGML_STATUS_SUCCESS) {
            fprintf(stderr, "graph_compute failed\n"); return 4;
        }
        mix(get_f32(st.out));
        const double e_out = nmse(get_f32(st.out), get_f32(sr.out));
        const double e_k = nmse(get_f16(mt.kc), get_f16(mr.kc)), e_v = nmse(get_f16(mt.vc), get_f16(mr.vc));
        if (steps <= 4 || t == 0 || t == steps - 1 || !(e_out <= 5e-4))
            printf("step %d: n_tokens=%d n_past=%d n_kv=%d  NMSE out %.3e  k cache %.3e  v cache %.3e\n", t, n_tokens, n_past, sr.n_kv, e_out, e_k, e_v);
        if (trace && e_out > 1e-9) for (int i = 0; i < ggml_graph_n_nodes(st.gf); ++i) {     // where does the difference enter?
            ggml_tensor * a = ggml_graph_node(st.gf, i), * b = ggml_graph_node(sr.gf, i);
            if (!ggml_is_contiguous(a) || (a->type != GGML_TYPE_F32 && a->type != GGML_TYPE_F16) || a->op == GGML_OP_CPY) continue;
            const double e = a->type == GGML_TYPE_F32 ? nmse(get_f32(a), get_f32(b)) : nmse(get_f16(a), get_f16(b));
            printf("    node %2d %-10s ne [%5lld %5lld %3lld]  NMSE %.3e\n", i, ggml_op_desc(a), (long long) a->ne[0], (long long) a->ne[1], (long long) a->ne[2], e);
        }
        worst_out = std::fmax(worst_out, e_out); worst_kv = std::fmax(worst_kv, std::fmax(e_k, e_v));
        // decode sizes: the projections are exact-integer GEMVs, the cache must agree to f16 rounding flips; more than 8 tokens: the
        // matrix-core tiers (bf16 operands for Q6_K) are held to the reference's op bound
        const double kv_tol = n_tokens > 8 ? 5e-4 : 1e-6;
        ok = ok && e_out <= 5e-4 && e_k <= kv_tol && e_v <= kv_tol && std::isfinite(e_out);
    }
    mix(get_f16(mt.kc)); mix(get_f16(mt.vc));
    printf("%d step(s): worst NMSE out %.3e, kv cache %.3e; device output digest %016llx\n", steps, worst_out, worst_kv, (unsigned long long) digest);
    if (iters > 0) {
        auto time_it = [&](ggml_backend_t be, ggml_cgraph * gf, int n) {
            ggml_backend_graph_compute(be, gf);
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) ggml_backend_graph_compute(be, gf);
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        };
        const double us_dev = time_it(be_dev, st.gf, iters), us_cpu = time_it(be_cpu, sr.gf, iters < 10 ? iters : 10);
        printf("one decoder layer (n_embd %d, n_ff %d, n_tokens=%d, n_kv=%d, %d graph nodes) through ggml_backend_graph_compute: %s %.1f us, CPU backend %.1f us\n",
               d.n_embd, d.n_ff, n_tokens, sr.n_kv, ggml_graph_n_nodes(st.gf), dev_name.c_str(), us_dev, us_cpu);
    }
    printf("%s\n", ok ? "LAYER PARITY OK" : "LAYER PARITY FAILED");
    ggml_backend_free(be_dev);       // (prints the backend's launch-graph statistics when MI355_GRAPH_STATS is set)
    return ok ? 0 : 1;
}

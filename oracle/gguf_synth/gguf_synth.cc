// gguf_synth.cc -- TEST / MEASUREMENT INFRASTRUCTURE (built into oracle/_ref/<variant>/ against the reference's public gguf / ggml API).
//
// Writes a synthetic llama-architecture GGUF model file that the reference's unmodified libllama loads and llama-bench times
// (SURVEY.md section 8d "Synthetic inputs"): exact tensor shapes of the named model, the tensor-type mix llama-quantize would
// produce for the file type (src/llama-quant.cpp:129-131 use_more_bits, :151-168 output, :235-255 attn_v, :291-297 ffn_down,
// :316-322 attn_output), random weights quantized by the reference's own quantizer (ggml_quantize_chunk), the `no_vocab`
// tokenizer (src/llama-vocab.cpp:1373-1391 creates vocab_size dummy tokens).  No network, no real checkpoint.
//
//   gguf_synth --preset 8b|70b|mixtral|small|tiny --ftype q4_k_m|q8_0|q5_k_m|iq4_xs [--layers N] [--vocab V] [--ctx C] --out FILE
//
// Weights: quantizing 8-70 G random floats on the host would take minutes, so every (tensor class, type) is quantized ONCE from a
// seeded slab (<= 2048 rows) and the slab is tiled over the rows of every tensor of that class.  llama-bench feeds random token ids
// and measures time; the values only have to be finite and of a sane scale.  (Parity is pinned elsewhere: tests/, oracle/model_parity.)
// The data section is streamed to the file tensor by tensor (the gguf API would hold a second copy of the whole model in memory).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ggml.h"
#include "gguf.h"

struct Dims { int n_embd, n_head, n_head_kv, n_ff, n_layer, n_vocab, n_ctx, n_expert, n_used; float rope_base; };

static bool more_bits(int il, int n) { return il < n / 8 || il >= 7 * n / 8 || (il - n / 8) % 3 == 2; }   // src/llama-quant.cpp:129-131

struct Recipe { ggml_type base, more, output, attn_v_rest, attn_output_moe, kv_moe; bool bump; };

struct Slab { std::vector<uint8_t> q; int64_t rows; };
static std::map<std::string, Slab> g_slabs;

// seeded slab of `rows` x k weights, uniform with variance 1/k (xorshift64), quantized by the reference
static const Slab & slab_for(ggml_type t, int64_t k, int64_t m, uint64_t salt) {
    const int64_t rows = m < 2048 ? m : 2048;
    const std::string key = std::string(ggml_type_name(t)) + ":" + std::to_string(k) + ":" + std::to_string(rows) + ":" + std::to_string(salt);
    auto it = g_slabs.find(key);
    if (it != g_slabs.end()) return it->second;
    std::vector<float> f((size_t) k * rows);
    uint64_t s = 88172645463325252ull + salt * 0x9E3779B97F4A7C15ull;
    const float sc = 1.7320508f / sqrtf((float) k);
    for (auto & x : f) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = ((float) (uint32_t) (s >> 40) * (2.0f / 16777216.0f) - 1.0f) * sc; }
    Slab sl; sl.rows = rows;
    if (t == GGML_TYPE_F32) { sl.q.resize(f.size() * 4); memcpy(sl.q.data(), f.data(), sl.q.size()); }
    else { sl.q.resize(ggml_row_size(t, k) * rows); ggml_quantize_chunk(t, f.data(), sl.q.data(), 0, rows, k, nullptr); }
    return g_slabs.emplace(key, std::move(sl)).first->second;
}

struct TensorPlan { std::string name; ggml_type type; int64_t ne[3]; int nd; int kind; uint64_t salt; };   // kind 0: matrix (tiled slab), 1: norm vector (1 + small noise)

int main(int argc, char ** argv) {
    std::string preset = "tiny", ftype = "q4_k_m", out;
    int layers = -1, vocab = -1, ctx = -1;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? std::string(argv[++i]) : std::string(); };
        if (a == "--preset") preset = next(); else if (a == "--ftype") ftype = next(); else if (a == "--out") out = next();
        else if (a == "--layers") layers = atoi(next().c_str()); else if (a == "--vocab") vocab = atoi(next().c_str()); else if (a == "--ctx") ctx = atoi(next().c_str());
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 3; }
    }
    if (out.empty()) { fprintf(stderr, "usage: gguf_synth --preset 8b|70b|mixtral|small|tiny --ftype q4_k_m|q8_0|q5_k_m|iq4_xs [--layers N] [--vocab V] --out FILE\n"); return 3; }
    Dims d;
    if (preset == "8b")           d = { 4096, 32, 8, 14336, 32, 128256, 8192, 0, 0, 500000.0f };
    else if (preset == "70b")     d = { 8192, 64, 8, 28672, 80, 128256, 8192, 0, 0, 500000.0f };
    else if (preset == "mixtral") d = { 4096, 32, 8, 14336, 32, 32000, 32768, 8, 2, 1000000.0f };
    else if (preset == "small")   d = { 2048, 16, 4, 4096, 4, 32000, 4096, 0, 0, 500000.0f };
    else if (preset == "tiny")    d = { 512, 4, 2, 1024, 2, 4096, 2048, 0, 0, 500000.0f };
    else { fprintf(stderr, "unknown preset\n"); return 3; }
    if (layers > 0) d.n_layer = layers;
    if (vocab > 0) d.n_vocab = vocab;
    if (ctx > 0) d.n_ctx = ctx;
    const int hd = d.n_embd / d.n_head;

    // file type -> tensor types (llama_tensor_get_type, src/llama-quant.cpp)
    Recipe R; int file_type;
    if (ftype == "q4_k_m")      { R = { GGML_TYPE_Q4_K, GGML_TYPE_Q6_K, GGML_TYPE_Q6_K, GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q8_0, true }; file_type = 15; }
    else if (ftype == "q5_k_m") { R = { GGML_TYPE_Q5_K, GGML_TYPE_Q6_K, GGML_TYPE_Q6_K, GGML_TYPE_Q5_K, GGML_TYPE_Q5_K, GGML_TYPE_Q8_0, true }; file_type = 17; }
    else if (ftype == "q8_0")   { R = { GGML_TYPE_Q8_0, GGML_TYPE_Q8_0, GGML_TYPE_Q8_0, GGML_TYPE_Q8_0, GGML_TYPE_Q8_0, GGML_TYPE_Q8_0, false }; file_type = 7; }
    else if (ftype == "iq4_xs") { R = { GGML_TYPE_IQ4_XS, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K, GGML_TYPE_IQ4_XS, GGML_TYPE_IQ4_XS, GGML_TYPE_Q8_0, true }; file_type = 30; }
    else { fprintf(stderr, "unknown ftype\n"); return 3; }
    const bool is70 = d.n_layer == 80 && d.n_head != d.n_head_kv;           // the "70B" rule of llama-quant.cpp:238-243 (n_gqa == 8 with 80 layers): attn_v Q4_K -> Q5_K
    if (ftype == "q4_k_m" && is70) R.attn_v_rest = GGML_TYPE_Q5_K;

    std::vector<TensorPlan> plan;
    auto mat = [&](const std::string & n, ggml_type t, int64_t k, int64_t m, int64_t e, uint64_t salt) {
        TensorPlan p; p.name = n; p.type = t; p.ne[0] = k; p.ne[1] = m; p.ne[2] = e; p.nd = e > 1 ? 3 : 2; p.kind = 0; p.salt = salt; plan.push_back(p); };
    auto vec = [&](const std::string & n, int64_t k) { TensorPlan p; p.name = n; p.type = GGML_TYPE_F32; p.ne[0] = k; p.ne[1] = p.ne[2] = 1; p.nd = 1; p.kind = 1; p.salt = 0; plan.push_back(p); };
    mat("token_embd.weight", R.base == GGML_TYPE_Q8_0 ? GGML_TYPE_Q8_0 : (ftype == "iq4_xs" ? GGML_TYPE_IQ4_XS : R.base), d.n_embd, d.n_vocab, 1, 100);
    const int n_kv_embd = d.n_head_kv * hd;
    for (int il = 0; il < d.n_layer; ++il) {
        const std::string b = "blk." + std::to_string(il) + ".";
        const bool mb = R.bump && more_bits(il, d.n_layer);
        const bool moe = d.n_expert > 0;
        vec(b + "attn_norm.weight", d.n_embd);
        mat(b + "attn_q.weight", R.base, d.n_embd, d.n_embd, 1, 1);
        mat(b + "attn_k.weight", moe && R.bump ? R.kv_moe : R.base, d.n_embd, n_kv_embd, 1, 2);      // llama-quant.cpp:244-255: 8-expert models keep attn_k / attn_v at Q8_0
        mat(b + "attn_v.weight", moe && R.bump ? R.kv_moe : (mb ? R.more : R.attn_v_rest), d.n_embd, n_kv_embd, 1, 3);
        mat(b + "attn_output.weight", moe && R.bump ? R.attn_output_moe : R.base, d.n_embd, d.n_embd, 1, 4);   // :316-322
        vec(b + "ffn_norm.weight", d.n_embd);
        if (moe) {
            { TensorPlan p; p.name = b + "ffn_gate_inp.weight"; p.type = GGML_TYPE_F32; p.ne[0] = d.n_embd; p.ne[1] = d.n_expert; p.ne[2] = 1; p.nd = 2; p.kind = 0; p.salt = 9; plan.push_back(p); }
            mat(b + "ffn_gate_exps.weight", R.base, d.n_embd, d.n_ff, d.n_expert, 5);
            mat(b + "ffn_down_exps.weight", mb ? R.more : R.base, d.n_ff, d.n_embd, d.n_expert, 7);
            mat(b + "ffn_up_exps.weight", R.base, d.n_embd, d.n_ff, d.n_expert, 6);
        } else {
            mat(b + "ffn_gate.weight", R.base, d.n_embd, d.n_ff, 1, 5);
            mat(b + "ffn_down.weight", mb ? R.more : R.base, d.n_ff, d.n_embd, 1, 7);
            mat(b + "ffn_up.weight", R.base, d.n_embd, d.n_ff, 1, 6);
        }
    }
    vec("output_norm.weight", d.n_embd);
    mat("output.weight", R.output, d.n_embd, d.n_vocab, 1, 8);

    gguf_context * g = gguf_init_empty();
    gguf_set_val_str(g, "general.architecture", "llama");
    gguf_set_val_str(g, "general.name", ("synthetic-" + preset + "-" + ftype).c_str());
    gguf_set_val_u32(g, "general.file_type", (uint32_t) file_type);
    gguf_set_val_u32(g, "llama.context_length", (uint32_t) d.n_ctx);
    gguf_set_val_u32(g, "llama.embedding_length", (uint32_t) d.n_embd);
    gguf_set_val_u32(g, "llama.block_count", (uint32_t) d.n_layer);
    gguf_set_val_u32(g, "llama.feed_forward_length", (uint32_t) d.n_ff);
    gguf_set_val_u32(g, "llama.attention.head_count", (uint32_t) d.n_head);
    gguf_set_val_u32(g, "llama.attention.head_count_kv", (uint32_t) d.n_head_kv);
    gguf_set_val_f32(g, "llama.attention.layer_norm_rms_epsilon", 1e-5f);
    gguf_set_val_f32(g, "llama.rope.freq_base", d.rope_base);
    gguf_set_val_u32(g, "llama.rope.dimension_count", (uint32_t) hd);
    gguf_set_val_u32(g, "llama.vocab_size", (uint32_t) d.n_vocab);
    if (d.n_expert > 0) { gguf_set_val_u32(g, "llama.expert_count", (uint32_t) d.n_expert); gguf_set_val_u32(g, "llama.expert_used_count", (uint32_t) d.n_used); }
    gguf_set_val_str(g, "tokenizer.ggml.model", "no_vocab");

    ggml_init_params ip = { ggml_tensor_overhead() * (plan.size() + 8), nullptr, true };
    ggml_context * c = ggml_init(ip);
    std::vector<ggml_tensor *> ts;
    for (const TensorPlan & p : plan) {
        ggml_tensor * t = p.nd == 1 ? ggml_new_tensor_1d(c, p.type, p.ne[0]) : p.nd == 2 ? ggml_new_tensor_2d(c, p.type, p.ne[0], p.ne[1]) : ggml_new_tensor_3d(c, p.type, p.ne[0], p.ne[1], p.ne[2]);
        ggml_set_name(t, p.name.c_str());
        gguf_add_tensor(g, t);
        ts.push_back(t);
    }
    // meta first (header, key-values, tensor infos, padded to the alignment), then the data section streamed in tensor order
    if (!gguf_write_to_file(g, out.c_str(), /*only_meta =*/ true)) return 1;
    FILE * f = fopen(out.c_str(), "ab");
    if (!f) { fprintf(stderr, "cannot append to %s\n", out.c_str()); return 1; }
    const size_t align = gguf_get_alignment(g);
    const std::vector<uint8_t> zeros(align, 0);
    size_t total = 0;
    for (size_t i = 0; i < plan.size(); ++i) {
        const TensorPlan & p = plan[i];
        const size_t nbytes = ggml_nbytes(ts[i]);
        if (gguf_get_tensor_offset(g, (int64_t) i) != total) { fprintf(stderr, "offset mismatch at %s\n", p.name.c_str()); return 1; }
        if (p.kind == 1) {
            std::vector<float> w((size_t) p.ne[0]);
            uint64_t s = 1234567ull + i;
            for (auto & x : w) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = 1.0f + 0.1f * ((float) (uint32_t) (s >> 40) * (2.0f / 16777216.0f) - 1.0f); }
            if (fwrite(w.data(), 1, nbytes, f) != nbytes) return 1;
        } else {
            const int64_t rows_total = p.ne[1] * p.ne[2];
            const Slab & sl = slab_for(p.type, p.ne[0], p.ne[1], p.salt);
            const size_t rb = ggml_row_size(p.type, p.ne[0]);
            for (int64_t r = 0; r < rows_total; r += sl.rows) {
                const int64_t nr = rows_total - r < sl.rows ? rows_total - r : sl.rows;
                if (fwrite(sl.q.data(), 1, rb * nr, f) != rb * nr) return 1;
            }
        }
        const size_t pad = GGML_PAD(nbytes, align) - nbytes;
        if (pad && fwrite(zeros.data(), 1, pad, f) != pad) return 1;
        total += nbytes + pad;
    }
    fclose(f);
    fprintf(stderr, "gguf_synth: %s  %zu tensors  %.2f GiB\n", out.c_str(), plan.size(), (double) total / (1024.0 * 1024.0 * 1024.0));
    ggml_free(c); gguf_free(g);
    return 0;
}

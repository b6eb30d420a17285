// ggml-mi355.cpp -- ggml backend plugin "MI355" (libggml-mi355.so).
//
// A dynamically loadable ggml backend (GGML_BACKEND_DL: exports ggml_backend_init / ggml_backend_score,
// ggml/src/ggml-backend-impl.h:215-251) that the reference's unmodified llama-bench / test-backend-ops
// pick up through GGML_BACKEND_PATH (ggml/src/ggml-backend-reg.cpp:585-589).  It implements the four
// vtables of ggml-backend-impl.h (reg :191-207, device :137-185, buffer type :17-35, buffer :41-66,
// backend :87-124) in plain C++ over the C-ABI of libmi355q.so (include/mi355q.h): no HIP code here.
//
// Scope: the quantized MUL_MAT / MUL_MAT_ID hot path and, so that a decode graph stays resident on the
// device, the f32 / f16 ops between those matmuls.  supports_op is true for
//   * GGML_OP_MUL_MAT    with quantized src0 (the 19 types of mi355q_type_supported) or f16 / f32 src0, f32 src1, f32 dst
//   * GGML_OP_MUL_MAT_ID with the same quantized src0 types (expert ids stay on the device)
//   * ADD SUB MUL DIV SCALE, UNARY (SILU RELU SIGMOID TANH NEG ABS), RMS_NORM, CPY CONT DUP (f32 <-> f16), GET_ROWS, SOFT_MAX, ROPE,
//     FLASH_ATTN_EXT (f16 K / V), ARGSORT, SUM_ROWS -- the variants a llama / mixtral graph uses (mi355_supports_op lists them)
//   * the no-op view ops NONE / RESHAPE / VIEW / PERMUTE / TRANSPOSE on our own buffers
// everything else stays on the CPU backend.  The fork's own DSP backend (ggml/src/ggml-dsp/ggml-dsp.cpp)
// is the structural template for where each hook goes; none of its code is reused.
//
// graph_compute (ggml_backend_i.graph_compute, :109): a one-token decode graph is matched against the llama decode patterns and compiled
// into a decode plan -- ONE persistent kernel launch per token (decode-plan.inc, mi355q_plan_*); plans are cached per graph key.  Graphs
// the matcher declines are issued node by node with fused groups, captured into a launch graph at their second sighting and replayed.
// Asynchronous interface: set/get/cpy_tensor_async, event_record / event_wait, device events, a pinned host buffer type; cpy_tensor_async
// between two MI355 devices is a peer copy ordered by an event (the layer-split hop of ggml_backend_sched).
//
// Device layout of quantized tensors: set_tensor converts canonical ggml rows into the planar device rows
// of libmi355q (same row size and stride; only the byte order inside a row changes), get_tensor converts
// back, so buffer contents stay opaque-but-round-trippable as the buffer interface requires.
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#include "ggml-impl.h"

#include "mi355q.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <chrono>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#define MI355_MAX_DEVICES 16

#define MQ_CHECK(expr)                                                                                             \
    do {                                                                                                           \
        const int rc_ = (expr);                                                                                    \
        if (rc_ != MI355Q_OK) GGML_ABORT("MI355: %s failed (%d): %s", #expr, rc_, mi355q_last_error());            \
    } while (0)

// ------------------------------------------------------------------------------------------------ contexts
struct mi355_plan_entry {               // a decode plan cached per graph key (decode-plan.inc)
    uint64_t key = 0, key2 = 0;         // two independent 64-bit hashes of the node list; a hit needs both and the node count
    int n_nodes = 0;
    mi355q_plan * plan = nullptr;
    std::vector<int> pre, post;         // node indices issued eagerly before / after the launch
    int n_stages = 0;
};
struct mi355_device_ctx {
    int         index;         // HIP device ordinal (MI355_DUP_DEVICES exposes one physical device under several names: same ordinal)
    std::string name;          // "MI355_0"
    std::string description;
    struct ggml_backend_buffer_type buft;
    std::string buft_name;
    struct ggml_backend_buffer_type host_buft;     // pinned host memory (get_host_buffer_type)
    std::string host_buft_name;
};

struct mi355_buffer_ctx {
    int    device;
    void * base;
};

struct mi355_backend_ctx {
    int         device;
    std::string name;
    void *      stream    = nullptr;
    void *      copy_event = nullptr;   // orders a peer copy issued on THIS backend's stream before the destination backend's stream
    void *      workspace = nullptr;
    size_t      workspace_size = 0;
    // launch-graph cache of the last compute graph (the reference's counterpart: ggml-cuda.cu:2470-2781)
    mi355q_graph * graph = nullptr;
    uint64_t    graph_key = 0;          // hash of the node list the cached / last seen graph was built from
    int         key_repeats = 0;        // how many consecutive times that key has been seen
    int         key_changes = 0;        // consecutive key changes: after a few the cache is given up for this backend
    bool        graphs_disabled = false;
    void **     dest_table = nullptr;   // device array: destination base pointers of the CPY nodes (updated every compute)
    bool        dest_valid = false;     // the table holds THIS call's pointers (set by graph_compute after the upload, cleared on return)
    std::vector<void *> dest_host;
    void **     dest_pinned = nullptr;  // pinned staging copy of the table: the per-token upload is one asynchronous copy from here
    bool        capturing = false;
    long        n_eager = 0, n_captured = 0, n_replayed = 0;   // graph_compute calls by how they ran (MI355_GRAPH_STATS=1 prints them)
    long        n_fused_norm = 0, n_fused_mats = 0, n_fused_act = 0, n_elided_cont = 0, n_fused_add = 0;   // launches saved by the fusions of mi355_issue_nodes
    // decode plans (decode-plan.inc): the N = 1 graph as one persistent launch, cached per graph key
    std::vector<struct mi355_plan_entry *> plans;
    uint64_t    plan_declined[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; int plan_declined_n = 0;   // graphs the matcher declined (alternating scheduler splits, MoE): not matched again
    long        test_inject_abort = -1; // MI355_TEST_INJECT_PLAN_ABORT=N: the plugin raises the abort word itself after N planned calls (tests the recovery path)
    double      t_phase[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; long n_timed = 0;   // MI355_TIMING=1: host microseconds of graph_compute by phase
    unsigned *  plan_abort = nullptr;   // pinned host word: != 0 once a plan's poll timed out
    bool        plans_disabled = false;
    long        n_planned = 0, n_plans_built = 0;
};
constexpr int MI355_MAX_CPY_DESTS = 4096;

static ggml_guid_t mi355_guid() {
    static ggml_guid guid = { 0x4d, 0x49, 0x33, 0x35, 0x35, 0x58, 0x2d, 0x71, 0x6d, 0x61, 0x74, 0x6d, 0x75, 0x6c, 0x2d, 0x31 };
    return &guid;
}

static bool mi355_is_quant(enum ggml_type t) { return mi355q_type_supported((int) t) == 1; }

// the tensor whose shape defines the device row layout (views share their source's rows)
static const struct ggml_tensor * mi355_root(const struct ggml_tensor * t) { return t->view_src ? t->view_src : t; }

// A Q8_0 / Q4_0 tensor outside a weight buffer is (or may become) a KV cache (-ctk q8_0 / q4_0, -ctv ...): the device writes its rows block by block (CPY) and the
// attention kernel reads heads out of them, so it keeps the CANONICAL block order; only Q8_0 tensors in buffers the application marked as weights are planar.
static bool mi355_q80_canonical(const struct ggml_tensor * t) {
    const struct ggml_tensor * r = mi355_root(t);
    return (t->type == GGML_TYPE_Q8_0 || t->type == GGML_TYPE_Q4_0) && !(r->buffer && r->buffer->usage == GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
}
// rows of a quantized tensor are stored planar iff (type, ne0, and for Q8_0 the buffer's usage) say so -- same predicate everywhere
static bool mi355_rows_planar(const struct ggml_tensor * t) {
    return mi355_is_quant(t->type) && mi355q_weights_are_planar((int) t->type, mi355_root(t)->ne[0]) == 1 && !mi355_q80_canonical(t);
}

// ------------------------------------------------------------------------------------------------ buffer
static void mi355_buffer_free(ggml_backend_buffer_t buffer) {
    mi355_buffer_ctx * ctx = (mi355_buffer_ctx *) buffer->context;
    mi355q_set_device(ctx->device);
    if (ctx->base) mi355q_free(ctx->base);
    delete ctx;
}

static void * mi355_buffer_get_base(ggml_backend_buffer_t buffer) { return ((mi355_buffer_ctx *) buffer->context)->base; }

static enum ggml_status mi355_buffer_init_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor) {
    GGML_UNUSED(buffer); GGML_UNUSED(tensor);
    return GGML_STATUS_SUCCESS;
}

static void mi355_buffer_memset_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor, uint8_t value, size_t offset, size_t size) {
    mi355_buffer_ctx * ctx = (mi355_buffer_ctx *) buffer->context;
    mi355q_set_device(ctx->device);
    MQ_CHECK(mi355q_memset((char *) tensor->data + offset, value, size, nullptr));
    MQ_CHECK(mi355q_device_synchronize());
}

// canonical host bytes [offset, offset+size) of `tensor` <-> device.  Quantized planar rows are converted in
// whole rows; a range that is not row-aligned is widened to whole rows (read-modify-write on upload).
static void mi355_buffer_transfer(mi355_buffer_ctx * ctx, const struct ggml_tensor * tensor, void * host, size_t offset, size_t size, bool upload) {
    mi355q_set_device(ctx->device);
    char * dev = (char *) tensor->data;
    if (!mi355_rows_planar(tensor)) {
        if (upload) MQ_CHECK(mi355q_memcpy_h2d(dev + offset, host, size, nullptr));
        else        MQ_CHECK(mi355q_memcpy_d2h(host, dev + offset, size, nullptr));
        return;
    }
    // The byte range is addressed in MEMORY order (that is what set/get on a permuted view means, cf.
    // tests/test-backend-ops.cpp init of permuted src0): rows are the root tensor's rows, wherever the view points.
    const int64_t k    = mi355_root(tensor)->ne[0];
    const size_t  rb   = (size_t) mi355q_row_size((int) tensor->type, k);
    char *        base = (char *) mi355_root(tensor)->data;
    const size_t  beg  = (size_t) (dev - base) + offset, end = beg + size;
    const size_t  r0 = beg / rb, r1 = (end + rb - 1) / rb;
    if (beg % rb == 0 && size % rb == 0) {
        if (upload) MQ_CHECK(mi355q_weights_upload((int) tensor->type, base + beg, host, (int64_t) (size / rb), k, nullptr));
        else        MQ_CHECK(mi355q_weights_download((int) tensor->type, host, base + beg, (int64_t) (size / rb), k, nullptr));
        return;
    }
    std::vector<char> rows((r1 - r0) * rb);
    MQ_CHECK(mi355q_weights_download((int) tensor->type, rows.data(), base + r0 * rb, (int64_t) (r1 - r0), k, nullptr));
    if (upload) {
        memcpy(rows.data() + (beg - r0 * rb), host, size);
        MQ_CHECK(mi355q_weights_upload((int) tensor->type, base + r0 * rb, rows.data(), (int64_t) (r1 - r0), k, nullptr));
    } else {
        memcpy(host, rows.data() + (beg - r0 * rb), size);
    }
}

static void mi355_buffer_set_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    mi355_buffer_transfer((mi355_buffer_ctx *) buffer->context, tensor, (void *) data, offset, size, true);
}

static void mi355_buffer_get_tensor(ggml_backend_buffer_t buffer, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    mi355_buffer_transfer((mi355_buffer_ctx *) buffer->context, tensor, data, offset, size, false);
}

static bool mi355_buffer_is_ours(ggml_backend_buffer_t buffer) { return buffer && buffer->iface.get_base == mi355_buffer_get_base; }

static bool mi355_buffer_cpy_tensor(ggml_backend_buffer_t buffer, const struct ggml_tensor * src, struct ggml_tensor * dst) {
    // device-to-device only when both sides keep the same byte image (same type and row length)
    if (!mi355_buffer_is_ours(src->buffer)) return false;
    if (src->type != dst->type || !ggml_is_contiguous(src) || !ggml_is_contiguous(dst) || ggml_nbytes(src) != ggml_nbytes(dst)) return false;
    if (mi355_is_quant(src->type) && (mi355_root(src)->ne[0] != mi355_root(dst)->ne[0] || mi355_rows_planar(src) != mi355_rows_planar(dst))) return false;   // (a Q8_0 tensor is planar in a weight buffer only)
    mi355_buffer_ctx * ctx = (mi355_buffer_ctx *) buffer->context;
    mi355q_set_device(ctx->device);
    MQ_CHECK(mi355q_memcpy_d2d(dst->data, src->data, ggml_nbytes(src), nullptr));
    MQ_CHECK(mi355q_device_synchronize());
    return true;
}

static void mi355_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value) {
    mi355_buffer_ctx * ctx = (mi355_buffer_ctx *) buffer->context;
    mi355q_set_device(ctx->device);
    MQ_CHECK(mi355q_memset(ctx->base, value, buffer->size, nullptr));
    MQ_CHECK(mi355q_device_synchronize());
}

static const struct ggml_backend_buffer_i mi355_buffer_iface = {
    /* .free_buffer   = */ mi355_buffer_free,
    /* .get_base      = */ mi355_buffer_get_base,
    /* .init_tensor   = */ mi355_buffer_init_tensor,
    /* .memset_tensor = */ mi355_buffer_memset_tensor,
    /* .set_tensor    = */ mi355_buffer_set_tensor,
    /* .get_tensor    = */ mi355_buffer_get_tensor,
    /* .cpy_tensor    = */ mi355_buffer_cpy_tensor,
    /* .clear         = */ mi355_buffer_clear,
    /* .reset         = */ nullptr,
};

// ------------------------------------------------------------------------------------------------ buffer type
static const char * mi355_buft_get_name(ggml_backend_buffer_type_t buft) { return ((mi355_device_ctx *) buft->device->context)->buft_name.c_str(); }

static ggml_backend_buffer_t mi355_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    mi355_device_ctx * dev = (mi355_device_ctx *) buft->device->context;
    mi355q_set_device(dev->index);
    void * base = nullptr;
    if (mi355q_malloc(&base, size + 256) != MI355Q_OK) {          // OOM is recoverable: the caller checks for NULL
        GGML_LOG_ERROR("MI355: allocating %.2f MiB on device %d failed: %s\n", size / 1024.0 / 1024.0, dev->index, mi355q_last_error());
        return nullptr;
    }
    mi355_buffer_ctx * ctx = new mi355_buffer_ctx{ dev->index, base };
    return ggml_backend_buffer_init(buft, mi355_buffer_iface, ctx, size);
}

static size_t mi355_buft_get_alignment(ggml_backend_buffer_type_t buft) { GGML_UNUSED(buft); return 256; }   // >= 16 for the planar rows

static const struct ggml_backend_buffer_type_i mi355_buft_iface = {
    /* .get_name       = */ mi355_buft_get_name,
    /* .alloc_buffer   = */ mi355_buft_alloc_buffer,
    /* .get_alignment  = */ mi355_buft_get_alignment,
    /* .get_max_size   = */ nullptr,
    /* .get_alloc_size = */ nullptr,          // device rows have exactly the canonical size: ggml_nbytes is right
    /* .is_host        = */ nullptr,
};

static bool mi355_buft_is_ours(ggml_backend_buffer_type_t buft) { return buft && buft->iface.get_name == mi355_buft_get_name; }

// ------------------------------------------------------------------------------------------------ the ops
static void * mi355_workspace(mi355_backend_ctx * ctx, size_t bytes) {
    if (bytes <= ctx->workspace_size) return ctx->workspace;
    GGML_ASSERT(!ctx->capturing && "workspace must be sized before a launch graph is captured");
    MQ_CHECK(mi355q_stream_synchronize(ctx->stream));
    if (ctx->workspace) mi355q_free(ctx->workspace);
    // a captured launch graph holds the old workspace pointer in its kernel arguments: it must not be replayed
    if (ctx->graph) { mi355q_graph_destroy(ctx->graph); ctx->graph = nullptr; }
    ctx->graph_key = 0; ctx->key_repeats = 0;
    const size_t want = bytes + (bytes >> 2) + (1u << 20);
    MQ_CHECK(mi355q_malloc(&ctx->workspace, want));
    ctx->workspace_size = want;
    return ctx->workspace;
}

// GGML_OP_MUL_MAT (ggml.c:2730-2745; CPU semantics ggml-cpu.c:1266-1458): dst[ne01, ne11, ne12, ne13],
// src0 broadcast over dims 2/3 with r2 = ne12/ne02, r3 = ne13/ne03.
// x_alias: read the activations from this (flat-contiguous) buffer instead of src1->data -- an elided CONT (mi355_issue_nodes)
static void mi355_mul_mat(mi355_backend_ctx * ctx, struct ggml_tensor * dst, const void * x_alias = nullptr) {
    const struct ggml_tensor * src0 = dst->src[0];
    const struct ggml_tensor * src1 = dst->src[1];
    const int64_t K = src0->ne[0], M = src0->ne[1], N = src1->ne[1];
    const int64_t r2 = src1->ne[2] / src0->ne[2], r3 = src1->ne[3] / src0->ne[3];
    const size_t  ws = mi355q_mul_mat_workspace((int) src0->type, M, N, K);
    // (a Q8_0 matrix that is not in a weight buffer holds canonical rows: the library's kernels read device rows, so a slice is packed into scratch first --
    //  the reference harness' MUL_MAT cases and a quantized K cache read by a plain MUL_MAT come this way, model weights never do)
    // The whole ROOT tensor is packed (its rows are dense; a permuted view's are not) and the view's addresses are carried over: the device layout
    // permutes bytes inside a row only.
    const struct ggml_tensor * root = mi355_root(src0);
    const bool   pack = mi355_q80_canonical(src0) && mi355q_weights_are_planar((int) src0->type, K) == 1 && root->ne[0] == K && ggml_is_contiguous(root);
    const size_t wsa = (ws + 255) & ~(size_t) 255, pk = pack ? ggml_nbytes(root) : 0;
    void * wsp = (ws || pk) ? mi355_workspace(ctx, wsa + pk) : nullptr;
    const char * w_base = (const char *) src0->data;
    if (pack) {
        MQ_CHECK(mi355q_weights_pack_d2d((int) src0->type, (char *) wsp + wsa, root->data, ggml_nelements(root) / K, K, ctx->stream));
        w_base = (const char *) wsp + wsa + ((const char *) src0->data - (const char *) root->data);
    }
    for (int64_t i13 = 0; i13 < src1->ne[3]; ++i13) {
        for (int64_t i12 = 0; i12 < src1->ne[2]; ++i12) {
            const char * w = w_base + (i12 / r2) * src0->nb[2] + (i13 / r3) * src0->nb[3];
            const char * x = (const char *) (x_alias ? x_alias : src1->data) + i12 * src1->nb[2] + i13 * src1->nb[3];
            char *       y = (char *) dst->data + i12 * dst->nb[2] + i13 * dst->nb[3];
            MQ_CHECK(mi355q_mul_mat((int) src0->type, w, (int64_t) src0->nb[1], (const float *) x, (int64_t) src1->nb[1],
                                    (float *) y, (int64_t) dst->nb[1], M, N, K, wsp, ws, 0, ctx->stream));
        }
    }
}

// GGML_OP_MUL_MAT_ID (ggml.c:2771-2796; CPU ggml-cpu.c:1540-1718): as [K, M, n_expert], b [K, n_used|1, n_tok],
// ids i32 [n_used, n_tok] -> dst [M, n_used, n_tok].  The ids stay on the device (no host sync, cf. ggml-cuda.cu:2008-2011).
static void mi355_mul_mat_id(mi355_backend_ctx * ctx, struct ggml_tensor * dst) {
    const struct ggml_tensor * as  = dst->src[0];
    const struct ggml_tensor * b   = dst->src[1];
    const struct ggml_tensor * ids = dst->src[2];
    const int64_t K = as->ne[0], M = as->ne[1], n_expert = as->ne[2];
    const int64_t n_used = ids->ne[0], n_tok = ids->ne[1], b_ne1 = b->ne[1];
    const size_t  ws = mi355q_mul_mat_id_workspace((int) as->type, M, K, n_used, n_tok, b_ne1, n_expert);
    const bool   pack = mi355_q80_canonical(as) && mi355q_weights_are_planar((int) as->type, K) == 1 && as->nb[2] == as->nb[1] * (size_t) M;     // (see mi355_mul_mat)
    const size_t wsa = (ws + 255) & ~(size_t) 255, pk = pack ? (size_t) n_expert * (size_t) as->nb[2] : 0;
    void * wsp = (ws || pk) ? mi355_workspace(ctx, wsa + pk) : nullptr;
    const void * as_data = as->data;
    if (pack) { MQ_CHECK(mi355q_weights_pack_d2d((int) as->type, (char *) wsp + wsa, as->data, M * n_expert, K, ctx->stream)); as_data = (const char *) wsp + wsa; }
    MQ_CHECK(mi355q_mul_mat_id((int) as->type, as_data, (int64_t) as->nb[1], (int64_t) as->nb[2], n_expert,
                               (const float *) b->data, b_ne1, (int64_t) b->nb[1], (int64_t) b->nb[2],
                               (const int32_t *) ids->data, (int64_t) ids->nb[1],
                               (float *) dst->data, M, K, n_used, n_tok, wsp, ws, 0, ctx->stream));
}

// ------------------------------------------------------------------------------------------------ backend (stream)
static const char * mi355_backend_get_name(ggml_backend_t backend) { return ((mi355_backend_ctx *) backend->context)->name.c_str(); }

static void mi355_backend_free(ggml_backend_t backend) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *) backend->context;
    mi355q_set_device(ctx->device);
    if (ctx->stream) { mi355q_stream_synchronize(ctx->stream); mi355q_stream_destroy(ctx->stream); }
    if (ctx->copy_event) mi355q_event_destroy(ctx->copy_event);
    if (ctx->workspace) mi355q_free(ctx->workspace);
    if (getenv("MI355_GRAPH_STATS")) {
        fprintf(stderr, "MI355 graph_compute calls: %ld eager, %ld captured, %ld replayed\n", ctx->n_eager, ctx->n_captured, ctx->n_replayed);
        fprintf(stderr, "MI355 fusions (launches saved while issuing nodes): %ld norm*weight, %ld joined matmuls, %ld act*mul, %ld elided CONT, %ld add+norm\n",
                ctx->n_fused_norm, ctx->n_fused_mats, ctx->n_fused_act, ctx->n_elided_cont, ctx->n_fused_add);
    }
    if (getenv("MI355_GRAPH_STATS")) fprintf(stderr, "MI355 decode plans: %ld graph_compute calls ran as one persistent launch, %ld plans built\n", ctx->n_planned, ctx->n_plans_built);
    for (mi355_plan_entry * e : ctx->plans) { mi355q_plan_destroy(e->plan); delete e; }
    if (ctx->plan_abort) mi355q_host_free(ctx->plan_abort);
    if (ctx->dest_pinned) mi355q_host_free(ctx->dest_pinned);
    if (getenv("MI355_TIMING") && ctx->n_timed) {
        static const char * names[8] = { "cpy dests + upload", "graph key", "plan lookup / compile", "prefix nodes", "plan launch", "status copy", "suffix nodes", "other paths" };
        fprintf(stderr, "MI355 graph_compute host time over %ld planned calls (us per call):", ctx->n_timed);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.1f;", names[i], ctx->t_phase[i] / (double) ctx->n_timed);
        fprintf(stderr, "\n");
    }
    if (ctx->graph) mi355q_graph_destroy(ctx->graph);
    if (ctx->dest_table) mi355q_free(ctx->dest_table);
    delete ctx;
    delete backend;
}

// A decode plan whose polls timed out (another persistent kernel holding CUs, a reduced CU mask) raises its sticky abort word; the plugin reads it
// into pinned memory behind every launch.  Recoverable-error convention of the reference (ggml-backend-impl.h:110: graph_compute returns a
// ggml_status; only programming errors abort): the plans are destroyed, planning is switched off for this backend -- every later graph runs node by
// node -- and the NEXT graph_compute returns GGML_STATUS_FAILED once, because the step that timed out has produced garbage the caller already holds
// (llama_decode reports the failure; the application can decode that batch again).  The process is never aborted.
static bool mi355_plan_aborted(mi355_backend_ctx * ctx) {
    if (!ctx->plan_abort || !*ctx->plan_abort) return false;
    GGML_LOG_ERROR("MI355: a decode plan timed out waiting for its producers (another persistent kernel holding CUs?): decode plans are now off for %s, "
                   "the step that timed out is invalid; set MI355_NO_PLAN=1 to start without them\n", ctx->name.c_str());
    (void) mi355q_stream_synchronize(ctx->stream);
    for (mi355_plan_entry * e : ctx->plans) { mi355q_plan_destroy(e->plan); delete e; }
    ctx->plans.clear();
    ctx->plans_disabled = true;
    *ctx->plan_abort = 0;
    return true;
}

static void mi355_backend_synchronize(ggml_backend_t backend) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *) backend->context;
    mi355q_set_device(ctx->device);
    MQ_CHECK(mi355q_stream_synchronize(ctx->stream));
    // (a raised abort word is reported by the next graph_compute: synchronize has no status to return)
}

// ---- residency ops: thin wrappers over mi355q_op_* -- the tensor descriptor is ggml's ne[] / nb[] verbatim
static mi355q_tensor mi355_td(const struct ggml_tensor * t) {
    mi355q_tensor d;
    d.data = t->data; d.type = t->type == GGML_TYPE_F16 ? MI355Q_T_F16 : t->type == GGML_TYPE_Q8_0 ? MI355Q_TYPE_Q8_0 : t->type == GGML_TYPE_Q4_0 ? MI355Q_TYPE_Q4_0 : MI355Q_T_F32;     // (Q8_0: a quantized KV cache, CPY destination / FLASH_ATTN_EXT operand)
    for (int i = 0; i < 4; ++i) { d.ne[i] = t->ne[i]; d.nb[i] = (int64_t) t->nb[i]; }
    return d;
}
static bool mi355_f32_or_f16(enum ggml_type t) { return t == GGML_TYPE_F32 || t == GGML_TYPE_F16; }

static int mi355_unary_code(enum ggml_unary_op u) {
    switch (u) {
    case GGML_UNARY_OP_SILU: return MI355Q_UNARY_SILU;   case GGML_UNARY_OP_RELU: return MI355Q_UNARY_RELU;
    case GGML_UNARY_OP_SIGMOID: return MI355Q_UNARY_SIGMOID; case GGML_UNARY_OP_TANH: return MI355Q_UNARY_TANH;
    case GGML_UNARY_OP_NEG: return MI355Q_UNARY_NEG;     case GGML_UNARY_OP_ABS: return MI355Q_UNARY_ABS;
    default: return 0;
    }
}

static void mi355_glue_op(mi355_backend_ctx * ctx, struct ggml_tensor * dst, int cpy_index = -1) {
    const mi355q_tensor d = mi355_td(dst);
    const mi355q_tensor a = mi355_td(dst->src[0]);
    switch (dst->op) {
    case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: {
        const mi355q_tensor b = mi355_td(dst->src[1]);
        const int op = dst->op == GGML_OP_ADD ? MI355Q_OP_ADD : dst->op == GGML_OP_SUB ? MI355Q_OP_SUB : dst->op == GGML_OP_MUL ? MI355Q_OP_MUL : MI355Q_OP_DIV;
        MQ_CHECK(mi355q_op_bin_bcast(op, &a, &b, &d, ctx->stream));
    } break;
    case GGML_OP_UNARY:
        MQ_CHECK(mi355q_op_unary(mi355_unary_code(ggml_get_unary_op(dst)), &a, &d, ctx->stream));
        break;
    case GGML_OP_RMS_NORM: {
        float eps; memcpy(&eps, dst->op_params, sizeof(float));
        MQ_CHECK(mi355q_op_rms_norm(&a, &d, eps, ctx->stream));
    } break;
    case GGML_OP_CPY: {                                       // dst = src[1] viewed; data goes src[0] -> dst (ggml.c ggml_cpy_impl)
        if (cpy_index >= 0) MQ_CHECK(mi355q_op_cpy_indirect(&a, &d, ctx->dest_table, cpy_index, ctx->stream));
        else                MQ_CHECK(mi355q_op_cpy(&a, &d, ctx->stream));
    } break;
    case GGML_OP_CONT: case GGML_OP_DUP:
        MQ_CHECK(mi355q_op_cpy(&a, &d, ctx->stream));
        break;
    case GGML_OP_SOFT_MAX: {
        float scale, max_bias;
        memcpy(&scale, (const float *) dst->op_params + 0, sizeof(float));
        memcpy(&max_bias, (const float *) dst->op_params + 1, sizeof(float));
        if (dst->src[1]) { const mi355q_tensor m = mi355_td(dst->src[1]); MQ_CHECK(mi355q_op_soft_max(&a, &m, &d, scale, max_bias, ctx->stream)); }
        else MQ_CHECK(mi355q_op_soft_max(&a, nullptr, &d, scale, max_bias, ctx->stream));
    } break;
    case GGML_OP_GET_ROWS: {
        mi355q_tensor ids = mi355_td(dst->src[1]);
        MQ_CHECK(mi355q_op_get_rows(&a, &ids, &d, ctx->stream));
    } break;
    case GGML_OP_FLASH_ATTN_EXT: {                            // q = src[0], k, v, mask (optional); op_params: scale, max_bias, logit_softcap
        float scale, max_bias, softcap;
        memcpy(&scale, (const float *) dst->op_params + 0, sizeof(float));
        memcpy(&max_bias, (const float *) dst->op_params + 1, sizeof(float));
        memcpy(&softcap, (const float *) dst->op_params + 2, sizeof(float));
        const mi355q_tensor k = mi355_td(dst->src[1]), v = mi355_td(dst->src[2]);
        mi355q_tensor m; if (dst->src[3]) m = mi355_td(dst->src[3]);
        const size_t ws = mi355q_op_flash_attn_ext_workspace(v.ne[0], a.ne[1], a.ne[2], a.ne[3], k.ne[1]);
        void * wsp = mi355_workspace(ctx, ws);
        MQ_CHECK(mi355q_op_flash_attn_ext(&a, &k, &v, dst->src[3] ? &m : nullptr, &d, scale, max_bias, softcap, wsp, ws, ctx->stream));
    } break;
    case GGML_OP_SCALE: {
        float sc; memcpy(&sc, dst->op_params, sizeof(float));
        MQ_CHECK(mi355q_op_scale(&a, &d, sc, ctx->stream));
    } break;
    case GGML_OP_ARGSORT: {                                   // op_params[0]: GGML_SORT_ORDER_ASC (0) / _DESC (1); dst is i32 (4-byte elements, same strides)
        mi355q_tensor di = d; di.type = MI355Q_T_F32;
        MQ_CHECK(mi355q_op_argsort(&a, &di, ((const int32_t *) dst->op_params)[0] == GGML_SORT_ORDER_DESC, ctx->stream));
    } break;
    case GGML_OP_SUM_ROWS:
        MQ_CHECK(mi355q_op_sum_rows(&a, &d, ctx->stream));
        break;
    case GGML_OP_ROPE: {
        mi355q_rope_params p;
        const int32_t * q = (const int32_t *) dst->op_params;
        p.n_dims = q[1]; p.mode = q[2]; p.n_ctx_orig = q[4];
        memcpy(&p.freq_base, q + 5, 4); memcpy(&p.freq_scale, q + 6, 4); memcpy(&p.ext_factor, q + 7, 4);
        memcpy(&p.attn_factor, q + 8, 4); memcpy(&p.beta_fast, q + 9, 4); memcpy(&p.beta_slow, q + 10, 4);
        MQ_CHECK(mi355q_op_rope(&a, (const int32_t *) dst->src[1]->data, dst->src[2] ? (const float *) dst->src[2]->data : nullptr, &d, &p, ctx->stream));
    } break;
    case GGML_OP_MUL_MAT: {                                   // f16 / f32 src0 (the quantized types go through mi355_mul_mat)
        const mi355q_tensor b = mi355_td(dst->src[1]);
        MQ_CHECK(mi355q_op_mul_mat_f(&a, &b, &d, ctx->stream));
    } break;
    default: GGML_ABORT("mi355_glue_op: unexpected op");
    }
}

// everything a captured launch / a decode plan depends on, EXCEPT the destination pointer of CPY nodes (read from dest_table on the device)
static uint64_t mi355_graph_key(const struct ggml_cgraph * cgraph, uint64_t * second) {
    // four independent lanes per hash, fed round-robin and folded at the end: one chain of dependent multiplies (2 per word, ~14k words) was 16 us per token
    uint64_t hl[4] = { 1469598103934665603ull, 0x9AE16A3B2F90404Full, 0xC3A5C85C97CB3127ull, 0xB492B66FBE98F273ull }, h2l[4] = { 0x2545F4914F6CDD1Dull, 0x1B873593CC9E2D51ull, 0x85EBCA6BC2B2AE35ull, 0x27D4EB2F165667C5ull };
    unsigned lane = 0;
    // A token's graph is ~1000 nodes and this runs on every graph_compute: per node only what a launch can depend on is mixed in, a 64-bit word
    // at a time -- the node's op, parameters, type, shape, strides and address, and per operand its address, type and shape (an operand's strides
    // are those of its own node, or of a leaf whose address and shape fix them).
    auto mix64 = [&](uint64_t w) { uint64_t & h = hl[lane & 3], & h2 = h2l[lane & 3]; ++lane; h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h2 = (h2 + w) * 0xD6E8FEB86659FD93ull; h2 ^= h2 >> 32; };
    auto mix = [&](const void * p, size_t n) {
        const uint8_t * b = (const uint8_t *) p;
        for (; n >= 8; n -= 8, b += 8) { uint64_t w; memcpy(&w, b, 8); mix64(w); }
        if (n) { uint64_t w = 0; memcpy(&w, b, n); mix64(w | ((uint64_t) n << 56)); }
    };
    for (int i = 0; i < cgraph->n_nodes; ++i) {
        const struct ggml_tensor * n = cgraph->nodes[i];
        // views / reshapes launch nothing: what they describe reaches the key through the nodes that consume them (a view into the KV
        // cache at the store position moves every token and must not invalidate the capture)
        if (n->op == GGML_OP_NONE || n->op == GGML_OP_RESHAPE || n->op == GGML_OP_VIEW || n->op == GGML_OP_PERMUTE || n->op == GGML_OP_TRANSPOSE) continue;
        const bool cpy = n->op == GGML_OP_CPY;
        // ~20 words per node (this loop is most of the plugin's host time per token: 40 us for 1100 nodes when every field of every operand was mixed in):
        // the node's op, type, shape, strides above the first and address; its parameters (all of them only where an op has many); per operand that
        // exists its address, the two leading extents and the row stride -- an operand's type and its other extents follow from those of the node that
        // produced it (hashed itself) or are fixed for a leaf of this address (a weight, a graph input whose extents are the ones mixed in)
        mix64(((uint64_t) n->op << 32) | (uint64_t) n->type);
        const bool many = n->op == GGML_OP_ROPE || n->op == GGML_OP_FLASH_ATTN_EXT || n->op == GGML_OP_SOFT_MAX || n->op == GGML_OP_UNARY || n->op == GGML_OP_ARGSORT;
        mix(n->op_params, many ? sizeof(n->op_params) : 16);
        mix(n->ne, sizeof(n->ne)); mix(n->nb + 1, sizeof(n->nb) - sizeof(n->nb[0]));
        if (!cpy) mix64((uint64_t) (uintptr_t) n->data);
        uint64_t present = 0;
        for (int j = 0; j < GGML_MAX_SRC; ++j) {
            const struct ggml_tensor * t = n->src[j];
            if (!t) continue;
            present |= 1ull << j;
            mix64((uint64_t) t->ne[0] ^ ((uint64_t) t->ne[1] << 32) ^ ((uint64_t) t->type << 58));
            mix64((uint64_t) t->nb[1] ^ ((uint64_t) t->ne[2] << 40));
            if (!(cpy && j == 1)) mix64((uint64_t) (uintptr_t) t->data);
        }
        mix64(present);
    }
    uint64_t h = lane, h2 = ~(uint64_t) lane;                  // (the word count: lanes of graphs of different lengths do not line up)
    for (int k = 0; k < 4; ++k) { h = (h ^ hl[k]) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h2 = (h2 + h2l[k]) * 0xD6E8FEB86659FD93ull; h2 ^= h2 >> 32; }
    if (second) *second = h2;
    return h;
}

// ---- fusions around the path ---------------------------------------------------------------------------------
// A decode step is ~23 tiny dependent launches per llama layer and each costs ~6-8 us of dispatch latency whatever its size (a
// captured launch graph does not remove that: profiles/round1_plugin_layer.md), so the lever is FEWER launches.  While issuing the
// nodes the backend joins what the reference's graph builder emits as separate nodes:
//   RMS_NORM -> MUL(weight)                    one kernel                       (build_norm, src/llama-graph.cpp)
//   MUL_MATs on the same activations           one multi-matrix GEMV launch (N <= 8) / one shared activation image (N > 8)   (wq/wk/wv, ffn_gate/ffn_up)
//   UNARY(SiLU) -> MUL(up)                     one kernel                       (build_ffn LLM_FFN_SILU / LLM_FFN_PAR)
//   CONT of an already flat tensor -> MUL_MAT  the copy is skipped, the matmul reads the source
// Every fused form performs the same f32 operations in the same order as the separate nodes (results bit-identical, checked by
// tests/test_plugin.py with MI355_NO_FUSION=1 against the default).  A node issued EARLIER than its position in the graph (a joined
// matmul, a fused MUL) writes its output early: it is only done when no node it jumps over reads or writes memory overlapping that
// output (the graph allocator reuses freed tensors' memory) and all its inputs are already computed.
static bool mi355_is_view_op(const struct ggml_tensor * t) {
    return t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE;
}
static bool mi355_overlap(const struct ggml_tensor * a, const struct ggml_tensor * b) {
    if (!a || !b || !a->data || !b->data) return false;
    const char * a0 = (const char *) a->data, * b0 = (const char *) b->data;
    return a0 < b0 + ggml_nbytes(b) && b0 < a0 + ggml_nbytes(a);
}
static bool mi355_flat_contiguous(const struct ggml_tensor * t) {      // memory order == logical order (dims of size 1 may carry any stride)
    size_t expect = ggml_type_size(t->type);
    if (ggml_blck_size(t->type) != 1) return false;
    for (int d = 0; d < GGML_MAX_DIMS; ++d) { if (t->ne[d] != 1 && t->nb[d] != expect) return false; expect *= (size_t) t->ne[d]; }
    return true;
}

struct mi355_fuser {
    struct ggml_cgraph * g;
    std::unordered_map<const struct ggml_tensor *, int> uses, producer;
    std::vector<char> done;
    explicit mi355_fuser(struct ggml_cgraph * cgraph) : g(cgraph), done((size_t) cgraph->n_nodes, 0) {
        for (int i = 0; i < g->n_nodes; ++i) {
            producer[g->nodes[i]] = i;
            for (int j = 0; j < GGML_MAX_SRC; ++j) if (g->nodes[i]->src[j]) ++uses[g->nodes[i]->src[j]];
        }
    }
    bool single_use(const struct ggml_tensor * t) const {
        if (t->flags & GGML_TENSOR_FLAG_OUTPUT) return false;
        auto it = uses.find(t); return it != uses.end() && it->second == 1;
    }
    // is t (through its view chain) computed once everything before node i and the nodes marked done have been issued?
    bool ready_before(const struct ggml_tensor * t, int i) const {
        for (; t; t = t->view_src) {
            auto it = producer.find(t);
            if (it != producer.end() && it->second >= i && !done[(size_t) it->second] && !mi355_is_view_op(t)) return false;
        }
        return true;
    }
    // may node j be issued at position i < j?  (`self`: the node at i that the fused kernel computes on the fly)
    bool can_issue_early(int i, int j, const struct ggml_tensor * self = nullptr) const {
        const struct ggml_tensor * nj = g->nodes[j];
        for (int s = 0; s < GGML_MAX_SRC; ++s) if (nj->src[s] && nj->src[s] != self && !ready_before(nj->src[s], i)) return false;
        for (int k = i; k < j; ++k) {
            const struct ggml_tensor * nk = g->nodes[k];
            if (done[(size_t) k] || mi355_is_view_op(nk) || ggml_is_empty(nk) || nk == self) continue;
            if (mi355_overlap(nk, nj)) return false;
            for (int s = 0; s < GGML_MAX_SRC; ++s) if (mi355_overlap(nk->src[s], nj)) return false;
        }
        return true;
    }
    int next_compute(int i) const {                               // the next node after i that launches something (-1: none)
        for (int k = i + 1; k < g->n_nodes; ++k) if (!mi355_is_view_op(g->nodes[k]) && !ggml_is_empty(g->nodes[k])) return done[(size_t) k] ? -1 : k;
        return -1;
    }
};

static bool mi355_joinable_mat(const struct ggml_tensor * n) {
    if (n->op != GGML_OP_MUL_MAT || !mi355_is_quant(n->src[0]->type)) return false;
    const struct ggml_tensor * w = n->src[0], * x = n->src[1];
    return x->type == GGML_TYPE_F32 && x->ne[2] == 1 && x->ne[3] == 1 && w->ne[2] == 1 && w->ne[3] == 1 &&
           mi355q_weights_are_planar((int) w->type, w->ne[0]) == 1 && !mi355_q80_canonical(w);
}

static bool mi355_operand_ok(const struct ggml_tensor * t);
#include "decode-plan.inc"

// `only`: issue just these nodes (the prefix / suffix around a decode plan), one by one without fusions
static enum ggml_status mi355_issue_nodes(mi355_backend_ctx * ctx, struct ggml_cgraph * cgraph, const std::vector<int> * only = nullptr) {
    static const bool no_fusion = getenv("MI355_NO_FUSION") != nullptr;
    const bool fuse = !no_fusion && cgraph->n_nodes >= 4 && !only;
    std::vector<char> want;
    if (only) { want.assign((size_t) cgraph->n_nodes, 0); for (int i : *only) want[(size_t) i] = 1; }
    mi355_fuser * fz = fuse ? new mi355_fuser(cgraph) : nullptr;
    struct fz_guard { mi355_fuser * p; ~fz_guard() { delete p; } } guard{ fz };
    int cpy_index = 0;
    for (int i = 0; i < cgraph->n_nodes; ++i) {
        struct ggml_tensor * node = cgraph->nodes[i];
        if (ggml_is_empty(node)) continue;
        if (node->op == GGML_OP_CPY) ++cpy_index;                 // (CPY nodes are never fused: their table slot is their ordinal)
        if (only && !want[(size_t) i]) continue;
        if (fz && fz->done[(size_t) i]) continue;
        switch (node->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            break;
        case GGML_OP_MUL_MAT:
            if (!mi355_is_quant(node->src[0]->type)) { mi355_glue_op(ctx, node); break; }
            if (fz && mi355_joinable_mat(node)) {                 // other matmuls on the same activations (at most 4 per launch)
                int idx[4] = { i, -1, -1, -1 }, n = 1;
                for (int j = i + 1; j < cgraph->n_nodes && j <= i + 24 && n < 4; ++j) {
                    const struct ggml_tensor * o = cgraph->nodes[j];
                    if (fz->done[(size_t) j] || !mi355_joinable_mat(o) || o->src[1] != node->src[1] || o->src[0]->ne[0] != node->src[0]->ne[0] ||
                        mi355q_act_type((int) o->src[0]->type) != mi355q_act_type((int) node->src[0]->type)) continue;   // one activation format per launch
                    bool ok = fz->can_issue_early(i, j);
                    for (int q = 1; q < n && ok; ++q) ok = !mi355_overlap(cgraph->nodes[idx[q]], o);
                    if (ok) idx[n++] = j;
                }
                if (n > 1) {
                    mi355q_mat mats[4];
                    for (int q = 0; q < n; ++q) {
                        const struct ggml_tensor * o = cgraph->nodes[idx[q]];
                        mats[q].type = (int) o->src[0]->type; mats[q].w = o->src[0]->data; mats[q].w_stride = (int64_t) o->src[0]->nb[1];
                        mats[q].y = (float *) o->data; mats[q].y_stride = (int64_t) o->nb[1]; mats[q].m = o->src[0]->ne[1];
                    }
                    const struct ggml_tensor * x = node->src[1];
                    size_t ws = 0;                            // (prefill sizes: the matrix-core tiers share one prepared copy of the activations)
                    for (int q = 0; q < n; ++q) { const size_t b = mi355q_mul_mat_workspace(mats[q].type, mats[q].m, x->ne[1], x->ne[0]); if (b > ws) ws = b; }
                    void * wsp = ws ? mi355_workspace(ctx, ws) : nullptr;
                    MQ_CHECK(mi355q_mul_mat_multi(mats, n, (const float *) x->data, (int64_t) x->nb[1], x->ne[1], x->ne[0], wsp, ws, 0, ctx->stream));
                    for (int q = 1; q < n; ++q) fz->done[(size_t) idx[q]] = 1;
                    ctx->n_fused_mats += n - 1;
                    break;
                }
            }
            mi355_mul_mat(ctx, node);
            break;
        case GGML_OP_MUL_MAT_ID: mi355_mul_mat_id(ctx, node); break;
        case GGML_OP_CPY:
            // the indirect form only when the table was uploaded by THIS graph_compute call (a stale table would send the K / V rows
            // of this token to an earlier call's cache slots)
            mi355_glue_op(ctx, node, ctx->dest_valid && cpy_index - 1 < MI355_MAX_CPY_DESTS ? cpy_index - 1 : -1);
            break;
        case GGML_OP_RMS_NORM: {
            const int j = fz ? fz->next_compute(i) : -1;          // RMS_NORM -> MUL by a [ne0] weight vector
            if (j >= 0 && cgraph->nodes[j]->op == GGML_OP_MUL && fz->single_use(node) && node->type == GGML_TYPE_F32 && node->nb[0] == 4) {
                struct ggml_tensor * mul = cgraph->nodes[j];
                const struct ggml_tensor * w = mul->src[0] == node ? mul->src[1] : (mul->src[1] == node ? mul->src[0] : nullptr);
                if (w && w != node && w->type == GGML_TYPE_F32 && ggml_is_contiguous(w) && w->ne[0] == node->ne[0] && ggml_nelements(w) == w->ne[0] &&
                    ggml_are_same_shape(mul, node) && mul->nb[0] == 4 && node->src[0]->nb[0] == 4 && fz->ready_before(w, i) &&
                    (mul->data == node->src[0]->data || !mi355_overlap(node->src[0], mul)) && !mi355_overlap(w, mul)) {   // in place or disjoint: rows are written while others are read
                    float eps; memcpy(&eps, node->op_params, sizeof(float));
                    const mi355q_tensor a = mi355_td(node->src[0]), d = mi355_td(mul);
                    MQ_CHECK(mi355q_op_add_rms_norm_mul(&a, nullptr, nullptr, (const float *) w->data, &d, eps, ctx->stream));
                    fz->done[(size_t) j] = 1; ++ctx->n_fused_norm;
                    break;
                }
            }
            mi355_glue_op(ctx, node);
        } break;
        case GGML_OP_UNARY: {
            const int uop = mi355_unary_code(ggml_get_unary_op(node));
            if (fz && (uop == MI355Q_UNARY_SILU || uop == MI355Q_UNARY_RELU || uop == MI355Q_UNARY_SIGMOID) && fz->single_use(node) &&
                node->type == GGML_TYPE_F32 && ggml_is_contiguous(node) && ggml_is_contiguous(node->src[0])) {
                int j = -1;                                       // the MUL that consumes it, a few nodes on (ffn_up's matmul sits in between)
                for (int k = i + 1; k < cgraph->n_nodes && k <= i + 8; ++k)
                    if (cgraph->nodes[k]->op == GGML_OP_MUL && (cgraph->nodes[k]->src[0] == node || cgraph->nodes[k]->src[1] == node)) { j = k; break; }
                if (j >= 0 && !fz->done[(size_t) j]) {
                    struct ggml_tensor * mul = cgraph->nodes[j];
                    const struct ggml_tensor * o = mul->src[0] == node ? mul->src[1] : mul->src[0];
                    if (o != node && o->type == GGML_TYPE_F32 && ggml_is_contiguous(o) && ggml_are_same_shape(o, node) && ggml_are_same_shape(mul, node) &&
                        ggml_is_contiguous(mul) && fz->can_issue_early(i, j, node)) {
                        const mi355q_tensor a = mi355_td(node->src[0]), b = mi355_td(o), d = mi355_td(mul);
                        MQ_CHECK(mi355q_op_unary_mul(uop, &a, &b, &d, ctx->stream));
                        fz->done[(size_t) j] = 1; ++ctx->n_fused_act;
                        break;
                    }
                }
            }
            mi355_glue_op(ctx, node);
        } break;
        case GGML_OP_CONT: {
            const int j = fz ? fz->next_compute(i) : -1;          // CONT of a flat tensor feeding a quantized matmul: no copy
            if (j >= 0 && cgraph->nodes[j]->op == GGML_OP_MUL_MAT && cgraph->nodes[j]->src[1] == node && mi355_is_quant(cgraph->nodes[j]->src[0]->type) &&
                fz->single_use(node) && node->type == GGML_TYPE_F32 && node->src[0]->type == GGML_TYPE_F32 && mi355_flat_contiguous(node->src[0]) &&
                ggml_is_contiguous(node) && !mi355_overlap(cgraph->nodes[j], node->src[0]) && ((uintptr_t) node->src[0]->data & 15) == 0) {
                mi355_mul_mat(ctx, cgraph->nodes[j], node->src[0]->data);
                fz->done[(size_t) j] = 1; ++ctx->n_elided_cont;
                break;
            }
            mi355_glue_op(ctx, node);
        } break;
        case GGML_OP_ADD: {
            // residual ADD -> RMS_NORM -> MUL(weight): one kernel that also stores the sum (it is the next residual's operand)
            const int j = fz ? fz->next_compute(i) : -1;
            const int k = j >= 0 && cgraph->nodes[j]->op == GGML_OP_RMS_NORM && cgraph->nodes[j]->src[0] == node ? fz->next_compute(j) : -1;
            if (k >= 0 && cgraph->nodes[k]->op == GGML_OP_MUL && node->type == GGML_TYPE_F32 && ggml_is_contiguous(node) &&
                ggml_are_same_shape(node->src[0], node) && ggml_are_same_shape(node->src[1], node) && node->src[0]->type == GGML_TYPE_F32 &&
                node->src[1]->type == GGML_TYPE_F32 && ggml_is_contiguous(node->src[0]) && ggml_is_contiguous(node->src[1])) {
                struct ggml_tensor * norm = cgraph->nodes[j], * mul = cgraph->nodes[k];
                const struct ggml_tensor * w = mul->src[0] == norm ? mul->src[1] : (mul->src[1] == norm ? mul->src[0] : nullptr);
                // the MUL output is written at the ADD's position: it may coincide exactly with an operand of the ADD (the kernel allows
                // that) but must not partially overlap one, nor touch the stored sum
                auto exact_or_disjoint = [&](const struct ggml_tensor * t) { return t->data == mul->data || !mi355_overlap(t, mul); };
                if (w && w != norm && fz->single_use(norm) && w->type == GGML_TYPE_F32 && ggml_is_contiguous(w) && w->ne[0] == node->ne[0] &&
                    ggml_nelements(w) == w->ne[0] && ggml_are_same_shape(mul, node) && ggml_is_contiguous(mul) && fz->ready_before(w, i) &&
                    exact_or_disjoint(node->src[0]) && exact_or_disjoint(node->src[1]) && !mi355_overlap(node, mul)) {
                    float eps; memcpy(&eps, norm->op_params, sizeof(float));
                    const mi355q_tensor a = mi355_td(node->src[0]), b = mi355_td(node->src[1]), sum = mi355_td(node), d = mi355_td(mul);
                    MQ_CHECK(mi355q_op_add_rms_norm_mul(&a, &b, &sum, (const float *) w->data, &d, eps, ctx->stream));
                    fz->done[(size_t) j] = 1; fz->done[(size_t) k] = 1; ++ctx->n_fused_add;
                    break;
                }
            }
            mi355_glue_op(ctx, node);
        } break;
        case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV:
        case GGML_OP_DUP: case GGML_OP_SOFT_MAX: case GGML_OP_ROPE: case GGML_OP_GET_ROWS: case GGML_OP_SCALE: case GGML_OP_FLASH_ATTN_EXT:
        case GGML_OP_ARGSORT: case GGML_OP_SUM_ROWS:
            mi355_glue_op(ctx, node); break;
        default:
            GGML_LOG_ERROR("MI355: op %s reached graph_compute but supports_op never accepts it\n", ggml_op_name(node->op));
            return GGML_STATUS_FAILED;
        }
    }
    return GGML_STATUS_SUCCESS;
}

// The nodes of a decode step are many and tiny (34 per llama layer): issued one by one the step is bound by host launch
// time.  A graph seen twice in a row is captured into a launch graph and replayed while its key stays the same; the one thing
// that legitimately moves every token -- where the new K / V rows are stored -- is passed through a device-side pointer table.
static enum ggml_status mi355_backend_graph_compute(ggml_backend_t backend, struct ggml_cgraph * cgraph) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *) backend->context;
    mi355q_set_device(ctx->device);
    if (mi355_plan_aborted(ctx)) return GGML_STATUS_FAILED;
    static const bool env_off = getenv("MI355_NO_GRAPHS") != nullptr;
    static const bool env_no_plan = getenv("MI355_NO_PLAN") != nullptr;
    static const bool timing = getenv("MI355_TIMING") != nullptr;
    static const long inject = getenv("MI355_TEST_INJECT_PLAN_ABORT") ? atol(getenv("MI355_TEST_INJECT_PLAN_ABORT")) : -1;
    bool try_graphs = !env_off && !ctx->graphs_disabled && cgraph->n_nodes >= 8;
    const bool try_plan = !env_no_plan && !ctx->plans_disabled && cgraph->n_nodes >= 8;
    // (MUL_MAT_ID groups its rows by expert on the device at every size: such graphs are capturable too)
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = timing ? now_us() : 0.0, t_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    auto lap = [&](int i) { if (timing) { const double t = now_us(); t_acc[i] += t - t_mark; t_mark = t; } };

    // destination pointers of the CPY nodes of THIS call: staged in pinned memory, uploaded by one asynchronous copy
    int n_cpy = 0;
    if (try_graphs || try_plan) {
        if (!ctx->dest_pinned) MQ_CHECK(mi355q_host_malloc((void **) &ctx->dest_pinned, sizeof(void *) * MI355_MAX_CPY_DESTS));
        for (int i = 0; i < cgraph->n_nodes; ++i) if (cgraph->nodes[i]->op == GGML_OP_CPY && !ggml_is_empty(cgraph->nodes[i])) {
            if (n_cpy < MI355_MAX_CPY_DESTS) ctx->dest_pinned[n_cpy] = cgraph->nodes[i]->data;
            ++n_cpy;
        }
        if (n_cpy > MI355_MAX_CPY_DESTS) n_cpy = -1;
        else if (n_cpy > 0) {
            if (!ctx->dest_table) MQ_CHECK(mi355q_malloc((void **) &ctx->dest_table, sizeof(void *) * MI355_MAX_CPY_DESTS));
            MQ_CHECK(mi355q_memcpy_h2d(ctx->dest_table, ctx->dest_pinned, sizeof(void *) * n_cpy, ctx->stream));
        }
    }
    ctx->dest_valid = n_cpy > 0;
    lap(0);
    enum ggml_status st = GGML_STATUS_SUCCESS;
    bool done = false;
    uint64_t key = 0, key2 = 0;
    if ((try_graphs || try_plan) && n_cpy >= 0) key = mi355_graph_key(cgraph, &key2);
    lap(1);

    // ---- the whole decode step as ONE persistent launch (decode-plan.inc), cached per graph key
    if (try_plan && n_cpy >= 0) {
        mi355_plan_entry * e = nullptr;
        for (size_t i = 0; i < ctx->plans.size(); ++i) if (ctx->plans[i]->key == key && ctx->plans[i]->key2 == key2 && ctx->plans[i]->n_nodes == cgraph->n_nodes) { e = ctx->plans[i]; if (i) std::swap(ctx->plans[i], ctx->plans[0]); break; }
        bool declined = false;
        for (int i = 0; i < ctx->plan_declined_n; ++i) declined = declined || ctx->plan_declined[i] == key;
        if (!e && !declined) {
            mi355_plan_entry * ne = new mi355_plan_entry();
            if (ctx->plans.size() >= 4) {                          // evict the least recently used plan FIRST (its launches must have finished): its device
                MQ_CHECK(mi355q_stream_synchronize(ctx->stream));  // block is what the new plan is built in (csrc/plan.hip keeps destroyed plans' blocks)
                mi355q_plan_destroy(ctx->plans.back()->plan); delete ctx->plans.back(); ctx->plans.pop_back();
            }
            const auto t_c0 = std::chrono::steady_clock::now();
            const bool compiled = mi355_plan_compile(ctx, cgraph, *ne);
            if (timing) fprintf(stderr, "MI355 plan compile: %.0f us (%s)\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_c0).count(), compiled ? "built" : "declined");
            if (compiled) {
                ne->key = key; ne->key2 = key2; ne->n_nodes = cgraph->n_nodes;
                ctx->plans.insert(ctx->plans.begin(), ne);
                e = ne; ++ctx->n_plans_built;
            } else { delete ne; ctx->plan_declined[ctx->plan_declined_n < 8 ? ctx->plan_declined_n++ : (int) (key & 7)] = key; }
        }
        lap(2);
        if (e) {
            if (!ctx->plan_abort) { MQ_CHECK(mi355q_host_malloc((void **) &ctx->plan_abort, 64)); *ctx->plan_abort = 0; }
            if (!e->pre.empty()) st = mi355_issue_nodes(ctx, cgraph, &e->pre);
            lap(3);
            if (st == GGML_STATUS_SUCCESS) {
                MQ_CHECK(mi355q_plan_run(e->plan, ctx->stream));
                lap(4);
                MQ_CHECK(mi355q_plan_status_async(e->plan, ctx->plan_abort, ctx->stream));
                lap(5);
                if (!e->post.empty()) st = mi355_issue_nodes(ctx, cgraph, &e->post);
                lap(6);
            }
            done = true; ++ctx->n_planned;
            if (inject >= 0 && ctx->n_planned == inject) { MQ_CHECK(mi355q_stream_synchronize(ctx->stream)); *ctx->plan_abort = 1; }      // (test hook: as if this launch had timed out)
            if (timing) { for (int i = 0; i < 8; ++i) ctx->t_phase[i] += t_acc[i]; ++ctx->n_timed; }
        }
    }

    if (!done && try_graphs && n_cpy >= 0) {
        if (key == ctx->graph_key) { ++ctx->key_repeats; ctx->key_changes = 0; }
        else {
            if (ctx->graph) { mi355q_graph_destroy(ctx->graph); ctx->graph = nullptr; }
            ctx->graph_key = key; ctx->key_repeats = 0;
            if (++ctx->key_changes >= 8) ctx->graphs_disabled = true;       // the graph changes every call: capturing would only cost
        }
        if (ctx->graph) {                                      // replay
            MQ_CHECK(mi355q_graph_launch(ctx->graph, ctx->stream));
            done = true; ++ctx->n_replayed;
        } else if (ctx->key_repeats >= 1) {                    // second sighting: capture (the first, eager run sized the workspace)
            if (mi355q_graph_capture_begin(ctx->stream) == MI355Q_OK) {
                ctx->capturing = true;
                st = mi355_issue_nodes(ctx, cgraph);
                ctx->capturing = false;
                mi355q_graph * g = nullptr;
                const int rc = mi355q_graph_capture_end(ctx->stream, &g);
                if (st == GGML_STATUS_SUCCESS && rc == MI355Q_OK && g) {
                    ctx->graph = g;
                    MQ_CHECK(mi355q_graph_launch(ctx->graph, ctx->stream));
                    done = true; ++ctx->n_captured;
                } else {
                    if (g) mi355q_graph_destroy(g);
                    ctx->graphs_disabled = true;               // not capturable on this system: stay eager
                    st = GGML_STATUS_SUCCESS;
                }
            }
        }
    }
    if (!done) { st = mi355_issue_nodes(ctx, cgraph); ++ctx->n_eager; }
    ctx->dest_valid = false;
    // The work is left running on the backend's stream: ggml_backend_graph_compute() synchronizes after this call and the asynchronous form's
    // callers call ggml_backend_synchronize() before they touch results (ggml-backend.cpp ggml_backend_graph_compute / _async), as with ggml-cuda.
    return st;
}

// ---- asynchronous tensor access, copies between backends and events (the --split-mode layer hop) ----------------------------
// Reference counterparts: ggml-cuda.cu:2360-2460 (set / get / cpy_tensor_async), :2784-2810 (event_record / event_wait), ggml-backend.cpp:1355-1445
// (how ggml_backend_sched uses them between splits).  Plain (non-quantized or canonical-layout) tensors are copied on the backend's stream;
// planar quantized rows need the layout conversion and take the synchronous buffer path.
static bool mi355_backend_is_ours(ggml_backend_t b) { return b && b->iface.get_name == mi355_backend_get_name; }

static void mi355_backend_set_tensor_async(ggml_backend_t backend, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *) backend->context;
    struct ggml_backend_buffer * buf = tensor->view_src ? tensor->view_src->buffer : tensor->buffer;
    GGML_ASSERT(mi355_buffer_is_ours(buf) && "unsupported buffer type");
    mi355q_set_device(ctx->device);
    if (mi355_rows_planar(tensor)) {                              // (weights are uploaded once, at load time)
        MQ_CHECK(mi355q_stream_synchronize(ctx->stream));
        mi355_buffer_transfer((mi355_buffer_ctx *) buf->context, tensor, (void *) data, offset, size, true);
        return;
    }
    MQ_CHECK(mi355q_memcpy_h2d((char *) tensor->data + offset, data, size, ctx->stream));
}

static void mi355_backend_get_tensor_async(ggml_backend_t backend, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *) backend->context;
    struct ggml_backend_buffer * buf = tensor->view_src ? tensor->view_src->buffer : tensor->buffer;
    GGML_ASSERT(mi355_buffer_is_ours(buf) && "unsupported buffer type");
    mi355q_set_device(ctx->device);
    if (mi355_rows_planar(tensor)) {
        MQ_CHECK(mi355q_stream_synchronize(ctx->stream));
        mi355_buffer_transfer((mi355_buffer_ctx *) buf->context, tensor, data, offset, size, false);
        return;
    }
    MQ_CHECK(mi355q_memcpy_d2h(data, (const char *) tensor->data + offset, size, ctx->stream));
}

// dst (in a buffer of backend_dst's device) = src (in a buffer of backend_src's device), without a host round trip and without blocking the
// host: the copy is enqueued on the SOURCE stream (behind the kernels that produce src), an event marks its end and the DESTINATION stream waits
// for that event -- hipMemcpyPeerAsync over xGMI between two devices, a device-to-device copy when both backends drive the same device.
static bool mi355_backend_cpy_tensor_async(ggml_backend_t backend_src, ggml_backend_t backend_dst, const struct ggml_tensor * src, struct ggml_tensor * dst) {
    if (!mi355_backend_is_ours(backend_src) || !mi355_backend_is_ours(backend_dst)) return false;
    struct ggml_backend_buffer * bs = src->view_src ? src->view_src->buffer : src->buffer;
    struct ggml_backend_buffer * bd = dst->view_src ? dst->view_src->buffer : dst->buffer;
    if (!mi355_buffer_is_ours(bs) || !mi355_buffer_is_ours(bd)) return false;
    if (mi355_is_quant(src->type) && (mi355_root(src)->ne[0] != mi355_root(dst)->ne[0] || mi355_rows_planar(src) != mi355_rows_planar(dst))) return false;   // (device rows of another row length, or of a weight / non-weight Q8_0 pair, are laid out differently)
    mi355_backend_ctx * cs = (mi355_backend_ctx *) backend_src->context, * cd = (mi355_backend_ctx *) backend_dst->context;
    const int dev_s = ((mi355_buffer_ctx *) bs->context)->device, dev_d = ((mi355_buffer_ctx *) bd->context)->device;
    if (dev_s != cs->device || dev_d != cd->device) return false;
    const size_t bytes = ggml_nbytes(src);                        // (same layout asserted by the caller: ggml_backend_tensor_copy_async)
    mi355q_set_device(cs->device);
    if (backend_src == backend_dst) {
        MQ_CHECK(mi355q_memcpy_d2d(dst->data, src->data, bytes, cs->stream));
        return true;
    }
    MQ_CHECK(mi355q_memcpy_peer(dst->data, dev_d, src->data, dev_s, bytes, cs->stream));
    if (!cs->copy_event) MQ_CHECK(mi355q_event_create(&cs->copy_event));
    MQ_CHECK(mi355q_event_record(cs->copy_event, cs->stream));
    mi355q_set_device(cd->device);
    MQ_CHECK(mi355q_event_wait(cd->stream, cs->copy_event));
    return true;
}

static void mi355_backend_event_record(ggml_backend_t backend, ggml_backend_event_t event) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *) backend->context;
    mi355q_set_device(ctx->device);
    MQ_CHECK(mi355q_event_record(event->context, ctx->stream));
}

static void mi355_backend_event_wait(ggml_backend_t backend, ggml_backend_event_t event) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *) backend->context;
    mi355q_set_device(ctx->device);
    MQ_CHECK(mi355q_event_wait(ctx->stream, event->context));
}

static const struct ggml_backend_i mi355_backend_iface = {
    /* .get_name           = */ mi355_backend_get_name,
    /* .free               = */ mi355_backend_free,
    /* .set_tensor_async   = */ mi355_backend_set_tensor_async,
    /* .get_tensor_async   = */ mi355_backend_get_tensor_async,
    /* .cpy_tensor_async   = */ mi355_backend_cpy_tensor_async,
    /* .synchronize        = */ mi355_backend_synchronize,
    /* .graph_plan_create  = */ nullptr,
    /* .graph_plan_free    = */ nullptr,
    /* .graph_plan_update  = */ nullptr,
    /* .graph_plan_compute = */ nullptr,
    /* .graph_compute      = */ mi355_backend_graph_compute,
    /* .event_record       = */ mi355_backend_event_record,
    /* .event_wait         = */ mi355_backend_event_wait,
};

// ------------------------------------------------------------------------------------------------ device
static const char * mi355_dev_get_name(ggml_backend_dev_t dev) { return ((mi355_device_ctx *) dev->context)->name.c_str(); }
static const char * mi355_dev_get_description(ggml_backend_dev_t dev) { return ((mi355_device_ctx *) dev->context)->description.c_str(); }

static void mi355_dev_get_memory(ggml_backend_dev_t dev, size_t * free, size_t * total) {
    mi355_device_ctx * ctx = (mi355_device_ctx *) dev->context;
    size_t f = 0, t = 0;
    mi355q_device_info(ctx->index, nullptr, 0, &f, &t, nullptr);
    *free = f; *total = t;                       // real numbers: they drive the layer-split ratios (llama-model.cpp:1441-1449)
}

static enum ggml_backend_dev_type mi355_dev_get_type(ggml_backend_dev_t dev) { GGML_UNUSED(dev); return GGML_BACKEND_DEVICE_TYPE_GPU; }

static void mi355_dev_get_props(ggml_backend_dev_t dev, struct ggml_backend_dev_props * props) {
    props->name = mi355_dev_get_name(dev);
    props->description = mi355_dev_get_description(dev);
    props->type = GGML_BACKEND_DEVICE_TYPE_GPU;
    mi355_dev_get_memory(dev, &props->memory_free, &props->memory_total);
    props->caps = { /* async */ true, /* host_buffer */ true, /* buffer_from_host_ptr */ false, /* events */ true };
}

static ggml_backend_t mi355_dev_init_backend(ggml_backend_dev_t dev, const char * params) {
    GGML_UNUSED(params);
    mi355_device_ctx * dctx = (mi355_device_ctx *) dev->context;
    if (mi355q_set_device(dctx->index) != MI355Q_OK) {
        GGML_LOG_ERROR("MI355: cannot select device %d: %s\n", dctx->index, mi355q_last_error());
        return nullptr;
    }
    mi355_backend_ctx * ctx = new mi355_backend_ctx;
    ctx->device = dctx->index;
    ctx->name = dctx->name;
    if (mi355q_stream_create(&ctx->stream) != MI355Q_OK) {
        GGML_LOG_ERROR("MI355: stream creation failed: %s\n", mi355q_last_error());
        delete ctx;
        return nullptr;
    }
    return new ggml_backend{ /* .guid = */ mi355_guid(), /* .iface = */ mi355_backend_iface, /* .device = */ dev, /* .context = */ ctx };
}

static ggml_backend_buffer_type_t mi355_dev_get_buffer_type(ggml_backend_dev_t dev) { return &((mi355_device_ctx *) dev->context)->buft; }

// a tensor we can read as an operand: lives in one of OUR buffers (or is not placed yet / is the loader's
// zero-size probe buffer, llama-model.cpp:236-241 -- only buffer->buft may be looked at there)
static bool mi355_operand_ok(const struct ggml_tensor * t) {
    return t->buffer == nullptr || mi355_buft_is_ours(t->buffer->buft);
}

static bool mi355_dev_supports_op(ggml_backend_dev_t dev, const struct ggml_tensor * op) {
    GGML_UNUSED(dev);
    switch (op->op) {
    case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
        return true;
    case GGML_OP_MUL_MAT: {
        const struct ggml_tensor * a = op->src[0];
        const struct ggml_tensor * b = op->src[1];
        if (mi355_f32_or_f16(a->type)) {                          // attention KQ / KQV: k contiguous in both operands, any other strides
            if (b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || !mi355_operand_ok(a) || !mi355_operand_ok(b)) return false;
            return a->nb[0] == ggml_type_size(a->type) && b->nb[0] == sizeof(float) && (op->ne[0] * op->ne[1] * op->ne[2] * op->ne[3] + 3) / 4 <= 0x7FFFFFFF;
        }
        if (!mi355_is_quant(a->type) || b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32) return false;
        if (!mi355_operand_ok(a)) return false;
        // src0: whole rows of its root tensor, rows dense, 16-byte aligned row pitch where the rows are planar
        if (a->nb[0] != ggml_type_size(a->type) || mi355_root(a)->ne[0] != a->ne[0]) return false;
        if (a->nb[1] < ggml_row_size(a->type, a->ne[0])) return false;
        if (mi355q_weights_are_planar((int) a->type, a->ne[0]) && (a->nb[1] % 16 || a->nb[2] % 16 || a->nb[3] % 16)) return false;
        if (mi355_q80_canonical(a) && mi355q_weights_are_planar((int) a->type, a->ne[0]) && !ggml_is_contiguous(mi355_root(a))) return false;   // (packed as a whole before the product: mi355_mul_mat)
        // src1 / dst: rows contiguous (any row pitch); permuted activations stay on the CPU
        if (b->nb[0] != sizeof(float) || b->nb[1] < b->ne[0] * sizeof(float)) return false;
        if (op->nb[0] != sizeof(float) || !ggml_is_contiguous(op)) return false;
        if (b->ne[1] > 65535) return false;
        return true;
    }
    case GGML_OP_MUL_MAT_ID: {
        const struct ggml_tensor * a = op->src[0];
        const struct ggml_tensor * b = op->src[1];
        const struct ggml_tensor * ids = op->src[2];
        if (!mi355_is_quant(a->type) || b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || ids->type != GGML_TYPE_I32) return false;
        if (!mi355_operand_ok(a)) return false;
        if (!ggml_is_contiguous(a) || a->view_src) return false;
        if (b->nb[0] != sizeof(float) || !ggml_is_contiguous(op) || ids->nb[0] != sizeof(int32_t)) return false;
        if (ids->ne[0] * ids->ne[1] > 65535) return false;
        return true;
    }
    // ---- residency ops: f32 (CPY/CONT/DUP also f16), operands in our buffers
    case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: {
        const struct ggml_tensor * a = op->src[0];
        const struct ggml_tensor * b = op->src[1];
        if (a->type != GGML_TYPE_F32 || b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32) return false;
        return mi355_operand_ok(a) && mi355_operand_ok(b) && ggml_are_same_shape(a, op) && ggml_can_repeat(b, a);
    }
    case GGML_OP_UNARY: {
        const struct ggml_tensor * a = op->src[0];
        if (a->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || !mi355_operand_ok(a)) return false;
        return mi355_unary_code(ggml_get_unary_op(op)) != 0;
    }
    case GGML_OP_RMS_NORM: {
        const struct ggml_tensor * a = op->src[0];
        return a->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && mi355_operand_ok(a) && a->nb[0] == sizeof(float) && op->nb[0] == sizeof(float);
    }
    case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: {
        const struct ggml_tensor * a = op->src[0];
        if (op->op == GGML_OP_CPY && (op->type == GGML_TYPE_Q8_0 || op->type == GGML_TYPE_Q4_0))      // the K / V stores of a quantized cache (-ctk q8_0 / q4_0): whole blocks, contiguous destination
            return a->type == GGML_TYPE_F32 && mi355_operand_ok(a) && a->nb[0] == sizeof(float) && a->ne[0] % 32 == 0 && op->ne[0] % 32 == 0 &&
                   ggml_is_contiguous(op) && ggml_nelements(a) == ggml_nelements(op) && mi355_q80_canonical(op);
        return mi355_f32_or_f16(a->type) && mi355_f32_or_f16(op->type) && mi355_operand_ok(a) && ggml_nelements(a) == ggml_nelements(op);
    }
    case GGML_OP_GET_ROWS: {
        const struct ggml_tensor * a = op->src[0];
        const struct ggml_tensor * ids = op->src[1];
        return mi355_f32_or_f16(a->type) && op->type == GGML_TYPE_F32 && ids->type == GGML_TYPE_I32 && mi355_operand_ok(a) && mi355_operand_ok(ids);
    }
    case GGML_OP_SCALE: {
        const struct ggml_tensor * a = op->src[0];
        return a->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && mi355_operand_ok(a);
    }
    case GGML_OP_ARGSORT: {                                   // the MoE router's top-k (ggml_top_k = argsort desc + view)
        const struct ggml_tensor * a = op->src[0];
        return a->type == GGML_TYPE_F32 && op->type == GGML_TYPE_I32 && mi355_operand_ok(a) && op->nb[0] == sizeof(int32_t) && a->ne[0] <= 32768;
    }
    case GGML_OP_SUM_ROWS: {
        const struct ggml_tensor * a = op->src[0];
        return a->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && mi355_operand_ok(a);
    }
    case GGML_OP_ROPE: {
        const struct ggml_tensor * a = op->src[0];
        const int mode = ((const int32_t *) op->op_params)[2];
        if (a->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || !mi355_operand_ok(a) || a->nb[0] != sizeof(float) || op->nb[0] != sizeof(float)) return false;
        if (mode != 0 && mode != 2) return false;                 // normal and neox; mrope / vision stay on the CPU
        if (op->src[2] && op->src[2]->type != GGML_TYPE_F32) return false;
        return a->ne[0] % 2 == 0;
    }
    case GGML_OP_FLASH_ATTN_EXT: {                            // f16 or Q8_0 KV cache, f32 queries, f16 mask; other cache types stay on the CPU
        const struct ggml_tensor * q = op->src[0], * k = op->src[1], * v = op->src[2], * m = op->src[3];
        const bool kv_q80 = (k->type == GGML_TYPE_Q8_0 && v->type == GGML_TYPE_Q8_0) || (k->type == GGML_TYPE_Q4_0 && v->type == GGML_TYPE_Q4_0);       // a Q8_0 / Q4_0 cache: the one-workgroup-per-row kernel with the CPU's arithmetic
        const size_t kv_bb = k->type == GGML_TYPE_Q4_0 ? 18u : 34u;
        if (q->type != GGML_TYPE_F32 || !((k->type == GGML_TYPE_F16 && v->type == GGML_TYPE_F16) || kv_q80) || op->type != GGML_TYPE_F32) return false;
        if (!mi355_operand_ok(q) || !mi355_operand_ok(k) || !mi355_operand_ok(v) || (m && !mi355_operand_ok(m))) return false;
        if (q->nb[0] != 4 || k->nb[0] != (kv_q80 ? kv_bb : 2u) || v->nb[0] != (kv_q80 ? kv_bb : 2u) || !ggml_is_contiguous(op)) return false;
        if (kv_q80 && (k->ne[0] % 32 || v->ne[0] % 32 || k->ne[1] > 8192 || !mi355_q80_canonical(k) || !mi355_q80_canonical(v))) return false;
        if (m && (m->type != GGML_TYPE_F16 || m->nb[0] != 2 || m->ne[0] < k->ne[1] || m->ne[1] < q->ne[1] || m->ne[2] != 1 || m->ne[3] != 1)) return false;
        return k->ne[0] <= 256 && v->ne[0] <= 256 && k->ne[1] >= 1 && k->ne[1] <= 36864 && q->ne[2] <= 65535 && q->ne[3] <= 65535;
    }
    case GGML_OP_SOFT_MAX: {
        const struct ggml_tensor * a = op->src[0];
        const struct ggml_tensor * m = op->src[1];
        if (a->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || !mi355_operand_ok(a) || !ggml_is_contiguous(a) || !ggml_is_contiguous(op)) return false;
        if (m && (!mi355_f32_or_f16(m->type) || !mi355_operand_ok(m) || !ggml_is_contiguous(m) || m->ne[0] != a->ne[0] || m->ne[1] < a->ne[1])) return false;
        return true;
    }
    default:
        return false;
    }
}

static bool mi355_dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft) {
    return mi355_buft_is_ours(buft) && buft->device == dev;
}

// ---- pinned host buffer type (ggml-backend-impl.h:137-185 get_host_buffer_type; ggml-cuda.cu:1050-1110): staging memory the DMA engines read directly
static const char * mi355_host_buft_get_name(ggml_backend_buffer_type_t buft) { return ((mi355_device_ctx *) buft->device->context)->host_buft_name.c_str(); }
static void mi355_host_buffer_free(ggml_backend_buffer_t buffer) { mi355q_host_free(buffer->context); }
static ggml_backend_buffer_t mi355_host_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    mi355_device_ctx * dev = (mi355_device_ctx *) buft->device->context;
    mi355q_set_device(dev->index);
    void * ptr = nullptr;
    if (mi355q_host_malloc(&ptr, size + 64) != MI355Q_OK) return ggml_backend_buft_alloc_buffer(ggml_backend_cpu_buffer_type(), size);   // (pageable memory still works)
    ggml_backend_buffer_t buffer = ggml_backend_cpu_buffer_from_ptr(ptr, size);     // host memory behaves like a CPU buffer in every other respect
    buffer->buft = buft;
    buffer->iface.free_buffer = mi355_host_buffer_free;
    return buffer;
}
static size_t mi355_host_buft_get_alignment(ggml_backend_buffer_type_t buft) { GGML_UNUSED(buft); return 64; }
static bool mi355_host_buft_is_host(ggml_backend_buffer_type_t buft) { GGML_UNUSED(buft); return true; }
static const struct ggml_backend_buffer_type_i mi355_host_buft_iface = {
    /* .get_name       = */ mi355_host_buft_get_name,
    /* .alloc_buffer   = */ mi355_host_buft_alloc_buffer,
    /* .get_alignment  = */ mi355_host_buft_get_alignment,
    /* .get_max_size   = */ nullptr,
    /* .get_alloc_size = */ nullptr,
    /* .is_host        = */ mi355_host_buft_is_host,
};
static ggml_backend_buffer_type_t mi355_dev_get_host_buffer_type(ggml_backend_dev_t dev) { return &((mi355_device_ctx *) dev->context)->host_buft; }

// an op whose weights live in host memory is worth running here when the batch is large (the weights then cross PCIe once per batch):
// the reference's rule, ggml-cuda.cu:3285-3291
static bool mi355_dev_offload_op(ggml_backend_dev_t dev, const struct ggml_tensor * op) {
    GGML_UNUSED(dev);
    const int64_t batch = op->op == GGML_OP_MUL_MAT_ID ? op->ne[2] : (op->op == GGML_OP_GET_ROWS ? 0 : op->ne[1]);
    return batch >= 32;
}

static ggml_backend_event_t mi355_dev_event_new(ggml_backend_dev_t dev) {
    mi355_device_ctx * dctx = (mi355_device_ctx *) dev->context;
    mi355q_set_device(dctx->index);
    void * ev = nullptr;
    if (mi355q_event_create(&ev) != MI355Q_OK) return nullptr;
    return new ggml_backend_event{ /* .device = */ dev, /* .context = */ ev };
}
static void mi355_dev_event_free(ggml_backend_dev_t dev, ggml_backend_event_t event) {
    mi355q_set_device(((mi355_device_ctx *) dev->context)->index);
    mi355q_event_destroy(event->context);
    delete event;
}
static void mi355_dev_event_synchronize(ggml_backend_dev_t dev, ggml_backend_event_t event) {
    mi355q_set_device(((mi355_device_ctx *) dev->context)->index);
    MQ_CHECK(mi355q_event_synchronize(event->context));
}

static const struct ggml_backend_device_i mi355_device_iface = {
    /* .get_name             = */ mi355_dev_get_name,
    /* .get_description      = */ mi355_dev_get_description,
    /* .get_memory           = */ mi355_dev_get_memory,
    /* .get_type             = */ mi355_dev_get_type,
    /* .get_props            = */ mi355_dev_get_props,
    /* .init_backend         = */ mi355_dev_init_backend,
    /* .get_buffer_type      = */ mi355_dev_get_buffer_type,
    /* .get_host_buffer_type = */ mi355_dev_get_host_buffer_type,
    /* .buffer_from_host_ptr = */ nullptr,
    /* .supports_op          = */ mi355_dev_supports_op,
    /* .supports_buft        = */ mi355_dev_supports_buft,
    /* .offload_op           = */ mi355_dev_offload_op,
    /* .event_new            = */ mi355_dev_event_new,
    /* .event_free           = */ mi355_dev_event_free,
    /* .event_synchronize    = */ mi355_dev_event_synchronize,
};

// ------------------------------------------------------------------------------------------------ registry
struct mi355_reg_ctx {
    std::vector<ggml_backend_device *> devices;
};

static const char * mi355_reg_get_name(ggml_backend_reg_t reg) { GGML_UNUSED(reg); return "MI355"; }
static size_t mi355_reg_get_device_count(ggml_backend_reg_t reg) { return ((mi355_reg_ctx *) reg->context)->devices.size(); }
static ggml_backend_dev_t mi355_reg_get_device(ggml_backend_reg_t reg, size_t index) {
    mi355_reg_ctx * ctx = (mi355_reg_ctx *) reg->context;
    GGML_ASSERT(index < ctx->devices.size());
    return ctx->devices[index];
}
static void * mi355_reg_get_proc_address(ggml_backend_reg_t reg, const char * name) {
    GGML_UNUSED(reg); GGML_UNUSED(name);
    return nullptr;      // no split buffers / n_threads / extra buffer types: NULL is the documented "not provided"
}

static const struct ggml_backend_reg_i mi355_reg_iface = {
    /* .get_name         = */ mi355_reg_get_name,
    /* .get_device_count = */ mi355_reg_get_device_count,
    /* .get_device       = */ mi355_reg_get_device,
    /* .get_proc_address = */ mi355_reg_get_proc_address,
};

extern "C" {
GGML_BACKEND_API ggml_backend_reg_t ggml_backend_mi355_reg(void);
}

ggml_backend_reg_t ggml_backend_mi355_reg(void) {
    static std::mutex mutex;
    static bool initialized = false;
    static struct ggml_backend_reg reg;
    std::lock_guard<std::mutex> lock(mutex);
    if (!initialized) {
        mi355_reg_ctx * ctx = new mi355_reg_ctx;
        int n_phys = mi355q_device_count();
        if (n_phys > MI355_MAX_DEVICES) n_phys = MI355_MAX_DEVICES;
        // MI355_DUP_DEVICES=k (testing): every physical device appears under k names, so that the scheduler's multi-device path (layer split,
        // cpy_tensor_async, events) can be exercised on a one-GPU box; the duplicates drive the same HIP device through their own streams
        int dup = 1;
        if (const char * e = getenv("MI355_DUP_DEVICES")) { dup = atoi(e); if (dup < 1) dup = 1; if (dup * n_phys > MI355_MAX_DEVICES) dup = MI355_MAX_DEVICES / (n_phys > 0 ? n_phys : 1); }
        for (int i = 0; i < n_phys * dup; ++i) {
            mi355_device_ctx * dctx = new mi355_device_ctx;
            dctx->index = i / dup;
            dctx->name = "MI355_" + std::to_string(i);
            char desc[256] = "AMD Instinct MI355X";
            mi355q_device_info(dctx->index, desc, sizeof(desc), nullptr, nullptr, nullptr);
            dctx->description = desc;
            dctx->buft_name = dctx->name;
            dctx->host_buft_name = dctx->name + "_Host";
            ggml_backend_device * dev = new ggml_backend_device{ /* .iface = */ mi355_device_iface, /* .reg = */ &reg, /* .context = */ dctx };
            dctx->buft = { /* .iface = */ mi355_buft_iface, /* .device = */ dev, /* .context = */ nullptr };
            dctx->host_buft = { /* .iface = */ mi355_host_buft_iface, /* .device = */ dev, /* .context = */ nullptr };
            ctx->devices.push_back(dev);
        }
        reg = { /* .api_version = */ GGML_BACKEND_API_VERSION, /* .iface = */ mi355_reg_iface, /* .context = */ ctx };
        initialized = true;
    }
    return &reg;
}

static int ggml_backend_mi355_score(void) { return mi355q_device_count() > 0 ? 100 : 0; }

GGML_BACKEND_DL_IMPL(ggml_backend_mi355_reg)
GGML_BACKEND_DL_SCORE_IMPL(ggml_backend_mi355_score)

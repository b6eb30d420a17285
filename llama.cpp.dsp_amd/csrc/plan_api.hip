// plan_api.hip -- mi355q_plan_*: the decode plan's C-ABI (include/mi355q.h), dispatched to one of two engines that execute the same stage list
// with the same arithmetic (both share gemv_stream.cuh / act_quant.cuh; GEMV outputs are bit-identical to mi355q_mul_mat in either):
//
//   "regs" (csrc/plan.hip, default)   16 waves per workgroup stream their rows from HBM through a ring of registers (8 KiB per wave); the
//                                     weights of stage s + 1 are requested before the stage's operands are polled.  Round 2's kernel.
//   "ring" (csrc/plan_ring.hip)       round 3: a LOADER wave per workgroup copies the workgroup's rows of every stage, in stage order, into a ring
//                                     of LDS pages with LDS-DMA and never waits for an activation; 15 consumer waves work from LDS.  The data path
//                                     alone streams at 5.1 TB/s (tools/loaderonly.py; 6.6-6.9 in tools/micro/ring_stream.hip), but the whole step
//                                     is not yet faster than "regs" (DESIGN.md section 5.6b has the measurements), so it is opt-in:
//                                     MI355Q_PLAN_ENGINE=ring.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mi355q.h"

extern "C" {
#define ENGINE_DECLS(E) \
    int     mi355q_##E##_plan_create(mi355q_plan ** out, const mi355q_stage * stages, int n_stages, int flags); \
    int     mi355q_##E##_plan_run(mi355q_plan * plan, void * stream); \
    int     mi355q_##E##_plan_status(mi355q_plan * plan); \
    int     mi355q_##E##_plan_status_async(mi355q_plan * plan, unsigned * host_flag, void * stream); \
    int     mi355q_##E##_plan_debug_set_runs(mi355q_plan * plan, unsigned long long runs); \
    int     mi355q_##E##_plan_debug_words(mi355q_plan * plan, unsigned * out32); \
    int64_t mi355q_##E##_plan_weight_bytes(const mi355q_plan * plan); \
    int     mi355q_##E##_plan_launch_stages(const mi355q_plan * plan); \
    int     mi355q_##E##_plan_destroy(mi355q_plan * plan);
ENGINE_DECLS(regs)
ENGINE_DECLS(ring)
void mi355q_set_error(const char * msg);

namespace { struct Handle { int ring; mi355q_plan * impl; }; }
#define H(p) ((Handle *) (p))
#define CALL(p, f, ...) (H(p)->ring ? mi355q_ring_plan_##f(H(p)->impl, ##__VA_ARGS__) : mi355q_regs_plan_##f(H(p)->impl, ##__VA_ARGS__))

int mi355q_plan_create(mi355q_plan ** out, const mi355q_stage * stages, int n_stages, int flags) {
    if (!out) { mi355q_set_error("plan_create: null argument"); return MI355Q_ERR_SHAPE; }
    static const int want_ring = [] { const char * e = getenv("MI355Q_PLAN_ENGINE"); return e && strcmp(e, "ring") == 0 ? 1 : 0; }();
    mi355q_plan * impl = nullptr;
    const int rc = want_ring ? mi355q_ring_plan_create(&impl, stages, n_stages, flags) : mi355q_regs_plan_create(&impl, stages, n_stages, flags);
    if (rc != MI355Q_OK) return rc;
    Handle * h = new Handle{ want_ring, impl };
    *out = (mi355q_plan *) h;
    return MI355Q_OK;
}
int     mi355q_plan_run(mi355q_plan * p, void * stream) { if (!p) { mi355q_set_error("plan_run: null plan"); return MI355Q_ERR_SHAPE; } return CALL(p, run, stream); }
int     mi355q_plan_status(mi355q_plan * p) { return p ? CALL(p, status) : MI355Q_ERR_SHAPE; }
int     mi355q_plan_status_async(mi355q_plan * p, unsigned * host_flag, void * stream) { return p ? CALL(p, status_async, host_flag, stream) : MI355Q_ERR_SHAPE; }
int     mi355q_plan_debug_set_runs(mi355q_plan * p, unsigned long long runs) { return p ? CALL(p, debug_set_runs, runs) : MI355Q_ERR_SHAPE; }
int     mi355q_plan_debug_words(mi355q_plan * p, unsigned * out32) { return p ? CALL(p, debug_words, out32) : MI355Q_ERR_SHAPE; }
int64_t mi355q_plan_weight_bytes(const mi355q_plan * p) { return p ? CALL(p, weight_bytes) : 0; }
int     mi355q_plan_launch_stages(const mi355q_plan * p) { return p ? CALL(p, launch_stages) : 0; }
int     mi355q_plan_destroy(mi355q_plan * p) {
    if (!p) return MI355Q_OK;
    const int rc = CALL(p, destroy);
    delete H(p);
    return rc;
}

} // extern "C"

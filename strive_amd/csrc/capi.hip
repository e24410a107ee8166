// Error reporting, options and version of the C ABI.
#include "common.h"
#include <stdarg.h>
#include <stddef.h>

static thread_local char g_err[512] = "";

void strive_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* strive_last_error(void) { return g_err; }
extern "C" int strive_abi_version(void) { return STRIVE_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------
// Options (round 6): what used to be 19 getenv() switches inside the entry points.
// ---------------------------------------------------------------------------------------------
namespace {
struct OptionDesc { const char* name; size_t off; int def; int lo, hi; };
#define STRIVE_OPT(field, def, lo, hi) {#field, offsetof(StriveTuning, field), def, lo, hi}
const OptionDesc g_options[] = {
    STRIVE_OPT(cnn_small_batch, 96, 0, 1 << 20),   // map CNN: up to this many samples a launch runs the small-batch chain (0: never)
    STRIVE_OPT(cnn_chunk, 512, 8, 1024),           // samples pushed through the layer stack together
    STRIVE_OPT(cnn_tail_s, 0, 0, 4),               // samples per workgroup of the fused tail (0: by size; 1, 2, 4)
    STRIVE_OPT(conv_ws, 1, 0, 1),                  // conv2 on specialised producer / consumer waves (0: conv_bf6_kernel)
    STRIVE_OPT(conv_wsx, 1, 0, 1),                 // conv3 / conv4 on specialised waves with streamed weights (0: conv_bf6_kernel)
    STRIVE_OPT(conv_ws_dbg, 0, 0, 255),            // measurement only (results invalid): phase switches of the specialised-wave kernels
    STRIVE_OPT(scene_kernels, 1, 0, 1),            // scene-resident decoder kernels where they apply (0: launch-per-phase kernels)
    STRIVE_OPT(scene_tiles, 1, 0, 1),              // scenes of > 16 agents: forward node phases on the scene kernel in 16-row tiles (0: per-phase kernels)
    STRIVE_OPT(scene_prof, 0, 0, 1),               // measurement only: phase clocks of workgroup 0 into the workspace
    STRIVE_OPT(scene_split, 12, 0, 64),            // scenes of >= this many agents share their edge chunks among K workgroups (0: never)
    STRIVE_OPT(scene_fwd_k, -1, -1, 4),            // workgroups per scene for those edge chunks (-1: one per chunk, <= 4)
    STRIVE_OPT(sweep_step, -1, -1, 4),             // reverse sweep one launch per step on K workgroups per scene (-1: from 3 chunks on; 0: never)
    STRIVE_OPT(train_overlap, 1, 0, 1),            // training: CNN backward of finished steps on the library's side stream
    STRIVE_OPT(train_overlap_rows, 256, 1, 1 << 20),
    STRIVE_OPT(wgrad_atomics, 0, 0, 1),            // A/B: weight gradients of the small dense layers by atomics (round-3 form)
    STRIVE_OPT(dgrad_igemm, 0, 0, 1),              // A/B: fp32 implicit-GEMM data gradient (round-2 form)
    STRIVE_OPT(wgrad_igemm, 0, 0, 1),              // A/B: fp32 implicit-GEMM weight gradient
    STRIVE_OPT(wgrad_tile, 0, 0, 1),               // A/B: fp32 LDS-tile weight gradient
    STRIVE_OPT(wgrad_dbg, 0, 0, 255),              // measurement only
    STRIVE_OPT(planner_prof, 0, 0, 1),             // measurement only: planner phase clocks into the workspace tail
    STRIVE_OPT(planner_dbg, 0, 0, 65535),            // measurement only (results invalid): phase switches of the planner kernels
};
#undef STRIVE_OPT
constexpr int N_OPTIONS = (int)(sizeof(g_options) / sizeof(g_options[0]));
StriveTuning make_defaults() {
    StriveTuning t;
    for (int i = 0; i < N_OPTIONS; ++i) *reinterpret_cast<int*>(reinterpret_cast<char*>(&t) + g_options[i].off) = g_options[i].def;
    return t;
}
StriveTuning g_tuning = make_defaults();
const OptionDesc* find_option(const char* name) {
    if (!name) return nullptr;
    for (int i = 0; i < N_OPTIONS; ++i)
        if (strcmp(name, g_options[i].name) == 0) return &g_options[i];
    return nullptr;
}
}  // namespace

StriveTuning& strive_tuning() { return g_tuning; }

extern "C" int strive_set_option(const char* name, int64_t value) {
    const OptionDesc* o = find_option(name);
    if (!o) { strive_set_error("strive_set_option: unknown option '%s'", name ? name : "(null)"); return -1; }
    if (value < o->lo || value > o->hi) { strive_set_error("strive_set_option: %s = %lld outside [%d, %d]", name, (long long)value, o->lo, o->hi); return -1; }
    *reinterpret_cast<int*>(reinterpret_cast<char*>(&g_tuning) + o->off) = (int)value;
    return 0;
}
extern "C" int strive_get_option(const char* name, int64_t* value) {
    const OptionDesc* o = find_option(name);
    if (!o || !value) { strive_set_error("strive_get_option: unknown option '%s'", name ? name : "(null)"); return -1; }
    *value = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(&g_tuning) + o->off);
    return 0;
}
extern "C" int strive_reset_options(void) { g_tuning = make_defaults(); return 0; }
extern "C" int32_t strive_option_count(void) { return N_OPTIONS; }
extern "C" const char* strive_option_name(int32_t i) { return (i >= 0 && i < N_OPTIONS) ? g_options[i].name : nullptr; }

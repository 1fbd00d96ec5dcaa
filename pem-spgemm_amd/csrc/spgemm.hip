// spgemm.hip -- rows a9-a14: the three-step tiled SpGEMM hot path and the C export.
//
// Reference: step 1 = SPA bitmask kernels run twice (count + emit) or the NSPARSE binned
// hash path (spgemm.cu:1141-1218); step 2a/2b = warp-per-C-tile list intersection by binary
// search, again run twice (spgemm.cu:387-497); step 2c/2d masks + intra-tile CSR
// (spgemm.cu:499-591); step 3 numeric with a global read-modify-write per product
// (spgemm.cu:593-661).
//
// Here: every (A tile (i,k), B tile (k,j)) with matching k is a *product* of tile row i keyed by j, so grouping
// a row's products by j yields the C tile list (step 1) and the pair lists in ascending k (step 2a/2b) in one
// pass -- no SPA, no hash tables, no binary searches, nothing computed twice.  Products whose tiles cannot
// meet (A tile's occupied columns miss B tile's occupied rows) are dropped up front (exact; PEM_PRUNE=0 keeps
// the reference's lists).  Default path (round 4): one kernel forms and tests every product once and leaves the live ones in a
// list; per-row sorts of that list (registers / LDS bitonic / LDS radix), binned by row size, emit the pair lists; the global
// expand + radix-sort path ("esc") is kept as PEM_STEP1=esc.  Step 2c is the boolean row product (C row r = OR of B rows kk
// over kk in A row r), one PAIR per lane with a segmented OR toward the C tile's first pair; step 3 keeps one C entry per
// lane in a register, accumulates in ascending k with one fma per product, and stores once.  All outputs keep
// the reference layouts (include/pem_spgemm.h).  Kernel variants (A/B baselines kept for tests) are plan options
// (pem_cplan_set_option); the environment only sets a new plan's defaults:
//   PEM_STEP1=esc  PEM_WIDE=0  PEM_PRUNE=0  PEM_NO_WARM=1  PEM_EXPORT=rows
// Units: step1.hip, step2.hip, step3.hip (kernels + host driver of each step), export.hip (a14), this file (plan, options,
// graph replay, pem_spgemm, pem_cplan_get_array).
#include "spgemm_internal.h"

using namespace pem;

__global__ void warm_verify_kernel(const long long *__restrict__ d_scalars, const int *__restrict__ bin_count, WarmCheck wc, int *__restrict__ flags)
{
    warm_check(wc, d_scalars, bin_count, flags);
}

// ------------------------------------------------------------------------------------------
// host drivers
// ------------------------------------------------------------------------------------------
extern "C" pem_status pem_cplan_create(pem_ctx *ctx, const pem_tiled *A, const pem_tiled *B, int32_t tr_lo, int32_t tr_hi, pem_cplan **out)
{
    if (!ctx || !A || !B || !out) return PEM_E_INVALID;
    *out = nullptr;
    if (A->cols != B->rows) {
        set_error("pem_cplan_create: A is %d x %d but B is %d x %d", A->rows, A->cols, B->rows, B->cols);
        return PEM_E_INVALID;
    }
    if (A->value_bytes != B->value_bytes) {
        set_error("pem_cplan_create: A holds %s values, B %s; both operands must share one value type", A->value_bytes == 4 ? "fp32" : "fp64",
                  B->value_bytes == 4 ? "fp32" : "fp64");
        return PEM_E_INVALID;
    }
    if (tr_hi < 0) tr_hi = A->tile_rows;
    if (tr_lo < 0 || tr_lo > tr_hi || tr_hi > A->tile_rows) {
        set_error("pem_cplan_create: tile-row range [%d, %d) outside [0, %d]", tr_lo, tr_hi, A->tile_rows);
        return PEM_E_INVALID;
    }
    pem_cplan *p = new pem_cplan();
    p->owner = ctx;
    p->A = A;
    p->B = B;
    p->tr_lo = tr_lo;
    p->tr_hi = tr_hi;
    p->a_lo = A->h_tile_rowptr[(size_t)tr_lo];
    p->a_hi = A->h_tile_rowptr[(size_t)tr_hi];
    for (int i = tr_lo; i < tr_hi; ++i)      // the longest tile row of the slice (grid of the oversized rows' expansion)
        p->max_row_tiles = std::max(p->max_row_tiles, A->h_tile_rowptr[(size_t)i + 1] - A->h_tile_rowptr[(size_t)i]);
    // Every switch is a property of the PLAN, latched here (pem_cplan_set_option changes it later): a C ABI whose behaviour
    // followed the process environment at call time is not a boundary a maintainer can bind.  The environment variables only
    // give the DEFAULTS a new plan starts from (test hooks: PEM_S1_FORCE_KEY64 / PEM_S1_XLCAP push small inputs through the
    // code a B with more than 2^17 tile columns / a tile row beyond the LDS bins selects).
    auto env_is = [](const char *name, const char *val) {
        const char *e = getenv(name);
        return e && !strcmp(e, val);
    };
    p->opt_prune = !env_is("PEM_PRUNE", "0");
    p->opt_key64 = env_is("PEM_S1_FORCE_KEY64", "1");
    {
        const char *e = getenv("PEM_S1_XLCAP");
        p->opt_xlcap = e ? atoi(e) : 0;
    }
    p->opt_band = !env_is("PEM_S3_BAND", "0");
    p->opt_step1_esc = env_is("PEM_STEP1", "esc");
    p->opt_wide = !env_is("PEM_WIDE", "0");
    p->opt_warm = !env_is("PEM_NO_WARM", "1");
    p->opt_export_rows = env_is("PEM_EXPORT", "rows");
    p->opt_s1_serial = getenv("PEM_S1_SERIAL") != nullptr;
    p->opt_decode = !env_is("PEM_S3_DECODE", "0");
    p->opt_xl_global = env_is("PEM_S1_XL_GLOBAL", "1");
    p->opt_s3_xcd = !env_is("PEM_S3_XCD", "0");
    p->opt_idx64 = env_is("PEM_S3_IDX64", "1");
    p->opt_mark = !env_is("PEM_S3_MARK", "0");
    p->opt_s1_segments = env_is("PEM_S1_SEGMENTS", "1");
    {
        const char *e = getenv("PEM_S3_EPW");
        p->opt_epw = e ? atoi(e) : 0;
    }
    *out = p;
    return PEM_OK;
}

// A plan's instantiated graph is not destroyed the moment the plan lets go of it.  Round 3 saw hipGraphLaunch fault (a segmentation
// fault inside the runtime, hip::Graph::UpdateStreams by the rocgdb backtrace of the time, which was not kept) after 8-12 create /
// capture / replay / destroy rounds of plans with forked streams (tools/split_tune_emulate.py).  Round 4 ran that script once under
// rocgdb with executables destroyed at once (PEM_DEBUG_GRAPH_DESTROY=1) -- on this round's library and on the round-3 tree: 32
// plans each, no fault (gpurun_out of the round).  The builder-side candidates were read through and do not hold:
// every auxiliary stream is joined into the main one inside the capture, so synchronising the main stream (pem_spgemm does, before it
// returns, and pem_cplan_destroy again) covers all captured work; the fork / join events are only edges of the captured graph --
// an instantiated executable does not reference them; captures are thread-local and a context is single-caller.  What round 3 had
// and round 4 has not is FIVE streams in one capture (three bins + the oversized-row chain + main) against the runtime's four
// hardware queues -- round 4 plans step 1 over four.  With the cause not established, executables are still retired rather than
// destroyed in place, but the list is bounded: at S_RETIRE_MAX the device is drained and the list emptied.
constexpr size_t S_RETIRE_MAX = 32;
void pem::retire_graph(pem_ctx *ctx, pem_cplan *plan)
{
    if (!plan->graph_exec) return;
    if (ctx && !ctx->dbg_destroy_graphs) {
        ctx->retired_graphs.push_back(plan->graph_exec);
        if (ctx->retired_graphs.size() >= S_RETIRE_MAX && !ctx->capturing) {
            (void)hipDeviceSynchronize();
            for (auto ge : ctx->retired_graphs) (void)hipGraphExecDestroy(ge);
            ctx->retired_graphs.clear();
        }
    } else {
        (void)hipGraphExecDestroy(plan->graph_exec);
    }
    plan->graph_exec = nullptr;
}

extern "C" pem_status pem_cplan_destroy(pem_ctx *ctx, pem_cplan *plan)
{
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    } else {
        (void)hipDeviceSynchronize();   // (no context given: the plan's blocks still go back to its arena)
    }
    if (plan) retire_graph(ctx, plan);
    delete plan;
    return PEM_OK;
}

extern "C" pem_status pem_cplan_get_info(const pem_cplan *p, pem_cplan_info *info)
{
    if (!p || !info) return PEM_E_INVALID;
    info->tile_row_begin = p->tr_lo;
    info->tile_row_end = p->tr_hi;
    info->row_begin = p->tr_lo * 16;
    info->row_end = p->tr_hi * 16 < p->A->rows ? p->tr_hi * 16 : p->A->rows;
    info->ntiles_c = p->ntiles_c;
    info->npairs = p->npairs;
    info->nnz_c = p->nnz_c;
    info->npairs_all = p->npairs_all;
    return PEM_OK;
}

static int *plan_option_slot(pem_cplan *p, pem_option which)
{
    switch ((int)which) {                              // (pem_test.h adds values to the enum's)
    case PEM_OPT_PRUNE: return &p->opt_prune;
    case PEM_OPT_STEP1_GLOBAL_SORT: return &p->opt_step1_esc;
    case PEM_OPT_WIDE: return &p->opt_wide;
    case PEM_OPT_WARM: return &p->opt_warm;
    case PEM_OPT_S3_BAND: return &p->opt_band;
    case PEM_OPT_S1_FORCE_KEY64: return &p->opt_key64;
    case PEM_OPT_S1_XLCAP: return &p->opt_xlcap;
    case PEM_OPT_EXPORT_ROWS: return &p->opt_export_rows;
    case PEM_OPT_S1_SERIAL: return &p->opt_s1_serial;
    case PEM_OPT_S3_DECODE: return &p->opt_decode;
    case PEM_OPT_S1_XL_GLOBAL: return &p->opt_xl_global;
    case PEM_OPT_S3_EPW: return &p->opt_epw;
    case PEM_OPT_S3_XCD: return &p->opt_s3_xcd;
    case PEM_OPT_S3_IDX64: return &p->opt_idx64;
    case PEM_OPT_S3_MARK: return &p->opt_mark;
    case PEM_OPT_S1_SEGMENTS: return &p->opt_s1_segments;
    case PEM_OPT_S2_TRANSPOSED: return &p->opt_s2_transposed;
    default: return nullptr;
    }
}

extern "C" pem_status pem_cplan_set_option(pem_cplan *plan, pem_option which, int64_t value)
{
    if (!plan) return PEM_E_INVALID;
    int *slot = plan_option_slot(plan, which);
    if (!slot || value < 0 || value > 0x7FFFFFFF) {
        set_error("pem_cplan_set_option: unknown option %d or value %lld out of range", (int)which, (long long)value);
        return PEM_E_INVALID;
    }
    const int v = (which == PEM_OPT_S1_XLCAP || which == PEM_OPT_S3_EPW || which == PEM_OPT_S2_TRANSPOSED) ? (int)value : (value != 0);
    if (*slot == v) return PEM_OK;
    *slot = v;
    // a repeat pass re-uses the sizes (and possibly the captured graph) of the previous one: whatever changes the kernels
    // that run or the sizes they produce starts the plan over
    plan->warm = false;
    // ... and the step-wise API too: results of a step that ran under the old value do not feed a step that runs under the new one
    // (step 3 derives its tile lookup from PRUNE, step 2's output depends on DECODE, ...)
    // -- back to the last step the option does not touch
    int keep = 3;
    switch ((int)which) {
    case PEM_OPT_PRUNE: case PEM_OPT_STEP1_GLOBAL_SORT: case PEM_OPT_S1_FORCE_KEY64: case PEM_OPT_S1_XLCAP: case PEM_OPT_S1_XL_GLOBAL:
    case PEM_OPT_S1_SEGMENTS: keep = 0; break;
    case PEM_OPT_WIDE: case PEM_OPT_S3_DECODE: case PEM_OPT_S2_TRANSPOSED: keep = 1; break;
    case PEM_OPT_S3_BAND: case PEM_OPT_S3_EPW: case PEM_OPT_S3_IDX64: case PEM_OPT_S3_MARK: case PEM_OPT_S3_XCD: keep = 2; break;
    default: break;                                    // warm passes, the export variant, serial bins: no step's result changes
    }
    if (plan->state > keep) plan->state = keep;
    if (keep == 0) {
        plan->pairs_ready = false;
        plan->compact_valid = plan->c_rowidx_valid = false;
    }
    if (keep <= 1) plan->c_rowptr_valid = plan->c_rowcolidx_valid = false;
    if (plan->graph_exec) {
        retire_graph(plan->owner, plan);
    }
    plan->graph_failed = false;
    return PEM_OK;
}

extern "C" pem_status pem_cplan_get_option(const pem_cplan *plan, pem_option which, int64_t *value)
{
    if (!plan || !value) return PEM_E_INVALID;
    const int *slot = plan_option_slot(const_cast<pem_cplan *>(plan), which);
    if (!slot) {
        set_error("pem_cplan_get_option: unknown option %d", (int)which);
        return PEM_E_INVALID;
    }
    *value = *slot;
    return PEM_OK;
}

// a device-side primitive gave up (the chained scan's bounded wait): fail the call rather than hand back wrong arrays
pem_status pem::check_internal(const int *hf)
{
    if (!hf[FLAG_INTERNAL]) return PEM_OK;
    set_error("internal: a device scan ran out of its poll budget; the pass's results are not valid");
    return PEM_E_HIP;
}

pem_status pem::step_elapsed(pem_ctx *ctx, int e0, int e1, double *dst)
{
    float ms = 0.f;
    PEM_HIP(hipEventElapsedTime(&ms, ctx->ev[e0], ctx->ev[e1]));
    *dst = ms;
    return PEM_OK;
}

extern "C" pem_status pem_spgemm_step1(pem_ctx *ctx, pem_cplan *plan)
{
    if (!ctx || !plan) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    PEM_TRY(step1_impl(ctx, plan, false));   // step-wise calls always read the sizes back
    PEM_TRY(ensure_compact(ctx, plan));      // ... and leave step 1's outputs in the reference layout
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    PEM_TRY(launch_status(ctx));
    return step_elapsed(ctx, 0, 1, &ctx->timings.step1_ms);
}

extern "C" pem_status pem_spgemm_step2(pem_ctx *ctx, pem_cplan *plan)
{
    if (!ctx || !plan) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    plan->warm_pass = false;                 // step-wise calls read every size back, also after a warm pem_spgemm on this plan
    PEM_TRY(step2_impl(ctx, plan));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    PEM_TRY(launch_status(ctx));
    return step_elapsed(ctx, 2, 3, &ctx->timings.step2_ms);
}

extern "C" pem_status pem_spgemm_step3(pem_ctx *ctx, pem_cplan *plan)
{
    if (!ctx || !plan) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    plan->warm_pass = false;
    PEM_TRY(step3_impl(ctx, plan));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    PEM_TRY(launch_status(ctx));
    return step_elapsed(ctx, 4, 5, &ctx->timings.step3_ms);
}

// one iteration of the reference's timed loop (spgemm.cu:1136-1341): wall clock around
// step1+step2+step3 including every allocation and size read-back, ended by a device sync.
extern "C" pem_status pem_set_graph_replay(pem_ctx *ctx, int on)
{
    if (!ctx) return PEM_E_INVALID;
    ctx->graph_replay = on != 0;
    return PEM_OK;
}

extern "C" pem_status pem_spgemm(pem_ctx *ctx, pem_cplan *plan)
{
    if (!ctx || !plan) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    auto t0 = std::chrono::high_resolution_clock::now();
    // back-to-back steps share their boundary events (each record is a barrier packet, ~5 us of pipeline bubble)
    struct Chain {
        pem_ctx *c;
        explicit Chain(pem_ctx *c_) : c(c_) { c->chain_events = true; }
        ~Chain() { c->chain_events = false; }
    } chain(ctx);
    auto launch_verify = [&]() {
        if (plan->verify_folded) return;   // step 2's last kernel has already compared the sizes
        const WarmCheck wc = {1, plan->w_P, plan->w_Pall, plan->w_TC, plan->w_nnz, plan->w_nxl,
                              {plan->w_counts[0], plan->w_counts[1], plan->w_counts[2], plan->w_counts[3], plan->w_counts[4]}, plan->w_nsegs, nullptr};
        PEM_LAUNCH(ctx, warm_verify_kernel, 1, 1, reinterpret_cast<const long long *>(ctx->d_scalars), plan->bin_count.as<int>(), wc, ctx->d_flags);
    };
    // Graph replay (pem_set_graph_replay): a repeat pass has fixed grids, sizes and buffer addresses, so its ~28 launches,
    // the fork onto the auxiliary streams and the joins are captured once and replayed as one hipGraph -- the launch
    // gaps go (3 % of a 2.3 ms pass, 7 % of a 0.27 ms one).  The graph is re-captured whenever any device buffer
    // was (re)allocated since the capture.
    const bool use_graph = ctx->graph_replay && plan->warm && !ctx->profiling && !plan->opt_step1_esc && plan->opt_warm;
    bool graphed = false;
    if (use_graph && !plan->graph_failed) {
        if (plan->graph_exec && plan->graph_gen != ctx->arena->generation.load()) {
            retire_graph(ctx, plan);
        }
        if (!plan->graph_exec) {   // capture; any failure falls back to plain launches for the rest of the plan's life
            hipGraph_t graph = nullptr;
            pem_status cs = PEM_OK;
            if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                ctx->capturing = true;
                cs = step1_impl(ctx, plan, true);
                if (cs == PEM_OK) cs = step2_impl(ctx, plan);
                if (cs == PEM_OK) cs = step3_impl(ctx, plan);
                if (cs == PEM_OK) launch_verify();
                const hipError_t ee = hipStreamEndCapture(ctx->stream, &graph);
                ctx->capturing = false;
                if (cs == PEM_OK && ee == hipSuccess && graph &&
                    hipGraphInstantiate(&plan->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess)
                    plan->graph_gen = ctx->arena->generation.load();
                else
                    plan->graph_exec = nullptr;
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (!plan->graph_exec) {
                (void)hipGetLastError();
                ctx->launch_err = hipSuccess;   // (what failed inside the capture is retried below as plain launches)
                plan->graph_failed = true;
            }
        }
        if (plan->graph_exec) {
            PEM_HIP(hipGraphLaunch(plan->graph_exec, ctx->stream));
            int hf[NUM_FLAGS];
            if (plan->flags_mirrored) {   // the flags are already in host memory when the graph ends: no copy behind it
                PEM_HIP(hipStreamSynchronize(ctx->stream));
                for (int i = 0; i < NUM_FLAGS; ++i) hf[i] = ctx->h_flags[i];
            } else {
                PEM_TRY(read_flags(ctx, hf));
            }
            PEM_TRY(check_internal(hf));
            if (hf[FLAG_CAPACITY]) {   // sizes differ from the captured ones (cannot happen while A and B are immutable)
                retire_graph(ctx, plan);
                plan->warm = false;
            } else {
                graphed = true;
            }
        }
    }
    if (!graphed) {
        PEM_TRY(step1_impl(ctx, plan, true));
        PEM_TRY(step2_impl(ctx, plan));
        PEM_TRY(step3_impl(ctx, plan));
        if (plan->warm_pass) {
            // repeat pass: the host never waited for P / T_C / C_nnz; check on the device that they are what it assumed
            launch_verify();
            int hf[NUM_FLAGS];
            if (plan->flags_mirrored) {
                PEM_HIP(hipStreamSynchronize(ctx->stream));   // the pass's one synchronisation
                for (int i = 0; i < NUM_FLAGS; ++i) hf[i] = ctx->h_flags[i];
            } else {
                PEM_TRY(read_flags(ctx, hf));
            }
            PEM_TRY(check_internal(hf));
            if (hf[FLAG_CAPACITY]) {        // cannot happen while A and B are immutable; recover by a full pass
                plan->warm = false;
                PEM_TRY(step1_impl(ctx, plan, false));
                PEM_TRY(step2_impl(ctx, plan));
                PEM_TRY(step3_impl(ctx, plan));
            }
        }
    }
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    PEM_TRY(launch_status(ctx));
    plan->warm = plan->pairs_ready && plan->state == 3;
    ctx->timings.spgemm_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    if (graphed) {   // a replayed graph carries no per-step events (recorded inside a capture they do not time the replay)
        ctx->timings.step1_ms = ctx->timings.step2_ms = ctx->timings.step3_ms = 0.0;
        return PEM_OK;
    }
    PEM_TRY(step_elapsed(ctx, 0, 1, &ctx->timings.step1_ms));
    PEM_TRY(step_elapsed(ctx, 1, 3, &ctx->timings.step2_ms));
    PEM_TRY(step_elapsed(ctx, 3, 5, &ctx->timings.step3_ms));
    return PEM_OK;
}

extern "C" pem_status pem_cplan_get_array(pem_ctx *ctx, const pem_cplan *p, pem_cplan_array which, void *host_dst, int64_t bytes)
{
    if (!ctx || !p || (!host_dst && bytes > 0)) return PEM_E_INVALID;
    const size_t TC = (size_t)p->ntiles_c, P = (size_t)p->npairs, NZ = (size_t)p->nnz_c, mt = (size_t)(p->tr_hi - p->tr_lo);
    const void *src = nullptr;
    size_t want = 0;
    int need = 1;
    switch (which) {
    case PEM_C_TILE_ROWPTR: src = p->c_tile_rowptr.p; want = 4 * (mt + 1); break;
    case PEM_C_TILE_ROWIDX:
        PEM_TRY(ensure_c_rowidx(ctx, p));
        src = p->c_tile_rowidx.p;
        want = 4 * TC;
        break;
    case PEM_C_TILE_COLIDX:
        PEM_TRY(ensure_compact(ctx, p));
        src = p->c_tile_colidx.p;
        want = 4 * TC;
        break;
    case PEM_C_PAIRS_OFFSET:
        PEM_TRY(ensure_compact(ctx, p));
        src = p->pairs_offset.p;
        want = 4 * (TC + 1);
        break;
    case PEM_C_PAIRS_A: src = p->pairs_a.p; want = 4 * P; need = 2; break;
    case PEM_C_PAIRS_B: src = p->pairs_b.p; want = 4 * P; need = 2; break;
    case PEM_C_MASK: src = p->c_mask.p; want = 32 * TC; need = 2; break;
    case PEM_C_TILE_NNZ_PTR: src = p->c_tile_nnz_ptr.p; want = 4 * (TC + 1); need = 2; break;
    case PEM_C_ROWPTR:
        PEM_TRY(ensure_c_rowptr(ctx, p));
        src = p->c_rowptr.p;
        want = 16 * TC;
        need = 2;
        break;
    case PEM_C_ROWCOLIDX:
        PEM_TRY(ensure_c_rowcolidx(ctx, p));
        src = p->c_rowcolidx.p;
        want = NZ;
        need = 2;
        break;
    case PEM_C_VALS: src = p->c_vals.p; want = (size_t)p->A->value_bytes * NZ; need = 3; break;   // native type
    default: set_error("unknown pem_cplan_array %d", (int)which); return PEM_E_INVALID;
    }
    if (p->state < need) {
        set_error("pem_cplan_get_array(%d): step %d has not run", (int)which, need);
        return PEM_E_STATE;
    }
    if ((size_t)bytes != want) {
        set_error("pem_cplan_get_array(%d): caller passed %lld bytes, array has %zu", (int)which, (long long)bytes, want);
        return PEM_E_INVALID;
    }
    if (want == 0) return PEM_OK;
    PEM_ENTER(ctx);
    PEM_HIP(hipMemcpyAsync(host_dst, src, want, hipMemcpyDeviceToHost, ctx->stream));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    return PEM_OK;
}

// mgpu.hip -- libpemmgpu.so: one process, one pem_ctx per GPU, and the path's one exchange step -- the gather of the
// per-rank CSR slices of C to a root device over RCCL (include/pem_mgpu.h).  The reference is single-GPU,
// so nothing of it is replaced here.  Host code only: no kernels; device work goes through the C ABI of
// libpemspgemm_hip.so (export) and RCCL (transfer).  On the xGMI mesh every slice has its own direct link into the root.
#include "pem_internal.h"
#include "../../include/pem_mgpu.h"
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>
#include <rccl/rccl.h>

using namespace pem;

struct pem_mgpu {
    int n = 0;
    std::vector<int> dev;
    std::vector<pem_ctx *> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;            // the transfers' streams, one per device
    std::vector<DevBuf> s_rp, s_ci, s_v;        // per rank: the slice as CSR on its own device
    DevBuf r_ci, r_v;                           // root: the assembled column indices / values
    // the chunked, overlapped pass (pem_mgpu_spgemm_gather_chunked)
    std::vector<DevBuf> c_rp, c_ci, c_v;        // per (rank, chunk): the chunk as CSR on its own device (non-root ranks: send staging)
    std::vector<hipEvent_t> c_ev;               // per (rank, chunk): the export is done
    std::vector<const pem_cplan *> c_plans;     // the plans the cached sizes belong to
    std::vector<int64_t> c_rows, c_nnz;
};

#define PEM_NCCL(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            pem::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r));  \
            return PEM_E_HIP;                                                                      \
        }                                                                                          \
    } while (0)

extern "C" pem_status pem_mgpu_create(int ndev, const int *devices, pem_mgpu **out)
{
    if (!out || ndev < 1 || !devices) return PEM_E_INVALID;
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) {
        set_error("pem_mgpu_create: no usable GPU (the product path has no CPU fallback)");
        return PEM_E_NODEVICE;
    }
    for (int g = 0; g < ndev; ++g) {
        if (devices[g] < 0 || devices[g] >= have) {
            set_error("pem_mgpu_create: device %d requested, %d visible", devices[g], have);
            return PEM_E_INVALID;
        }
        for (int h = 0; h < g; ++h)
            if (devices[h] == devices[g]) {
                set_error("pem_mgpu_create: device %d listed twice (one rank per GPU)", devices[g]);
                return PEM_E_INVALID;
            }
    }
    pem_mgpu *m = new pem_mgpu();
    m->n = ndev;
    m->dev.assign(devices, devices + ndev);
    m->ctx.assign((size_t)ndev, nullptr);
    m->comm.assign((size_t)ndev, nullptr);
    m->stream.assign((size_t)ndev, nullptr);
    m->s_rp = std::vector<DevBuf>((size_t)ndev);
    m->s_ci = std::vector<DevBuf>((size_t)ndev);
    m->s_v = std::vector<DevBuf>((size_t)ndev);
    auto fail = [&](pem_status s) {
        pem_mgpu_destroy(m);
        return s;
    };
    for (int g = 0; g < ndev; ++g) {
        pem_status s = pem_ctx_create(devices[g], &m->ctx[(size_t)g]);
        if (s != PEM_OK) return fail(s);
        if (hipSetDevice(devices[g]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[(size_t)g], hipStreamNonBlocking) != hipSuccess) {
            set_error("pem_mgpu_create: stream on device %d", devices[g]);
            return fail(PEM_E_HIP);
        }
    }
    ncclResult_t r = ncclCommInitAll(m->comm.data(), ndev, devices);
    if (r != ncclSuccess) {
        set_error("ncclCommInitAll over %d devices: %s", ndev, ncclGetErrorString(r));
        for (auto &c : m->comm) c = nullptr;
        return fail(PEM_E_HIP);
    }
    *out = m;
    return PEM_OK;
}

extern "C" pem_status pem_mgpu_destroy(pem_mgpu *m)
{
    if (!m) return PEM_OK;
    for (int g = 0; g < m->n; ++g) {
        (void)hipSetDevice(m->dev[(size_t)g]);
        if (m->stream[(size_t)g]) (void)hipStreamSynchronize(m->stream[(size_t)g]);
        if (m->comm[(size_t)g]) (void)ncclCommDestroy(m->comm[(size_t)g]);
        m->s_rp[(size_t)g].release();
        m->s_ci[(size_t)g].release();
        m->s_v[(size_t)g].release();
        if (m->stream[(size_t)g]) (void)hipStreamDestroy(m->stream[(size_t)g]);
    }
    m->r_ci.release();
    m->r_v.release();
    for (auto &b : m->c_rp) b.release();
    for (auto &b : m->c_ci) b.release();
    for (auto &b : m->c_v) b.release();
    for (auto e : m->c_ev)
        if (e) (void)hipEventDestroy(e);
    for (int g = 0; g < m->n; ++g)
        if (m->ctx[(size_t)g]) (void)pem_ctx_destroy(m->ctx[(size_t)g]);
    delete m;
    return PEM_OK;
}

extern "C" int pem_mgpu_size(const pem_mgpu *m) { return m ? m->n : 0; }
extern "C" pem_ctx *pem_mgpu_ctx(pem_mgpu *m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[(size_t)rank] : nullptr; }

extern "C" void pem_mgpu_slice_offsets(int n, const int64_t *nrows, const int64_t *nnz, int64_t *row_off, int64_t *nnz_off)
{
    row_off[0] = nnz_off[0] = 0;
    for (int g = 0; g < n; ++g) {
        row_off[g + 1] = row_off[g] + nrows[g];
        nnz_off[g + 1] = nnz_off[g] + nnz[g];
    }
}

extern "C" void pem_mgpu_rebase_rowptr(int n, const int64_t *nrows, const int32_t *const *slice_rowptr, const int64_t *row_off,
                                       const int64_t *nnz_off, int32_t *out)
{
    for (int g = 0; g < n; ++g)
        for (int64_t r = 0; r < nrows[g]; ++r) out[row_off[g] + r] = (int32_t)((int64_t)slice_rowptr[g][r] + nnz_off[g]);
    out[row_off[n]] = (int32_t)nnz_off[n];
}

extern "C" pem_status pem_mgpu_gather_csr(pem_mgpu *m, pem_cplan *const *plans, int root, int64_t *nrows_out, int64_t *nnz_out,
                                          int32_t *rowptr, int32_t *colidx, double *vals, double *gather_ms)
{
    if (!m || !plans || root < 0 || root >= m->n) return PEM_E_INVALID;
    const int n = m->n;
    if (n > 1 && !m->comm[(size_t)root]) {
        set_error("pem_mgpu_gather_csr: the communicators were aborted by an earlier failed exchange");
        return PEM_E_STATE;
    }
    std::vector<int64_t> nrows((size_t)n), nnz((size_t)n), row_off((size_t)n + 1), nnz_off((size_t)n + 1);
    for (int g = 0; g < n; ++g) {
        if (!plans[g]) return PEM_E_INVALID;
        pem_cplan_info ci;
        PEM_TRY(pem_cplan_get_info(plans[g], &ci));
        if (g > 0) {
            pem_cplan_info prev;
            PEM_TRY(pem_cplan_get_info(plans[g - 1], &prev));
            if (prev.tile_row_end != ci.tile_row_begin) {
                set_error("pem_mgpu_gather_csr: plan %d covers tile rows [%d, %d), plan %d ends at %d -- row blocks must abut in rank order", g,
                          ci.tile_row_begin, ci.tile_row_end, g - 1, prev.tile_row_end);
                return PEM_E_INVALID;
            }
        }
        nrows[(size_t)g] = ci.row_end - ci.row_begin;
        nnz[(size_t)g] = ci.nnz_c;
    }
    pem_mgpu_slice_offsets(n, nrows.data(), nnz.data(), row_off.data(), nnz_off.data());
    if (nnz_off[(size_t)n] > 0x7FFFFFFFll) {
        set_error("pem_mgpu_gather_csr: the assembled C has %lld nonzeros, beyond int32 row pointers", (long long)nnz_off[(size_t)n]);
        return PEM_E_OVERFLOW;
    }
    if (nrows_out) *nrows_out = row_off[(size_t)n];
    if (nnz_out) *nnz_out = nnz_off[(size_t)n];
    if (!rowptr) return PEM_OK;   // size query
    const size_t total = (size_t)nnz_off[(size_t)n];
    if (total && (!colidx || !vals)) return PEM_E_INVALID;

    // buffers first (each from its own device's arena), so that the clock below times the exchange and not a first call's
    // device allocations
    {
        PEM_HIP(hipSetDevice(m->dev[(size_t)root]));
        ArenaBind bind(m->ctx[(size_t)root]->arena);
        PEM_TRY(m->r_ci.reserve(sizeof(int32_t) * (total + 4)));
        PEM_TRY(m->r_v.reserve(sizeof(double) * (total + 1)));
    }
    for (int g = 0; g < n; ++g) {
        PEM_HIP(hipSetDevice(m->dev[(size_t)g]));
        ArenaBind bind(m->ctx[(size_t)g]->arena);
        PEM_TRY(m->s_rp[(size_t)g].reserve(sizeof(int32_t) * ((size_t)nrows[(size_t)g] + 4)));
        if (g != root) {
            PEM_TRY(m->s_ci[(size_t)g].reserve(sizeof(int32_t) * ((size_t)nnz[(size_t)g] + 4)));
            PEM_TRY(m->s_v[(size_t)g].reserve(sizeof(double) * ((size_t)nnz[(size_t)g] + 1)));
        }
    }
    const auto t0 = std::chrono::high_resolution_clock::now();
    // every rank: tiled C slice -> CSR on its own device (the root writes its slice straight into the assembled arrays)
    for (int g = 0; g < n; ++g) {
        PEM_HIP(hipSetDevice(m->dev[(size_t)g]));
        int32_t *ci = nullptr;
        double *v = nullptr;
        if (g == root) {
            ci = m->r_ci.as<int32_t>() + nnz_off[(size_t)g];
            v = m->r_v.as<double>() + nnz_off[(size_t)g];
        } else {
            ci = m->s_ci[(size_t)g].as<int32_t>();
            v = m->s_v[(size_t)g].as<double>();
        }
        PEM_TRY(pem_c_export_csr_device(m->ctx[(size_t)g], plans[g], m->s_rp[(size_t)g].as<int32_t>(), ci, v));
    }
    for (int g = 0; g < n; ++g) PEM_TRY(pem_ctx_synchronize(m->ctx[(size_t)g]));   // the exports ran on the contexts' streams
    // one RCCL group: slice g -> its place in the root's arrays.  Nothing returns from inside the group: the first failure is
    // noted, the group is always closed, and a failed exchange aborts the communicators (an open or half-issued group
    // leaves every later call on them -- ncclCommDestroy included -- undefined)
    if (n > 1) {
        pem_status gs = PEM_OK;
        auto nccl_ok = [&](ncclResult_t r, const char *what) {
            if (r != ncclSuccess && gs == PEM_OK) {
                set_error("pem_mgpu_gather_csr: %s -> %s", what, ncclGetErrorString(r));
                gs = PEM_E_HIP;
            }
        };
        auto hip_ok = [&](hipError_t e, const char *what) {
            if (e != hipSuccess && gs == PEM_OK) {
                set_error("pem_mgpu_gather_csr: %s -> %s", what, hipGetErrorString(e));
                gs = PEM_E_HIP;
            }
        };
        PEM_NCCL(ncclGroupStart());
        for (int g = 0; g < n && gs == PEM_OK; ++g) {
            if (g == root || nnz[(size_t)g] == 0) continue;
            hip_ok(hipSetDevice(m->dev[(size_t)g]), "hipSetDevice(sender)");
            if (gs != PEM_OK) break;
            nccl_ok(ncclSend(m->s_ci[(size_t)g].p, (size_t)nnz[(size_t)g], ncclInt32, root, m->comm[(size_t)g], m->stream[(size_t)g]), "ncclSend(colidx)");
            nccl_ok(ncclSend(m->s_v[(size_t)g].p, (size_t)nnz[(size_t)g], ncclFloat64, root, m->comm[(size_t)g], m->stream[(size_t)g]), "ncclSend(vals)");
        }
        if (gs == PEM_OK) hip_ok(hipSetDevice(m->dev[(size_t)root]), "hipSetDevice(root)");
        for (int g = 0; g < n && gs == PEM_OK; ++g) {
            if (g == root || nnz[(size_t)g] == 0) continue;
            nccl_ok(ncclRecv(m->r_ci.as<int32_t>() + nnz_off[(size_t)g], (size_t)nnz[(size_t)g], ncclInt32, g, m->comm[(size_t)root],
                             m->stream[(size_t)root]), "ncclRecv(colidx)");
            nccl_ok(ncclRecv(m->r_v.as<double>() + nnz_off[(size_t)g], (size_t)nnz[(size_t)g], ncclFloat64, g, m->comm[(size_t)root],
                             m->stream[(size_t)root]), "ncclRecv(vals)");
        }
        nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
        if (gs != PEM_OK) {
            for (int g = 0; g < n; ++g)
                if (m->comm[(size_t)g]) {
                    (void)ncclCommAbort(m->comm[(size_t)g]);
                    m->comm[(size_t)g] = nullptr;      // pem_mgpu_destroy skips it; the handle is dead for further gathers
                }
            return gs;
        }
        for (int g = 0; g < n; ++g) {
            PEM_HIP(hipSetDevice(m->dev[(size_t)g]));
            PEM_HIP(hipStreamSynchronize(m->stream[(size_t)g]));
        }
    }
    if (gather_ms) *gather_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    // row pointers: a few MB, copied out by their owners and rebased on the host
    std::vector<std::vector<int32_t>> rp((size_t)n);
    std::vector<const int32_t *> rpp((size_t)n);
    for (int g = 0; g < n; ++g) {
        rp[(size_t)g].resize((size_t)nrows[(size_t)g] + 1);
        PEM_HIP(hipSetDevice(m->dev[(size_t)g]));
        PEM_HIP(hipMemcpy(rp[(size_t)g].data(), m->s_rp[(size_t)g].p, sizeof(int32_t) * ((size_t)nrows[(size_t)g] + 1), hipMemcpyDeviceToHost));
        rpp[(size_t)g] = rp[(size_t)g].data();
    }
    pem_mgpu_rebase_rowptr(n, nrows.data(), rpp.data(), row_off.data(), nnz_off.data(), rowptr);
    if (total) {
        PEM_HIP(hipSetDevice(m->dev[(size_t)root]));
        PEM_HIP(hipMemcpy(colidx, m->r_ci.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
        PEM_HIP(hipMemcpy(vals, m->r_v.p, sizeof(double) * total, hipMemcpyDeviceToHost));
    }
    return PEM_OK;
}

extern "C" pem_status pem_mgpu_recut_bounds(int nparts, int mt, const double *weights, const int32_t *bounds, const double *ms, double fixed_ms,
                                            int32_t *out)
{
    if (nparts < 1 || mt < 0 || !weights || !bounds || !ms || !out || bounds[0] != 0 || bounds[nparts] != mt) return PEM_E_INVALID;
    for (int p = 0; p < nparts; ++p)
        if (bounds[p + 1] < bounds[p]) return PEM_E_INVALID;
    std::vector<double> cpre((size_t)mt + 1, 0.0);
    for (int p = 0; p < nparts; ++p) {
        double part_w = 0.0;
        for (int i = bounds[p]; i < bounds[p + 1]; ++i) part_w += weights[i];
        const double var = std::max(ms[p] - fixed_ms, 0.05 * ms[p]);      // (a rank is never modelled as costing nothing)
        const double rate = part_w > 0.0 ? var / part_w : 0.0;
        for (int i = bounds[p]; i < bounds[p + 1]; ++i) cpre[(size_t)i + 1] = weights[i] * rate;
    }
    for (int i = 0; i < mt; ++i) cpre[(size_t)i + 1] += cpre[(size_t)i];
    out[0] = 0;
    int row = 0;
    for (int g = 1; g < nparts; ++g) {
        const double target = cpre[(size_t)mt] * (double)g / (double)nparts;
        while (row < mt && cpre[(size_t)row + 1] <= target) ++row;
        out[g] = row;
    }
    out[nparts] = mt;
    return PEM_OK;
}

namespace {
struct SpinBarrier {                    // the ranks' host threads of one pass
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    unsigned long gen = 0;
    explicit SpinBarrier(int n_) : n(n_) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long g = gen;
        if (++waiting == n) {
            waiting = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
    }
};
}   // namespace

extern "C" pem_status pem_mgpu_spgemm_gather_chunked(pem_mgpu *m, pem_cplan *const *plans, int nchunks, int root, int64_t *nrows_out,
                                                     int64_t *nnz_out, int32_t *rowptr, int32_t *colidx, double *vals, double *pass_ms,
                                                     double *tail_ms)
{
    if (!m || !plans || nchunks < 1 || root < 0 || root >= m->n) return PEM_E_INVALID;
    const int n = m->n, K = nchunks, S = n * K;
    if (n > 1 && !m->comm[(size_t)root]) {
        set_error("pem_mgpu_spgemm_gather_chunked: the communicators were aborted by an earlier failed exchange");
        return PEM_E_STATE;
    }
    for (int s = 0; s < S; ++s) {
        if (!plans[s]) return PEM_E_INVALID;
        if (plans[s]->owner != m->ctx[(size_t)(s / K)]) {
            set_error("pem_mgpu_spgemm_gather_chunked: plan %d was not created on rank %d's context", s, s / K);
            return PEM_E_INVALID;
        }
        if (plans[s]->A->value_bytes != 8) {
            set_error("pem_mgpu_spgemm_gather_chunked: fp64 plans only");
            return PEM_E_INVALID;
        }
        if (s > 0 && plans[s - 1]->tr_hi != plans[s]->tr_lo) {
            set_error("pem_mgpu_spgemm_gather_chunked: chunk %d covers tile rows [%d, %d), chunk %d ends at %d -- chunks must abut in (rank, chunk) order",
                      s, plans[s]->tr_lo, plans[s]->tr_hi, s - 1, plans[s - 1]->tr_hi);
            return PEM_E_INVALID;
        }
    }
    std::vector<pem_status> st((size_t)n, PEM_OK);
    std::vector<std::string> err((size_t)n);
    auto run_ranks = [&](const std::function<pem_status(int)> &fn) -> pem_status {
        std::vector<std::thread> th;
        for (int g = 0; g < n; ++g)
            th.emplace_back([&, g] {
                st[(size_t)g] = fn(g);
                if (st[(size_t)g] != PEM_OK) err[(size_t)g] = pem_last_error();
            });
        for (auto &t : th) t.join();
        for (int g = 0; g < n; ++g)
            if (st[(size_t)g] != PEM_OK) {
                set_error("rank %d: %s", g, err[(size_t)g].c_str());
                return st[(size_t)g];
            }
        return PEM_OK;
    };
    // sizes of the chunks: known from an earlier call on the same plans, else one pass over them now (it also warms the plans)
    bool cached = (int)m->c_plans.size() == S;
    for (int s = 0; s < S && cached; ++s) cached = m->c_plans[(size_t)s] == plans[s];
    if (!cached) {
        PEM_TRY(run_ranks([&](int g) -> pem_status {
            for (int c = 0; c < K; ++c) PEM_TRY(pem_spgemm(m->ctx[(size_t)g], plans[g * K + c]));
            return PEM_OK;
        }));
        m->c_plans.assign(plans, plans + S);
        m->c_rows.assign((size_t)S, 0);
        m->c_nnz.assign((size_t)S, 0);
        for (int s = 0; s < S; ++s) {
            pem_cplan_info ci;
            PEM_TRY(pem_cplan_get_info(plans[s], &ci));
            m->c_rows[(size_t)s] = ci.row_end - ci.row_begin;
            m->c_nnz[(size_t)s] = ci.nnz_c;
        }
    }
    std::vector<int64_t> row_off((size_t)S + 1), nnz_off((size_t)S + 1);
    pem_mgpu_slice_offsets(S, m->c_rows.data(), m->c_nnz.data(), row_off.data(), nnz_off.data());
    if (nnz_off[(size_t)S] > 0x7FFFFFFFll) {
        set_error("pem_mgpu_spgemm_gather_chunked: the assembled C has %lld nonzeros, beyond int32 row pointers", (long long)nnz_off[(size_t)S]);
        return PEM_E_OVERFLOW;
    }
    if (nrows_out) *nrows_out = row_off[(size_t)S];
    if (nnz_out) *nnz_out = nnz_off[(size_t)S];
    const size_t total = (size_t)nnz_off[(size_t)S];
    if (rowptr && total && (!colidx || !vals)) return PEM_E_INVALID;
    // buffers first, each from its own device's arena: the clock below times the pass, not first-call allocations
    if ((int)m->c_rp.size() != S) {
        for (auto e : m->c_ev)
            if (e) (void)hipEventDestroy(e);
        m->c_rp = std::vector<DevBuf>((size_t)S);
        m->c_ci = std::vector<DevBuf>((size_t)S);
        m->c_v = std::vector<DevBuf>((size_t)S);
        m->c_ev.assign((size_t)S, nullptr);
    }
    {
        PEM_HIP(hipSetDevice(m->dev[(size_t)root]));
        ArenaBind bind(m->ctx[(size_t)root]->arena);
        PEM_TRY(m->r_ci.reserve(sizeof(int32_t) * (total + 4)));
        PEM_TRY(m->r_v.reserve(sizeof(double) * (total + 1)));
    }
    for (int s = 0; s < S; ++s) {
        const int g = s / K;
        PEM_HIP(hipSetDevice(m->dev[(size_t)g]));
        ArenaBind bind(m->ctx[(size_t)g]->arena);
        PEM_TRY(m->c_rp[(size_t)s].reserve(sizeof(int32_t) * ((size_t)m->c_rows[(size_t)s] + 4)));
        if (g != root) {
            PEM_TRY(m->c_ci[(size_t)s].reserve(sizeof(int32_t) * ((size_t)m->c_nnz[(size_t)s] + 4)));
            PEM_TRY(m->c_v[(size_t)s].reserve(sizeof(double) * ((size_t)m->c_nnz[(size_t)s] + 1)));
        }
        if (!m->c_ev[(size_t)s]) PEM_HIP(hipEventCreateWithFlags(&m->c_ev[(size_t)s], hipEventDisableTiming));
    }
    // the pass.  Rank g, chunk c: steps 1-3, CSR export on the device (the root's chunks straight into the assembled arrays), and --
    // behind an event, on the rank's transfer stream -- the chunk's two sends, which travel while chunk c + 1 computes.  The root
    // posts its receives chunk by chunk before it starts computing (one RCCL group per chunk index, every other rank's chunk c
    // straight into its place).  A failed RCCL call aborts the communicators, as in pem_mgpu_gather_csr.
    SpinBarrier bar(n);
    std::chrono::high_resolution_clock::time_point t_start, t_compute, t_all;
    std::atomic<int> rccl_failed{0};
    const pem_status rs = run_ranks([&](int g) -> pem_status {
        pem_ctx *ctx = m->ctx[(size_t)g];
        PEM_HIP(hipSetDevice(m->dev[(size_t)g]));
        pem_status mine = PEM_OK;
        auto nccl_ok = [&](ncclResult_t r, const char *what) {
            if (r != ncclSuccess && mine == PEM_OK) {
                set_error("pem_mgpu_spgemm_gather_chunked: %s -> %s", what, ncclGetErrorString(r));
                mine = PEM_E_HIP;
                rccl_failed = 1;
            }
        };
        bar.wait();
        if (g == 0) t_start = std::chrono::high_resolution_clock::now();
        if (g == root && n > 1) {
            for (int c = 0; c < K; ++c) {
                nccl_ok(ncclGroupStart(), "ncclGroupStart");
                for (int h = 0; h < n; ++h) {
                    const int s = h * K + c;
                    if (h == root || m->c_nnz[(size_t)s] == 0) continue;
                    nccl_ok(ncclRecv(m->r_ci.as<int32_t>() + nnz_off[(size_t)s], (size_t)m->c_nnz[(size_t)s], ncclInt32, h, m->comm[(size_t)root],
                                     m->stream[(size_t)root]), "ncclRecv(colidx)");
                    nccl_ok(ncclRecv(m->r_v.as<double>() + nnz_off[(size_t)s], (size_t)m->c_nnz[(size_t)s], ncclFloat64, h, m->comm[(size_t)root],
                                     m->stream[(size_t)root]), "ncclRecv(vals)");
                }
                nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
            }
        }
        for (int c = 0; c < K && mine == PEM_OK; ++c) {
            const int s = g * K + c;
            pem_cplan *plan = plans[s];
            mine = pem_spgemm(ctx, plan);
            if (mine != PEM_OK) break;
            if (plan->nnz_c != m->c_nnz[(size_t)s]) {
                set_error("pem_mgpu_spgemm_gather_chunked: chunk %d has %lld entries, the sizing pass saw %lld", s, (long long)plan->nnz_c,
                          (long long)m->c_nnz[(size_t)s]);
                mine = PEM_E_STATE;
                break;
            }
            int32_t *ci = g == root ? m->r_ci.as<int32_t>() + nnz_off[(size_t)s] : m->c_ci[(size_t)s].as<int32_t>();
            double *v = g == root ? m->r_v.as<double>() + nnz_off[(size_t)s] : m->c_v[(size_t)s].as<double>();
            mine = pem_c_export_csr_device(ctx, plan, m->c_rp[(size_t)s].as<int32_t>(), ci, v);
            if (mine != PEM_OK) break;
            if (g != root && n > 1 && m->c_nnz[(size_t)s] > 0) {
                if (hipEventRecord(m->c_ev[(size_t)s], ctx->stream) != hipSuccess || hipStreamWaitEvent(m->stream[(size_t)g], m->c_ev[(size_t)s], 0) != hipSuccess) {
                    set_error("pem_mgpu_spgemm_gather_chunked: event between the export and the send");
                    mine = PEM_E_HIP;
                    break;
                }
                nccl_ok(ncclGroupStart(), "ncclGroupStart");
                nccl_ok(ncclSend(ci, (size_t)m->c_nnz[(size_t)s], ncclInt32, root, m->comm[(size_t)g], m->stream[(size_t)g]), "ncclSend(colidx)");
                nccl_ok(ncclSend(v, (size_t)m->c_nnz[(size_t)s], ncclFloat64, root, m->comm[(size_t)g], m->stream[(size_t)g]), "ncclSend(vals)");
                nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
            }
        }
        if (mine == PEM_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) mine = PEM_E_HIP;
        bar.wait();                                     // every rank has finished its last chunk's steps 1-3 (and export)
        if (g == 0) t_compute = std::chrono::high_resolution_clock::now();
        if (!rccl_failed && hipStreamSynchronize(m->stream[(size_t)g]) != hipSuccess && mine == PEM_OK) mine = PEM_E_HIP;
        bar.wait();
        if (g == 0) t_all = std::chrono::high_resolution_clock::now();
        return mine;
    });
    if (rccl_failed) {
        for (int g = 0; g < n; ++g)
            if (m->comm[(size_t)g]) {
                (void)ncclCommAbort(m->comm[(size_t)g]);
                m->comm[(size_t)g] = nullptr;
            }
    }
    PEM_TRY(rs);
    if (pass_ms) *pass_ms = std::chrono::duration<double, std::milli>(t_compute - t_start).count();
    if (tail_ms) *tail_ms = std::chrono::duration<double, std::milli>(t_all - t_compute).count();
    if (!rowptr) return PEM_OK;
    // row pointers: a few MB, copied out by their owners and rebased on the host
    std::vector<std::vector<int32_t>> rp((size_t)S);
    std::vector<const int32_t *> rpp((size_t)S);
    for (int s = 0; s < S; ++s) {
        rp[(size_t)s].resize((size_t)m->c_rows[(size_t)s] + 1);
        PEM_HIP(hipSetDevice(m->dev[(size_t)(s / K)]));
        PEM_HIP(hipMemcpy(rp[(size_t)s].data(), m->c_rp[(size_t)s].p, sizeof(int32_t) * ((size_t)m->c_rows[(size_t)s] + 1), hipMemcpyDeviceToHost));
        rpp[(size_t)s] = rp[(size_t)s].data();
    }
    pem_mgpu_rebase_rowptr(S, m->c_rows.data(), rpp.data(), row_off.data(), nnz_off.data(), rowptr);
    if (total) {
        PEM_HIP(hipSetDevice(m->dev[(size_t)root]));
        PEM_HIP(hipMemcpy(colidx, m->r_ci.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
        PEM_HIP(hipMemcpy(vals, m->r_v.p, sizeof(double) * total, hipMemcpyDeviceToHost));
    }
    return PEM_OK;
}

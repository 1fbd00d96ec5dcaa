// pemspgemm -- command-line front end with the reference's surface (spgemm.cu:720-1568):
//     pemspgemm <matrix.mtx> [0|1] [any third argument => C = A*A^T]
// reads one Matrix-Market file, forms C = A*A (or A*A^T), times WARMUP + REPEAT passes of
// step1+step2+step3, prints the reference's report, appends the reference's 14 CSV columns
// to ./pemspgemm_benchmark_result.csv and, when the 2nd argument is non-zero, writes the
// sorted COO result to /tmp/SPGEMM_RESULT_{NNZ,ROWS,COLS,VALS}.txt.
// Host C++ over the C ABI of libpemspgemm_hip.so; no HIP calls in this file.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <vector>
#include "../../include/pem_host.h"
#include "../../include/pem_mgpu.h"
#include "../../include/pem_spgemm.h"

static int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return s && *s ? atoi(s) : dflt;
}

#define CHECK(expr)                                                              \
    do {                                                                         \
        pem_status _s = (expr);                                                  \
        if (_s != PEM_OK) {                                                      \
            fprintf(stderr, "pemspgemm: %s failed (%d): %s\n", #expr, _s, pem_last_error()); \
            return 2;                                                            \
        }                                                                        \
    } while (0)

// ------------------------------------------------------------------------------------------------------------------
// --gpus N (beyond the single-GPU reference): one host thread and one context per device, A split into N
// tile-row blocks balanced on tile-level products, B whole on every device, steps 1-3 with no communication, then the
// path's one exchange step: the CSR slices gathered to device 0 over RCCL (libpemmgpu.so).  A pass is timed from the
// moment all ranks start to the moment the last one finishes; the gather is timed on its own (t_gather).
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct Barrier {
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    unsigned long gen = 0;
    explicit Barrier(int n_) : n(n_) {}
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

struct MultiResult {
    double step1 = 0, step2 = 0, step3 = 0, total = 0, cold = 0, gather_ms = 0, conv_ms = 0;
    double a_conv_kernel_ms = 0, b_conv_kernel_ms = 0;
    uint64_t flop = 0;
    int64_t ntiles_c = 0, npairs = 0, nnz_c = 0, rows = 0;
    int32_t cols_b = 0;
    std::vector<int32_t> rowptr, colidx;
    std::vector<double> vals;
    bool gathered = false;
    // the measured re-cut of the row split, and the chunked pass with the gather overlapped (libpemmgpu.so)
    int tune_rounds = 0;
    double tune_first_max = 0, tune_best_max = 0;
    std::vector<int32_t> bounds;
    int chunks = 0;
    double chunk_pass_ms = 0, chunk_tail_ms = 0;
};
}   // namespace

static int run_multi(int ngpu, const pem_coo &ca, const pem_coo *cb, bool aat, bool want_c, int WARMUP, int REPEAT, bool fastest,
                     std::chrono::high_resolution_clock::time_point conv_start, MultiResult &res, bool tune_split, int nchunks)
{
    std::vector<int> devs((size_t)ngpu);
    for (int g = 0; g < ngpu; ++g) devs[(size_t)g] = g;
    pem_mgpu *m = nullptr;
    if (pem_mgpu_create(ngpu, devs.data(), &m) != PEM_OK) {
        fprintf(stderr, "pemspgemm: --gpus %d: %s\n", ngpu, pem_last_error());
        return 2;
    }
    std::vector<pem_tiled *> A((size_t)ngpu, nullptr), B((size_t)ngpu, nullptr);
    std::vector<pem_cplan *> plan((size_t)ngpu, nullptr);
    std::vector<int32_t> bounds((size_t)ngpu + 1, 0);
    std::vector<std::vector<double>> s1((size_t)ngpu), s2((size_t)ngpu), s3((size_t)ngpu);
    std::vector<double> wall;
    std::atomic<int> failed{0};
    std::vector<std::string> errs((size_t)ngpu);
    Barrier bar(ngpu);
    std::vector<int32_t> cur((size_t)ngpu + 1, 0);          // the cut being timed (split tuning)
    std::vector<double> weights, tune_ms((size_t)ngpu, 0.0);
    int mt_all = 0;
    bool tune_done = false;
    const pem_coo &b_src = cb ? *cb : ca;
    auto worker = [&](int g) {
        pem_ctx *ctx = pem_mgpu_ctx(m, g);
        auto fail = [&](const char *what) {
            errs[(size_t)g] = std::string(what) + ": " + pem_last_error();
            failed = 1;
        };
        // every device tiles B whole (replicated) and A whole (its plan covers only its tile-row block)
        if (pem_tiled_from_coo(ctx, ca.rows, ca.cols, ca.nnz, ca.I, ca.J, ca.V, 0, &A[(size_t)g]) != PEM_OK) fail("conversion of A");
        if (!failed) {
            if (cb || aat) {
                if (pem_tiled_from_coo(ctx, b_src.rows, b_src.cols, b_src.nnz, b_src.I, b_src.J, b_src.V, aat ? 1 : 0, &B[(size_t)g]) != PEM_OK)
                    fail("conversion of B");
            } else {
                B[(size_t)g] = A[(size_t)g];
            }
        }
        bar.wait();
        if (g == 0 && !failed) {
            pem_tiled_info ia, ib;
            pem_tiled_get_info(A[0], &ia);
            pem_tiled_get_info(B[0], &ib);
            if (ia.cols != ib.rows) {
                errs[0] = "inner dimensions differ";
                failed = 1;
            } else if (pem_split_tile_rows(ctx, A[0], B[0], ngpu, bounds.data()) != PEM_OK || pem_flop_count(ctx, A[0], B[0], &res.flop) != PEM_OK) {
                fail("row split");
            } else {
                mt_all = ia.tile_rows;
                weights.resize((size_t)mt_all);
                if (tune_split && ngpu > 1 && pem_tile_row_weights(ctx, A[0], B[0], weights.data()) != PEM_OK) fail("tile-row weights");
                cur = bounds;
            }
            res.rows = ia.rows;
            res.cols_b = ib.cols;
            res.a_conv_kernel_ms = ia.conv_tile_kernel_ms;
            res.b_conv_kernel_ms = ib.conv_tile_kernel_ms;
            res.conv_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - conv_start).count();
        }
        bar.wait();
        // Measure-and-recut of the row split (setup, before anything is timed; the Python harness' multigpu.tune_row_bounds): every
        // rank times twenty repeat passes of its block as a replayed graph, rank 0 re-cuts the blocks from the times
        // (pem_mgpu_recut_bounds) so that ranks whose rows cost more per unit of weight get fewer of them; up to three rounds, and a
        // later cut is only kept if it beats the first by 3 % (pass times move by that much from run to run).
        if (tune_split && ngpu > 1) {
            for (int rnd = 0; rnd <= 3 && !failed; ++rnd) {
                pem_cplan *tp = nullptr;
                if (pem_cplan_create(ctx, A[(size_t)g], B[(size_t)g], cur[(size_t)g], cur[(size_t)g + 1], &tp) != PEM_OK) fail("plan (split tuning)");
                pem_set_graph_replay(ctx, 1);
                for (int k = 0; k < 3 && !failed; ++k)
                    if (pem_spgemm(ctx, tp) != PEM_OK) fail("pem_spgemm (split tuning)");
                bar.wait();
                const auto t0 = std::chrono::high_resolution_clock::now();
                for (int k = 0; k < 20 && !failed; ++k)
                    if (pem_spgemm(ctx, tp) != PEM_OK) fail("pem_spgemm (split tuning)");
                tune_ms[(size_t)g] = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count() / 20.0;
                pem_set_graph_replay(ctx, 0);
                if (tp) pem_cplan_destroy(ctx, tp);
                bar.wait();
                if (g == 0 && !failed) {
                    const double mx = *std::max_element(tune_ms.begin(), tune_ms.end()), mn = *std::min_element(tune_ms.begin(), tune_ms.end());
                    double mean = 0;
                    for (double x : tune_ms) mean += x / (double)ngpu;
                    if (rnd == 0) {
                        res.tune_first_max = res.tune_best_max = mx;
                        bounds = cur;
                    } else if (mx < res.tune_best_max && mx < 0.97 * res.tune_first_max) {
                        res.tune_best_max = mx;
                        bounds = cur;
                    }
                    res.tune_rounds = rnd + 1;
                    tune_done = rnd == 3 || mx <= 1.04 * mean;
                    if (!tune_done) {
                        std::vector<int32_t> nxt((size_t)ngpu + 1);
                        if (pem_mgpu_recut_bounds(ngpu, mt_all, weights.data(), cur.data(), tune_ms.data(), (rnd == 0 ? 0.6 : 0.5) * mn, nxt.data()) != PEM_OK)
                            tune_done = true;
                        else
                            cur = nxt;
                    }
                }
                bar.wait();
                if (tune_done) break;
            }
        }
        if (!failed && pem_cplan_create(ctx, A[(size_t)g], B[(size_t)g], bounds[(size_t)g], bounds[(size_t)g + 1], &plan[(size_t)g]) != PEM_OK)
            fail("plan");
        for (int n = 0; n < WARMUP + REPEAT; ++n) {
            bar.wait();
            const auto t0 = std::chrono::high_resolution_clock::now();
            if (!failed && pem_spgemm(ctx, plan[(size_t)g]) != PEM_OK) fail("pem_spgemm");
            bar.wait();   // the pass ends when the last rank has finished
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
            pem_timings t;
            memset(&t, 0, sizeof t);
            pem_get_timings(ctx, &t);
            if (g == 0 && n == 0) res.cold = ms;
            if (n >= WARMUP) {
                s1[(size_t)g].push_back(t.step1_ms);
                s2[(size_t)g].push_back(t.step2_ms);
                s3[(size_t)g].push_back(t.step3_ms);
                if (g == 0) wall.push_back(ms);
            }
        }
    };
    std::vector<std::thread> th;
    for (int g = 0; g < ngpu; ++g) th.emplace_back(worker, g);
    for (auto &t : th) t.join();
    int rc = 0;
    if (failed) {
        for (int g = 0; g < ngpu; ++g)
            if (!errs[(size_t)g].empty()) fprintf(stderr, "pemspgemm: rank %d: %s\n", g, errs[(size_t)g].c_str());
        if (!errs[0].empty() && errs[0] == "inner dimensions differ") printf("inner dimensions differ. Exiting.\n");
        rc = errs[0] == "inner dimensions differ" ? 1 : 2;
    } else {
        // the pass an iteration is judged by: the slowest rank's step spans, the barrier-to-barrier wall
        size_t fidx = 0;
        if (fastest && !wall.empty()) fidx = (size_t)(std::min_element(wall.begin(), wall.end()) - wall.begin());
        auto pick = [&](const std::vector<double> &v) {
            if (v.empty()) return 0.0;
            if (fastest) return v[fidx];
            double s = 0;
            for (double x : v) s += x;
            return s / (double)v.size();
        };
        for (int g = 0; g < ngpu; ++g) {
            res.step1 = std::max(res.step1, pick(s1[(size_t)g]));
            res.step2 = std::max(res.step2, pick(s2[(size_t)g]));
            res.step3 = std::max(res.step3, pick(s3[(size_t)g]));
            pem_cplan_info ci;
            pem_cplan_get_info(plan[(size_t)g], &ci);
            res.ntiles_c += ci.ntiles_c;
            res.npairs += ci.npairs;
        }
        res.total = pick(wall);
        // the exchange step: CSR slices -> device 0 over RCCL
        int64_t nrows = 0, nnz = 0;
        if (pem_mgpu_gather_csr(m, plan.data(), 0, &nrows, &nnz, nullptr, nullptr, nullptr, nullptr) != PEM_OK) {
            fprintf(stderr, "pemspgemm: gather: %s\n", pem_last_error());
            rc = 2;
        } else {
            res.nnz_c = nnz;
            const char *ge = getenv("PEM_GATHER");
            const bool do_gather = want_c || (ge ? atoi(ge) != 0 : nnz <= 400000000ll);
            if (do_gather) {
                res.rowptr.resize((size_t)nrows + 1);
                res.colidx.resize((size_t)nnz);
                res.vals.resize((size_t)nnz);
                if (pem_mgpu_gather_csr(m, plan.data(), 0, &nrows, &nnz, res.rowptr.data(), res.colidx.data(), res.vals.data(), &res.gather_ms) != PEM_OK) {
                    fprintf(stderr, "pemspgemm: gather: %s\n", pem_last_error());
                    rc = 2;
                } else {
                    res.gathered = true;
                }
            }
        }
    }
    res.bounds = bounds;
    // the same product as ONE call with the exchange overlapped: every rank's block in `nchunks` chunks, chunk c travelling to
    // device 0 while chunk c + 1 computes (pem_mgpu_spgemm_gather_chunked; the Python harness' ChunkedRowBlock)
    if (rc == 0 && nchunks != 0 && res.nnz_c > 0) {
        int K = nchunks;
        if (K < 0) {                                         // from the bytes one rank sends: ~48 MB of CSR per chunk, 2 .. 8
            const double per_rank = 12.0 * (double)res.nnz_c / (double)ngpu;
            K = (int)std::max(2.0, std::min(8.0, std::ceil(per_rank / (48.0 * 1048576.0))));
        }
        std::vector<int32_t> cb_((size_t)ngpu * (size_t)K + 1, 0);
        std::vector<pem_cplan *> cp((size_t)ngpu * (size_t)K, nullptr);
        bool ok = pem_split_tile_rows(pem_mgpu_ctx(m, 0), A[0], B[0], ngpu * K, cb_.data()) == PEM_OK;
        for (int s = 0; ok && s < ngpu * K; ++s)
            ok = pem_cplan_create(pem_mgpu_ctx(m, s / K), A[(size_t)(s / K)], B[(size_t)(s / K)], cb_[(size_t)s], cb_[(size_t)s + 1], &cp[(size_t)s]) == PEM_OK;
        int64_t nr = 0, nz = 0;
        double pass = 0, tail = 0, sp = 0, st = 0;
        if (ok) ok = pem_mgpu_spgemm_gather_chunked(m, cp.data(), K, 0, &nr, &nz, nullptr, nullptr, nullptr, &pass, &tail) == PEM_OK;   // sizes, warm plans
        for (int n = 0; ok && n < REPEAT; ++n) {
            ok = pem_mgpu_spgemm_gather_chunked(m, cp.data(), K, 0, &nr, &nz, nullptr, nullptr, nullptr, &pass, &tail) == PEM_OK;
            sp += pass / REPEAT;
            st += tail / REPEAT;
        }
        if (ok && nz != res.nnz_c) {
            fprintf(stderr, "pemspgemm: chunked pass: %lld entries, the row-block pass had %lld\n", (long long)nz, (long long)res.nnz_c);
            ok = false;
        }
        if (ok) {
            res.chunks = K;
            res.chunk_pass_ms = sp;
            res.chunk_tail_ms = st;
        } else {
            fprintf(stderr, "pemspgemm: chunked pass: %s\n", pem_last_error());
            rc = 2;
        }
        for (int s = 0; s < ngpu * K; ++s)
            if (cp[(size_t)s]) pem_cplan_destroy(pem_mgpu_ctx(m, s / K), cp[(size_t)s]);
    }
    for (int g = 0; g < ngpu; ++g) {
        pem_ctx *ctx = pem_mgpu_ctx(m, g);
        if (plan[(size_t)g]) pem_cplan_destroy(ctx, plan[(size_t)g]);
        if (B[(size_t)g] && B[(size_t)g] != A[(size_t)g]) pem_tiled_destroy(ctx, B[(size_t)g]);
        if (A[(size_t)g]) pem_tiled_destroy(ctx, A[(size_t)g]);
    }
    pem_mgpu_destroy(m);
    return rc;
}

// the input: a Matrix-Market file (spgemm.cu:43-110), or "standin:NAME" for the seeded generator (--standin)
static double g_standin_scale = 1.0;
static int read_input(const char *path, pem_coo *out)
{
    if (!strncmp(path, "standin:", 8)) {
        const int rc = pem_standin_generate(path + 8, g_standin_scale, out);
        if (rc != 0) fprintf(stderr, "pemspgemm: --standin %s (scale %g): no such stand-in (have: %s) or bad scale\n", path + 8, g_standin_scale, pem_standin_names());
        return rc;
    }
    const int rc = pem_mm_read(path, 0, out);
    if (rc != 0) fprintf(stderr, "pemspgemm: %s\n", pem_host_last_error());
    return rc;
}

int main(int argc, char *argv[])
{
    // Beyond the reference: `--B <file.mtx>` multiplies by a second matrix instead of A itself,
    // `--out <file.mtx>` writes C as a Matrix-Market file.  They are stripped before the reference's own grammar.
    // `--cache <dir>` keeps each tiling as <dir>/<stem>.<A|AT>.pemtile and, while the .mtx is unchanged
    // (size + mtime), loads it instead of parsing and converting.
    // `--fp32` computes in float: the values are rounded once at conversion, step 3 runs one fmaf per product.
    // `--gpus N`: N tile-row blocks of A on N devices, C gathered to device 0 over RCCL (see run_multi); the blocks are
    // re-cut from measured pass times first (`--no-tune-split` keeps the product-balanced cut), and the product is also run as
    // `--chunks K` chunks per rank with the gather overlapped (default: K from C's size; 0: off).
    // `--standin NAME [--scale S]`: the seeded C++ stand-in generator takes the place of the file (no SuiteSparse
    // file exists offline); NAME then stands where the path stood, and the remaining positional arguments keep their meaning.
    const char *b_path = nullptr, *out_path = nullptr, *cache_dir = nullptr, *standin = nullptr;
    static std::string standin_path;
    static std::vector<char *> argv_store;
    bool fp32 = false, tune_split = true;
    int ngpu = 0;   // 0: the reference's single-device path
    int nchunks = -1;   // --gpus: chunks per rank of the overlapped pass (-1: from C's size, 0: off)
    {
        int w = 1;
        for (int r = 1; r < argc; ++r) {
            if (!strcmp(argv[r], "--B") && r + 1 < argc) b_path = argv[++r];
            else if (!strcmp(argv[r], "--standin") && r + 1 < argc) standin = argv[++r];
            else if (!strcmp(argv[r], "--scale") && r + 1 < argc) g_standin_scale = atof(argv[++r]);
            else if (!strcmp(argv[r], "--gpus") && r + 1 < argc) ngpu = atoi(argv[++r]);
            else if (!strcmp(argv[r], "--out") && r + 1 < argc) out_path = argv[++r];
            else if (!strcmp(argv[r], "--cache") && r + 1 < argc) cache_dir = argv[++r];
            else if (!strcmp(argv[r], "--fp32")) fp32 = true;
            else if (!strcmp(argv[r], "--no-tune-split")) tune_split = false;
            else if (!strcmp(argv[r], "--chunks") && r + 1 < argc) nchunks = atoi(argv[++r]);
            else argv[w++] = argv[r];
        }
        argc = w;
        if (standin) {
            standin_path = std::string("standin:") + standin;
            argv_store.assign(argv, argv + argc);
            argv_store.insert(argv_store.begin() + 1, const_cast<char *>(standin_path.c_str()));
            argv = argv_store.data();
            ++argc;
        }
    }
    if (argc <= 1 || argc > 4) {   // spgemm.cu:722-725
        printf("Provide a matrix market file path. Exiting.\n");
        return 1;
    }
    const int WARMUP = env_int("PEM_WARMUP", 1);    // spgemm.cu:712-714
    const int REPEAT = env_int("PEM_REPEAT", 10);   // reference Makefile:34 (-DREPEAT=10)
    const bool fastest = env_int("PEM_FASTEST", 0) != 0;   // spgemm.cu:1359-1363
    const bool save = argc >= 3 && atoi(argv[2]) != 0;     // spgemm.cu:1485 (the reference dereferences argv[2] unconditionally)
    const bool aat = argc == 4;                            // spgemm.cu:788: presence of a 3rd argument, value ignored

    if (ngpu > 0) {
        if (fp32 || cache_dir) {
            printf("--gpus does not combine with --fp32 / --cache. Exiting.\n");
            return 1;
        }
        auto conv_start_m = std::chrono::high_resolution_clock::now();
        pem_coo ca, cb;
        memset(&ca, 0, sizeof ca);
        memset(&cb, 0, sizeof cb);
        if (read_input(argv[1], &ca) != 0 || (b_path && read_input(b_path, &cb) != 0)) return 1;
        if (!b_path && !aat && ca.rows != ca.cols) {
            printf("input is rectangular. Only AAt is possible. Exiting.\n");   // spgemm.cu:782-786
            return 1;
        }
        printf("MATRIX A\nfilepath: %s\nRows: %d\nCols: %d\nNnz: %lld\n", argv[1], ca.rows, ca.cols, (long long)ca.nnz);
        MultiResult mr;
        int rc = run_multi(ngpu, ca, b_path ? &cb : nullptr, aat, save || out_path, WARMUP, REPEAT, fastest, conv_start_m, mr, tune_split, nchunks);
        if (rc == 0) {
            const double kernel = mr.step1 + mr.step2 + mr.step3, malloc_ms = mr.total - kernel;
            const double gflops = mr.total > 0 ? (double)mr.flop * 2.0 / (mr.total * 1e6) : 0.0;
            const double gflops_x = mr.total + mr.gather_ms > 0 ? (double)mr.flop * 2.0 / ((mr.total + mr.gather_ms) * 1e6) : 0.0;
            const double ratio = mr.nnz_c ? (double)mr.flop / (double)mr.nnz_c : 0.0;
            printf("\n%d GPUs, row blocks of A balanced on tile-level products, B replicated\nwarm up %d time\naverage over %d iterations\n\n", ngpu, WARMUP, REPEAT);
            printf("<---Program done--->\n");
            printf("total conversion overhead----------------%.2fms\n\n", mr.conv_ms);
            printf("step1 - High Level Multiplication took---%.2fms\n", mr.step1);
            printf("step2 - Allocating C took----------------%.2fms\n", mr.step2);
            printf("step3 - Accumulation took----------------%.2fms\n\n", mr.step3);
            printf("pemSpGEMM took %.2fms ----- GFlops: %.2f\nKernel time %.2fms\nmalloc time %.2fms\n", mr.total, gflops, kernel, malloc_ms);
            printf("first pass (allocations + size read-backs) %.2fms\n", mr.cold);
            if (mr.gathered) printf("gather of C to GPU 0 over RCCL took %.2fms ----- GFlops with it: %.2f\n", mr.gather_ms, gflops_x);
            if (mr.tune_rounds > 0)
                printf("row split re-cut from measured pass times: %d round(s), slowest rank %.3fms -> %.3fms\n", mr.tune_rounds, mr.tune_first_max,
                       mr.tune_best_max);
            printf("tile-row blocks:");
            for (size_t g = 0; g + 1 < mr.bounds.size(); ++g) printf(" [%d,%d)", mr.bounds[g], mr.bounds[g + 1]);
            printf("\n");
            if (mr.chunks > 0)
                printf("%d chunks per rank, gather overlapped: compute %.2fms + transfers still in flight %.2fms ----- GFlops with the exchange: %.2f\n",
                       mr.chunks, mr.chunk_pass_ms, mr.chunk_tail_ms,
                       mr.chunk_pass_ms + mr.chunk_tail_ms > 0 ? (double)mr.flop * 2.0 / ((mr.chunk_pass_ms + mr.chunk_tail_ms) * 1e6) : 0.0);
            printf("Flop count: %llu\n\nC tiles: %lld\nC nnz: %lld\nCompression ratio %.2f\n", (unsigned long long)mr.flop, (long long)mr.ntiles_c,
                   (long long)mr.nnz_c, ratio);
            const double b_alg = 12.0 * ((double)ca.nnz + (double)(b_path ? cb.nnz : ca.nnz) + (double)mr.nnz_c) + 4.0 * 3.0 * ((double)ca.rows + 1);
            const double frac_kernel = kernel > 0 ? b_alg / (kernel * 1e-3) / (8.0e12 * ngpu) : 0.0;
            const double frac_total = mr.total > 0 ? b_alg / (mr.total * 1e-3) / (8.0e12 * ngpu) : 0.0;
            std::string path = argv[1];
            size_t slash = path.find_last_of('/');
            std::string stem = slash == std::string::npos ? path : path.substr(slash + 1);
            if (stem.size() > 4 && stem.compare(stem.size() - 4, 4, ".mtx") == 0) stem.resize(stem.size() - 4);
            pem_csv_record rec = {stem.c_str(), mr.flop, mr.nnz_c, ratio, mr.a_conv_kernel_ms, mr.b_conv_kernel_ms, mr.conv_ms,
                                  mr.step1, mr.step2, mr.step3, mr.total, kernel, malloc_ms, gflops};
            char extra[320];
            // columns 15..: gpus, B_alg, roofline fraction (kernel, total), C tiles, live pairs, first-pass ms, gather ms (sequential),
            // chunks per rank, chunked pass ms, transfers still in flight after it ms, split-tuning rounds, slowest rank before / after (ms)
            snprintf(extra, sizeof extra, "%d,%.0f,%.4f,%.4f,%lld,%lld,%.2f,%.2f,%d,%.2f,%.2f,%d,%.3f,%.3f", ngpu, b_alg, frac_kernel, frac_total,
                     (long long)mr.ntiles_c, (long long)mr.npairs, mr.cold, mr.gather_ms, mr.chunks, mr.chunk_pass_ms, mr.chunk_tail_ms, mr.tune_rounds,
                     mr.tune_first_max, mr.tune_best_max);
            const char *csv = getenv("PEM_CSV") ? getenv("PEM_CSV") : "./pemspgemm_benchmark_result.csv";
            if (pem_csv_append(csv, &rec, extra) != 0) fprintf(stderr, "pemspgemm: %s\n", pem_host_last_error());
            if (!save) {
                printf("Not saving results. Exiting.\n");
            } else {
                const char *dir = getenv("PEM_RESULT_DIR") ? getenv("PEM_RESULT_DIR") : "/tmp";
                std::vector<int32_t> rows((size_t)mr.nnz_c);
                for (int64_t r = 0; r < mr.rows; ++r)   // sorted (row, col) order = CSR order (spgemm.cu:1516-1519)
                    for (int32_t e = mr.rowptr[(size_t)r]; e < mr.rowptr[(size_t)r + 1]; ++e) rows[(size_t)e] = (int32_t)r;
                printf("Saving results to %s/SPGEMM_RESULT_*.txt\n", dir);
                if (pem_write_result_files(dir, mr.nnz_c, rows.data(), mr.colidx.data(), mr.vals.data()) != 0) {
                    fprintf(stderr, "pemspgemm: %s\n", pem_host_last_error());
                    rc = 2;
                }
            }
            if (out_path) {
                if (pem_write_mtx_csr(out_path, (int32_t)mr.rows, mr.cols_b, mr.rowptr.data(), mr.colidx.data(), mr.vals.data(),
                                      "C = A*B by pemspgemm (pem-spgemm_amd), row blocks gathered over RCCL") != 0) {
                    fprintf(stderr, "pemspgemm: %s\n", pem_host_last_error());
                    rc = 2;
                } else {
                    printf("C written to %s\n", out_path);
                }
            }
        }
        printf("CLEANING UP RESOURCES\n\n");
        pem_coo_free(&ca);
        pem_coo_free(&cb);
        return rc;
    }
    pem_ctx *ctx = nullptr;   // device + memory set-up precede the conversion clock, as in the reference (spgemm.cu:730-758)
    CHECK(pem_ctx_create(env_int("PEM_DEVICE", 0), &ctx));
    auto conv_start = std::chrono::high_resolution_clock::now();   // spgemm.cu:760: the clock starts before the file is read
    // One tiling = (source file, transposed?).  With --cache it is loaded from <dir>/<stem>.<A|AT>.pemtile when that
    // file was made from the same .mtx (size + mtime), else parsed + converted and then saved there.
    struct Source {
        const char *path;
        pem_coo coo;
        bool read = false;
    };
    Source srcA = {argv[1], {}, false}, srcB = {b_path, {}, false};
    memset(&srcA.coo, 0, sizeof srcA.coo);
    memset(&srcB.coo, 0, sizeof srcB.coo);
    int cache_hits = 0, cache_misses = 0;
    auto make_tiling = [&](Source &src, int transpose, pem_tiled **out) -> int {
        std::string cpath;
        pem_cache_key key;
        memset(&key, 0, sizeof key);
        bool keyed = false;
        if (cache_dir) {
            struct stat st;
            if (stat(src.path, &st) == 0) {
                key.source_size = (uint64_t)st.st_size;
                key.source_mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
                key.transpose = (uint32_t)transpose;
                keyed = true;
                std::string p = src.path;
                size_t sl = p.find_last_of('/');
                std::string stem = sl == std::string::npos ? p : p.substr(sl + 1);
                cpath = std::string(cache_dir) + "/" + stem + (transpose ? ".AT" : ".A") + (fp32 ? ".f32" : "") + ".pemtile";
                pem_status ls = pem_tiled_load(ctx, cpath.c_str(), &key, out);
                if (ls == PEM_OK) {
                    pem_tiled_info li;
                    if (pem_tiled_get_info(*out, &li) == PEM_OK && li.value_bytes == (fp32 ? 4 : 8)) {
                        ++cache_hits;
                        return 0;
                    }
                    pem_tiled_destroy(ctx, *out);   // a file of the other value type under this name: rebuild
                    *out = nullptr;
                    ls = PEM_E_STALE;
                }
                if (ls != PEM_E_IO && ls != PEM_E_STALE) {
                    fprintf(stderr, "pemspgemm: %s\n", pem_last_error());
                    return 2;
                }
                ++cache_misses;   // absent, stale or damaged: rebuild below and overwrite
            }
        }
        if (!src.read) {
            if (read_input(src.path, &src.coo) != 0) return 1;
            src.read = true;
        }
        const pem_coo &m = src.coo;
        pem_status cs;
        if (fp32) {
            std::vector<float> vf((size_t)m.nnz);
            for (int64_t e = 0; e < m.nnz; ++e) vf[(size_t)e] = (float)m.V[e];
            cs = pem_tiled_from_coo_f32(ctx, m.rows, m.cols, m.nnz, m.I, m.J, vf.data(), transpose, out);
        } else {
            cs = pem_tiled_from_coo(ctx, m.rows, m.cols, m.nnz, m.I, m.J, m.V, transpose, out);
        }
        if (cs != PEM_OK) {
            fprintf(stderr, "pemspgemm: conversion of %s failed: %s\n", src.path, pem_last_error());
            return 2;
        }
        if (keyed && pem_tiled_save(ctx, *out, cpath.c_str(), &key) != PEM_OK)
            fprintf(stderr, "pemspgemm: cache not written: %s\n", pem_last_error());   // not fatal
        return 0;
    };
    pem_tiled *A = nullptr, *B = nullptr;
    if (int rc = make_tiling(srcA, 0, &A)) return rc;
    if (b_path) {
        if (int rc = make_tiling(srcB, aat ? 1 : 0, &B)) return rc;
    } else if (aat) {
        if (int rc = make_tiling(srcA, 1, &B)) return rc;
    } else {
        B = A;   // the reference converts the same file twice (spgemm.cu:778-779); one tiling serves both roles here
    }
    pem_tiled_info ia, ib;
    CHECK(pem_tiled_get_info(A, &ia));
    CHECK(pem_tiled_get_info(B, &ib));
    if (ia.cols != ib.rows) {
        if (b_path)
            printf("inner dimensions differ: A is %d x %d, B%s is %d x %d. Exiting.\n", ia.rows, ia.cols, aat ? "^T" : "", ib.rows, ib.cols);
        else
            printf("input is rectangular. Only AAt is possible. Exiting.\n");   // spgemm.cu:782-786
        return 1;
    }
    if (fp32) printf("value type: fp32 (the reference computes in fp64)\n");
    printf("MATRIX A\nfilepath: %s\nRows: %d\nCols: %d\nNnz: %lld\n", argv[1], ia.rows, ia.cols, (long long)ia.nnz);   // spgemm.cu:794-806
    printf("MATRIX B\nfilepath: %s\nRows: %d\nCols: %d\nNnz: %lld\n", b_path ? b_path : argv[1], ib.rows, ib.cols, (long long)ib.nnz);
    if (cache_dir) printf("tiled-format cache %s: %d loaded, %d rebuilt\n", cache_dir, cache_hits, cache_misses);
    const double conv_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - conv_start).count();
    uint64_t flop = 0;
    CHECK(pem_flop_count(ctx, A, B, &flop));   // spgemm.cu:1068-1079

    pem_cplan *plan = nullptr;
    CHECK(pem_cplan_create(ctx, A, B, 0, -1, &plan));
    std::vector<double> s1, s2, s3, wall;
    double cold_ms = 0.0;   // the first pass on the plan: every allocation + the three size read-backs (what the reference pays per iteration)
    for (int n = 0; n < WARMUP + REPEAT; ++n) {   // spgemm.cu:1133-1357
        if (n == 0) printf("\nstep1 pemSpGEMM (row-local tile-level expand + sort)\n\nstep2 pemSpGEMM\n\nstep3 pemSpGEMM\n\n\n");
        CHECK(pem_spgemm(ctx, plan));
        pem_timings t;
        CHECK(pem_get_timings(ctx, &t));
        if (n == 0) cold_ms = t.spgemm_wall_ms;
        if (n >= WARMUP) {
            s1.push_back(t.step1_ms);
            s2.push_back(t.step2_ms);
            s3.push_back(t.step3_ms);
            wall.push_back(t.spgemm_wall_ms);
        }
    }
    auto pick = [&](const std::vector<double> &v, size_t idx) {
        if (v.empty()) return 0.0;
        if (fastest) return v[idx];
        double s = 0;
        for (double x : v) s += x;
        return s / (double)v.size();
    };
    size_t fidx = 0;
    if (fastest && !wall.empty()) {
        fidx = (size_t)(std::min_element(wall.begin(), wall.end()) - wall.begin());
        printf("fidx: %zu\n\n", fidx);
    } else {
        printf("warm up %d time\naverage over %d iterations\n\n", WARMUP, REPEAT);
    }
    const double step1 = pick(s1, fidx), step2 = pick(s2, fidx), step3 = pick(s3, fidx), total = pick(wall, fidx);
    const double kernel = step1 + step2 + step3, malloc_ms = total - kernel;   // spgemm.cu:1353-1354
    pem_cplan_info ci;
    CHECK(pem_cplan_get_info(plan, &ci));
    const double gflops = total > 0 ? (double)flop * 2.0 / (total * 1e6) : 0.0;   // spgemm.cu:1403
    const double ratio = ci.nnz_c ? (double)flop / (double)ci.nnz_c : 0.0;       // spgemm.cu:1404

    printf("<---Program done--->\n");   // spgemm.cu:1406-1422
    printf("Matrix A CSR to tile kernel took---------%.2fms\n", ia.conv_tile_kernel_ms);
    printf("Matrix B CSR to tile kernel took---------%.2fms\n", ib.conv_tile_kernel_ms);
    printf("total conversion overhead----------------%.2fms\n\n", conv_ms);
    printf("step1 - High Level Multiplication took---%.2fms\n", step1);
    printf("step2 - Allocating C took----------------%.2fms\n", step2);
    printf("step3 - Accumulation took----------------%.2fms\n\n", step3);
    printf("pemSpGEMM took %.2fms ----- GFlops: %.2f\nKernel time %.2fms\nmalloc time %.2fms\n", total, gflops, kernel, malloc_ms);
    printf("first pass (allocations + size read-backs) %.2fms; the passes above re-use the plan's buffers and sizes\n", cold_ms);
    printf("Flop count: %llu\n\n", (unsigned long long)flop);
    printf("C tiles: %lld\n", (long long)ci.ntiles_c);
    printf("C nnz: %lld\n", (long long)ci.nnz_c);
    printf("Compression ratio %.2f\n", ratio);
    // roofline bookkeeping
    const double b_alg = 12.0 * ((double)ia.nnz + (double)ib.nnz + (double)ci.nnz_c) + 4.0 * ((double)ia.rows + 1) + 4.0 * ((double)ib.rows + 1) +
                         4.0 * ((double)ia.rows + 1);
    const double frac_kernel = kernel > 0 ? b_alg / (kernel * 1e-3) / 8.0e12 : 0.0, frac_total = total > 0 ? b_alg / (total * 1e-3) / 8.0e12 : 0.0;
    printf("B_alg %.0f bytes; HBM roofline fraction (8 TB/s): kernel %.4f, total %.4f\n", b_alg, frac_kernel, frac_total);

    // CSV (spgemm.cu:1424-1450); matrix name = file stem (the reference's regex needs a '/' in the path)
    std::string path = argv[1];
    size_t slash = path.find_last_of('/');
    std::string stem = slash == std::string::npos ? path : path.substr(slash + 1);
    if (stem.size() > 4 && stem.compare(stem.size() - 4, 4, ".mtx") == 0) stem.resize(stem.size() - 4);
    pem_csv_record rec = {stem.c_str(), flop, ci.nnz_c, ratio, ia.conv_tile_kernel_ms, ib.conv_tile_kernel_ms, conv_ms,
                          step1, step2, step3, total, kernel, malloc_ms, gflops};
    char extra[256];
    // columns 15..: gpus, B_alg, roofline fraction (kernel, total), C tiles, live pairs, first-pass ms, gather ms
    snprintf(extra, sizeof extra, "1,%.0f,%.4f,%.4f,%lld,%lld,%.2f,0.00", b_alg, frac_kernel, frac_total, (long long)ci.ntiles_c, (long long)ci.npairs,
             cold_ms);
    const char *csv = getenv("PEM_CSV") ? getenv("PEM_CSV") : "./pemspgemm_benchmark_result.csv";
    if (pem_csv_append(csv, &rec, extra) != 0) fprintf(stderr, "pemspgemm: %s\n", pem_host_last_error());

    int rc = 0;
    if (!save) {
        printf("Not saving results. Exiting.\n");   // spgemm.cu:1487
    } else {
        const char *dir = getenv("PEM_RESULT_DIR") ? getenv("PEM_RESULT_DIR") : "/tmp";
        std::vector<int32_t> rows((size_t)ci.nnz_c), cols((size_t)ci.nnz_c);
        std::vector<double> vals((size_t)ci.nnz_c);
        int64_t nnz = 0;
        if (fp32) {
            std::vector<float> vf((size_t)ci.nnz_c);
            CHECK(pem_c_export_coo_f32(ctx, plan, &nnz, rows.data(), cols.data(), vf.data()));
            for (size_t e = 0; e < vf.size(); ++e) vals[e] = vf[e];
        } else {
            CHECK(pem_c_export_coo(ctx, plan, &nnz, rows.data(), cols.data(), vals.data()));   // spgemm.cu:1493-1543
        }
        pem_timings t;
        CHECK(pem_get_timings(ctx, &t));
        printf("sanitize_C took %.2fms\n", t.export_ms);
        printf("Saving results to %s/SPGEMM_RESULT_*.txt\n", dir);
        if (pem_write_result_files(dir, nnz, rows.data(), cols.data(), vals.data()) != 0) {
            fprintf(stderr, "pemspgemm: %s\n", pem_host_last_error());
            rc = 2;
        }
    }
    if (out_path) {   // C as a Matrix-Market file (CSR export: rows ascending, columns ascending)
        std::vector<int32_t> rp((size_t)(ci.row_end - ci.row_begin) + 1), cidx((size_t)ci.nnz_c);
        std::vector<double> cv((size_t)ci.nnz_c);
        int64_t nnz = 0;
        if (fp32) {
            std::vector<float> vf((size_t)ci.nnz_c);
            CHECK(pem_c_export_csr_f32(ctx, plan, &nnz, rp.data(), cidx.data(), vf.data()));
            for (size_t e = 0; e < vf.size(); ++e) cv[e] = vf[e];
        } else {
            CHECK(pem_c_export_csr(ctx, plan, &nnz, rp.data(), cidx.data(), cv.data()));
        }
        if (pem_write_mtx_csr(out_path, ia.rows, ib.cols, rp.data(), cidx.data(), cv.data(), "C = A*B by pemspgemm (pem-spgemm_amd)") != 0) {
            fprintf(stderr, "pemspgemm: %s\n", pem_host_last_error());
            rc = 2;
        } else {
            printf("C written to %s\n", out_path);
        }
    }
    printf("CLEANING UP RESOURCES\n\n");
    pem_cplan_destroy(ctx, plan);
    if (B != A) pem_tiled_destroy(ctx, B);
    pem_tiled_destroy(ctx, A);
    pem_ctx_destroy(ctx);
    pem_coo_free(&srcA.coo);
    pem_coo_free(&srcB.coo);
    return rc;
}

// pemspgemm -- command-line front end with the reference's surface (spgemm.cu:720-1568):
//     pemspgemm <matrix.mtx> [0|1] [any third argument => C = A*A^T]
// reads one Matrix-Market file, forms C = A*A (or A*A^T), times WARMUP + REPEAT passes of
// step1+step2+step3, prints the reference's report, appends the reference's 14 CSV columns
// to ./pemspgemm_benchmark_result.csv and, when the 2nd argument is non-zero, writes the
// sorted COO result to /tmp/SPGEMM_RESULT_{NNZ,ROWS,COLS,VALS}.txt.
// Host C++ over the C ABI of libpemspgemm_hip.so; no HIP calls in this file.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <vector>
#include "../../include/pem_host.h"
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

int main(int argc, char *argv[])
{
    // Beyond the reference: `--B <file.mtx>` multiplies by a second matrix instead of A itself,
    // `--out <file.mtx>` writes C as a Matrix-Market file.  They are stripped before the reference's own grammar.
    // `--cache <dir>` keeps each tiling as <dir>/<stem>.<A|AT>.pemtile and, while the .mtx is unchanged
    // (size + mtime), loads it instead of parsing and converting.
    // `--fp32` computes in float: the values are rounded once at conversion, step 3 runs one fmaf per product.
    const char *b_path = nullptr, *out_path = nullptr, *cache_dir = nullptr;
    bool fp32 = false;
    {
        int w = 1;
        for (int r = 1; r < argc; ++r) {
            if (!strcmp(argv[r], "--B") && r + 1 < argc) b_path = argv[++r];
            else if (!strcmp(argv[r], "--out") && r + 1 < argc) out_path = argv[++r];
            else if (!strcmp(argv[r], "--cache") && r + 1 < argc) cache_dir = argv[++r];
            else if (!strcmp(argv[r], "--fp32")) fp32 = true;
            else argv[w++] = argv[r];
        }
        argc = w;
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
            if (pem_mm_read(src.path, 0, &src.coo) != 0) {
                fprintf(stderr, "pemspgemm: %s\n", pem_host_last_error());
                return 1;
            }
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
    for (int n = 0; n < WARMUP + REPEAT; ++n) {   // spgemm.cu:1133-1357
        if (n == 0) printf("\nstep1 pemSpGEMM (tile-level expand + radix sort)\n\nstep2 pemSpGEMM\n\nstep3 pemSpGEMM\n\n\n");
        CHECK(pem_spgemm(ctx, plan));
        pem_timings t;
        CHECK(pem_get_timings(ctx, &t));
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
    snprintf(extra, sizeof extra, "1,%.0f,%.4f,%.4f,%lld,%lld", b_alg, frac_kernel, frac_total, (long long)ci.ntiles_c, (long long)ci.npairs);
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

// mmio.cpp -- Matrix-Market coordinate reader + result/CSV writers of the pemspgemm CLI.
// Replaces read_matrix_market<T> (spgemm.cu:43-110; fast_matrix_market is not available) with
// an mmap + multi-threaded chunk parser.  Semantics (pattern -> 1, complex -> real part,
// symmetric / skew / hermitian generalised, diagonal once) are documented in DESIGN.md.
#include "../../include/pem_host.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <charconv>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *pem_host_last_error(void) { return g_err; }

namespace {

struct Chunk {
    std::vector<int32_t> I, J;
    std::vector<double> V;
    bool bad = false;
};

inline const char *skip_ws(const char *p, const char *e)
{
    while (p < e && (*p == ' ' || *p == '\t')) ++p;
    return p;
}

// parse the lines of [b, e) (b at a line start); field: 0 real 1 integer 2 pattern 3 complex
void parse_chunk(const char *b, const char *e, int field, long long rows, long long cols, Chunk *out)
{
    const char *p = b;
    while (p < e) {
        const char *eol = static_cast<const char *>(memchr(p, '\n', (size_t)(e - p)));
        if (!eol) eol = e;
        const char *q = skip_ws(p, eol);
        if (q < eol && *q != '%' && *q != '\r') {
            long long i = 0, j = 0;
            auto r1 = std::from_chars(q, eol, i);
            if (r1.ec != std::errc()) { out->bad = true; return; }
            q = skip_ws(r1.ptr, eol);
            auto r2 = std::from_chars(q, eol, j);
            if (r2.ec != std::errc()) { out->bad = true; return; }
            double v = 1.0;
            if (field != 2) {
                q = skip_ws(r2.ptr, eol);
                if (q < eol && *q == '+') ++q;
                auto r3 = std::from_chars(q, eol, v);
                if (r3.ec != std::errc()) { out->bad = true; return; }
            }
            if (i < 1 || j < 1 || i > rows || j > cols) { out->bad = true; return; }
            out->I.push_back((int32_t)(i - 1));
            out->J.push_back((int32_t)(j - 1));
            out->V.push_back(v);
        }
        p = eol + 1;
    }
}

std::string lower(std::string s)
{
    for (auto &c : s) c = (char)tolower((unsigned char)c);
    return s;
}

}  // namespace

extern "C" int pem_mm_read(const char *path, int threads, pem_coo *out)
{
    if (!path || !out) { set_err("pem_mm_read: null argument"); return -1; }
    memset(out, 0, sizeof *out);
    int fd = open(path, O_RDONLY);
    if (fd < 0) { set_err("cannot open %s: %s", path, strerror(errno)); return -1; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size == 0) { close(fd); set_err("%s: empty or unreadable", path); return -2; }
    const size_t size = (size_t)st.st_size;
    const char *data = static_cast<const char *>(mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0));
    close(fd);
    if (data == MAP_FAILED) { set_err("mmap(%s) failed: %s", path, strerror(errno)); return -1; }
    const char *end = data + size;
    auto fail = [&](int code, const char *msg) { munmap(const_cast<char *>(data), size); set_err("%s: %s", path, msg); return code; };

    // banner
    const char *eol = static_cast<const char *>(memchr(data, '\n', size));
    if (!eol) return fail(-2, "no header line");
    char banner[64], object[64], format[64], field_s[64], symm_s[64];
    {
        std::string line(data, eol);
        if (sscanf(line.c_str(), "%63s %63s %63s %63s %63s", banner, object, format, field_s, symm_s) != 5) return fail(-2, "malformed banner");
    }
    if (lower(banner) != "%%matrixmarket" || lower(object) != "matrix") return fail(-2, "not a MatrixMarket matrix file");
    if (lower(format) != "coordinate") return fail(-3, "only coordinate files are supported (dense 'array' files are out of scope)");
    const std::string fs = lower(field_s), ss = lower(symm_s);
    const int field = fs == "real" || fs == "double" ? 0 : fs == "integer" ? 1 : fs == "pattern" ? 2 : fs == "complex" ? 3 : -1;
    const int sym = ss == "general" ? 0 : ss == "symmetric" ? 1 : ss == "skew-symmetric" ? 2 : ss == "hermitian" ? 3 : -1;
    if (field < 0 || sym < 0) return fail(-2, "unknown field or symmetry in the banner");
    // comments, blank lines, size line
    const char *p = eol + 1;
    long long rows = 0, cols = 0, n = 0;
    for (;;) {
        if (p >= end) return fail(-2, "missing size line");
        eol = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        if (!eol) eol = end;
        const char *q = skip_ws(p, eol);
        if (q < eol && *q != '%' && *q != '\r') {
            std::string line(q, eol);
            if (sscanf(line.c_str(), "%lld %lld %lld", &rows, &cols, &n) != 3) return fail(-2, "malformed size line");
            p = eol < end ? eol + 1 : end;
            break;
        }
        p = eol + 1;
    }
    if (rows <= 0 || cols <= 0 || n < 0 || rows > 0x7FFFFFFFll || cols > 0x7FFFFFFFll) return fail(-2, "bad dimensions");

    // chunked parallel parse of the entry lines
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    const size_t body = (size_t)(end - p);
    if (body < (size_t)(1 << 20)) nt = 1;
    if (nt > 64) nt = 64;
    std::vector<const char *> cut((size_t)nt + 1);
    cut[0] = p;
    cut[(size_t)nt] = end;
    for (int t = 1; t < nt; ++t) {
        const char *c = p + body * (size_t)t / (size_t)nt;
        const char *nl = static_cast<const char *>(memchr(c, '\n', (size_t)(end - c)));
        cut[(size_t)t] = nl ? nl + 1 : end;
    }
    std::vector<Chunk> chunks((size_t)nt);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t)
        pool.emplace_back(parse_chunk, cut[(size_t)t], cut[(size_t)t + 1], field, rows, cols, &chunks[(size_t)t]);
    parse_chunk(cut[0], cut[1], field, rows, cols, &chunks[0]);
    for (auto &th : pool) th.join();
    long long got = 0;
    for (auto &c : chunks) {
        if (c.bad) return fail(-4, "malformed or out-of-range entry line");
        got += (long long)c.I.size();
    }
    if (got != n) return fail(-5, "entry count differs from the size line");
    munmap(const_cast<char *>(data), size);

    const size_t cap = (size_t)(sym ? 2 * n : n);
    int32_t *I = static_cast<int32_t *>(malloc(sizeof(int32_t) * (cap ? cap : 1)));
    int32_t *J = static_cast<int32_t *>(malloc(sizeof(int32_t) * (cap ? cap : 1)));
    double *V = static_cast<double *>(malloc(sizeof(double) * (cap ? cap : 1)));
    if (!I || !J || !V) { free(I); free(J); free(V); set_err("%s: out of host memory", path); return -6; }
    size_t cnt = 0;
    for (auto &c : chunks) {
        memcpy(I + cnt, c.I.data(), sizeof(int32_t) * c.I.size());
        memcpy(J + cnt, c.J.data(), sizeof(int32_t) * c.J.size());
        memcpy(V + cnt, c.V.data(), sizeof(double) * c.V.size());
        cnt += c.I.size();
    }
    if (sym) {
        const size_t base = cnt;
        for (size_t e = 0; e < base; ++e) {
            if (I[e] == J[e]) continue;       // diagonal once (deviation from fmm's extra-zero default)
            I[cnt] = J[e];
            J[cnt] = I[e];
            V[cnt] = sym == 2 ? -V[e] : V[e];
            ++cnt;
        }
    }
    out->rows = (int32_t)rows;
    out->cols = (int32_t)cols;
    out->nnz = (int64_t)cnt;
    out->I = I;
    out->J = J;
    out->V = V;
    out->symmetric = sym != 0;
    out->field = field;
    return 0;
}

extern "C" void pem_coo_free(pem_coo *m)
{
    if (!m) return;
    free(m->I);
    free(m->J);
    free(m->V);
    memset(m, 0, sizeof *m);
}

extern "C" int pem_write_result_files(const char *dir, int64_t nnz, const int32_t *rows, const int32_t *cols, const double *vals)
{
    const std::string d = dir && *dir ? dir : "/tmp";
    auto open_w = [&](const char *name) { return fopen((d + "/" + name).c_str(), "w"); };
    FILE *f = open_w("SPGEMM_RESULT_NNZ.txt");
    if (!f) { set_err("cannot write result files under %s", d.c_str()); return -1; }
    fprintf(f, "%lld", (long long)nnz);                               // spgemm.cu:1546 (no newline)
    fclose(f);
    std::vector<char> buf(1 << 20);
    auto dump_int = [&](const char *name, const int32_t *a) {
        FILE *g = open_w(name);
        if (!g) return -1;
        setvbuf(g, buf.data(), _IOFBF, buf.size());
        for (int64_t e = 0; e < nnz; ++e) fprintf(g, "%d\n", a[e]);  // spgemm.cu:1535, 1550, 1554
        fclose(g);
        return 0;
    };
    if (dump_int("SPGEMM_RESULT_ROWS.txt", rows) || dump_int("SPGEMM_RESULT_COLS.txt", cols)) { set_err("cannot write result files under %s", d.c_str()); return -1; }
    FILE *g = open_w("SPGEMM_RESULT_VALS.txt");
    if (!g) { set_err("cannot write result files under %s", d.c_str()); return -1; }
    setvbuf(g, buf.data(), _IOFBF, buf.size());
    for (int64_t e = 0; e < nnz; ++e) fprintf(g, "%.17f\n", vals[e]);  // std::fixed, precision max_digits10 = 17 (spgemm.cu:1558)
    fclose(g);
    return 0;
}

extern "C" int pem_write_mtx_csr(const char *path, int32_t rows, int32_t cols, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                 const char *comment)
{
    if (!path || !rowptr || rows < 0 || cols < 0) { set_err("pem_write_mtx_csr: bad arguments"); return -1; }
    FILE *f = fopen(path, "w");
    if (!f) { set_err("cannot write %s: %s", path, strerror(errno)); return -1; }
    std::vector<char> buf(1 << 22);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n");
    if (comment && *comment) fprintf(f, "%% %s\n", comment);
    fprintf(f, "%d %d %lld\n", rows, cols, (long long)rowptr[rows]);
    char num[40];
    for (int32_t r = 0; r < rows; ++r)
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            auto res = std::to_chars(num, num + sizeof num, vals[e], std::chars_format::general, 17);
            *res.ptr = 0;
            fprintf(f, "%d %d %s\n", r + 1, colidx[e] + 1, num);
        }
    if (fclose(f) != 0) { set_err("write error on %s", path); return -1; }
    return 0;
}

extern "C" int pem_csv_append(const char *path, const pem_csv_record *r, const char *extra)
{
    FILE *f = fopen(path, "a");
    if (!f) { set_err("cannot append to %s: %s", path, strerror(errno)); return -1; }
    // spgemm.cu:1432-1448: "\n" then 14 fields, fixed with 2 decimals for the floating ones
    fprintf(f, "\n%s,%llu,%lld,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f", r->matrix ? r->matrix : "", (unsigned long long)r->flop,
            (long long)r->c_nnz, r->compression_ratio, r->a_conversion_kernel_ms, r->b_conversion_kernel_ms, r->total_conversion_ms, r->step1_ms,
            r->step2_ms, r->step3_ms, r->spgemm_ms, r->kernel_ms, r->malloc_ms, r->gflops);
    if (extra && *extra) fprintf(f, ",%s", extra);
    fclose(f);
    return 0;
}

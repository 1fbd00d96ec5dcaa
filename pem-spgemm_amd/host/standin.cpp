// standin.cpp -- seeded synthetic stand-ins for the SuiteSparse inputs BASELINE.json names ("generator in
// C++, splitmix64 / xoshiro256**, fixed seeds, values uniform in [-1, 1) excluding 0, no duplicates, sorted rows").
// No SuiteSparse file exists offline, so every benchmark configuration is generated; `pemspgemm --standin NAME` and
// bench.py go through pem_standin_generate.  Round 3: the webbase-1M and cage15 models are calibrated so that the
// PRODUCT matches the literature (webbase-1M A^2 flop 69.5 M / C nnz 51.1 M, cage15 A^2 ~2.08 G / ~0.93 G),
// not only shape, nnz and degree skew -- the round-2 generators (kept as "<name>-r2" in standins.py) compressed 1.02x and
// 1.08x where the real products compress 1.36x and 2.24x, so step 3's accumulate chain was barely exercised.
//
//   webbase-1M  pages grouped in hosts (consecutive index ranges, power-law sizes up to 250).  A host's first page (and
//               the second one of hosts of 8+ pages) is an index page: it links to most pages of the host and most pages
//               link back to it -- the navigation structure that makes i -> k -> j reach the same j through several k.
//               Plus a few random links inside the host, a few to out-degree-proportional pages anywhere, and forty
//               directory pages with up to 4 700 links (the real matrix's largest row).
//   cage15      a symmetric pattern on a two-scale lattice: neighbours i +- c*1 and i +- c*997, c = 1..7, each edge kept
//               with a probability that varies smoothly along the index (degree 8..29, mean 19.2); node labels shuffled
//               inside blocks of 64 rows, which thins the 16x16 tiles (3.8 entries per tile) without changing the graph.
//               Sums of lattice steps collide, so A^2 compresses like the real matrix (the DNA-electrophoresis state graphs
//               of the cage family are products of small move sets).
//   scircuit, mc2depi, cage4: the round-2 models (shape, nnz and degree structure), ported.
#include "../../include/pem_host.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {   // xoshiro256** seeded through splitmix64
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x)
    {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed)
    {
        for (auto &w : s) w = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
    uint64_t below(uint64_t n) { return n ? (uint64_t)(((unsigned __int128)next() * n) >> 64) : 0; }
    int poisson(double lambda)   // small lambda: Knuth
    {
        const double L = std::exp(-lambda);
        int k = 0;
        double p = 1.0;
        do {
            ++k;
            p *= uniform();
        } while (p > L);
        return k - 1;
    }
    // integer ~ x^-alpha on [lo, hi]
    int64_t powerlaw(double alpha, double lo, double hi)
    {
        const double a = 1.0 - alpha, u = uniform();
        return (int64_t)std::floor(std::pow(std::pow(lo, a) + u * (std::pow(hi, a) - std::pow(lo, a)), 1.0 / a));
    }
};

// stateless hash -> [0, 1): decisions that both endpoints of an edge must agree on, and the values
inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xFF51AFD7ED558CCDull;
    x ^= x >> 33;
    x *= 0xC4CEB9FE1A85EC53ull;
    x ^= x >> 33;
    return x;
}
inline double hash01(uint64_t a, uint64_t b, uint64_t seed) { return (double)(mix64(mix64(a + seed) ^ (b * 0x9E3779B97F4A7C15ull)) >> 11) * (1.0 / 9007199254740992.0); }
inline double value_of(uint64_t i, uint64_t j, uint64_t seed)
{
    double v = hash01(i, j, seed ^ 0xA5A5A5A5ull) * 2.0 - 1.0;
    return v == 0.0 ? 0.5 : v;
}

// keys (i * cols + j) -> sorted unique; then exactly `want` entries: evenly spread entries are dropped, or -- when the draw
// fell short -- evenly spread rows get one more column
void finish_keys(std::vector<uint64_t> &key, int64_t rows, int64_t cols, int64_t want, uint64_t seed)
{
    std::sort(key.begin(), key.end());
    key.erase(std::unique(key.begin(), key.end()), key.end());
    if (want <= 0 || want > rows * cols) return;
    if ((int64_t)key.size() > want) {
        const size_t have = key.size(), drop = have - (size_t)want;
        std::vector<uint64_t> out;
        out.reserve((size_t)want);
        // drop entry floor(k * have / drop), k = 0..drop-1
        size_t k = 0, next_drop = 0;
        for (size_t x = 0; x < have; ++x) {
            if (k < drop && x == next_drop) {
                ++k;
                next_drop = (size_t)(((unsigned __int128)k * have) / drop);
                continue;
            }
            out.push_back(key[x]);
        }
        key.swap(out);
    }
    if ((int64_t)key.size() < want) {          // top up with uniform entries that are not there yet
        Rng rng(seed ^ 0x70707070ull);
        std::vector<uint64_t> add;
        const size_t need = (size_t)want - key.size();
        while (add.size() < need) {
            const uint64_t k = rng.below((uint64_t)rows) * (uint64_t)cols + rng.below((uint64_t)cols);
            if (std::binary_search(key.begin(), key.end(), k)) continue;
            if (std::find(add.begin(), add.end(), k) != add.end() && add.size() < 4096) continue;
            add.push_back(k);
        }
        std::sort(add.begin(), add.end());
        add.erase(std::unique(add.begin(), add.end()), add.end());   // (large top-ups: duplicates among the new ones are rare; re-draw the few)
        while (add.size() < need) {
            const uint64_t k = rng.below((uint64_t)rows) * (uint64_t)cols + rng.below((uint64_t)cols);
            if (std::binary_search(key.begin(), key.end(), k) || std::binary_search(add.begin(), add.end(), k)) continue;
            add.insert(std::upper_bound(add.begin(), add.end(), k), k);
        }
        const size_t mid = key.size();
        key.insert(key.end(), add.begin(), add.end());
        std::inplace_merge(key.begin(), key.begin() + (long)mid, key.end());
    }
}

int emit(const std::vector<uint64_t> &key, int64_t rows, int64_t cols, uint64_t seed, pem_coo *out)
{
    const size_t n = key.size();
    int32_t *I = static_cast<int32_t *>(malloc(sizeof(int32_t) * (n ? n : 1)));
    int32_t *J = static_cast<int32_t *>(malloc(sizeof(int32_t) * (n ? n : 1)));
    double *V = static_cast<double *>(malloc(sizeof(double) * (n ? n : 1)));
    if (!I || !J || !V) {
        free(I);
        free(J);
        free(V);
        return -6;
    }
    const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
            for (size_t x = n * t / nt; x < n * (t + 1) / nt; ++x) {
                const uint64_t i = key[x] / (uint64_t)cols, j = key[x] % (uint64_t)cols;
                I[x] = (int32_t)i;
                J[x] = (int32_t)j;
                V[x] = value_of(i, j, seed);
            }
        });
    for (auto &t : th) t.join();
    out->rows = (int32_t)rows;
    out->cols = (int32_t)cols;
    out->nnz = (int64_t)n;
    out->I = I;
    out->J = J;
    out->V = V;
    out->symmetric = 0;
    out->field = 0;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
int gen_cage4(pem_coo *out)
{
    // 9 x 9, 49 entries: full diagonal + 40 off-diagonals -- the pattern of the round-1 fixture (tests/golden/mm_cage4_standin.mtx;
    // the real cage4 is not available offline), one row mask per row
    static const unsigned rowmask[9] = {0x011, 0x033, 0x10F, 0x0BF, 0x0F7, 0x129, 0x0C9, 0x1F5, 0x1FF};
    std::vector<uint64_t> key;
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j)
            if ((rowmask[i] >> j) & 1u) key.push_back((uint64_t)i * 9 + (uint64_t)j);
    return emit(key, 9, 9, 4, out);
}

int gen_scircuit(double scale, pem_coo *out)
{
    const int64_t n = std::max<int64_t>(32, (int64_t)(170998 * scale)), nnz = std::max<int64_t>(64, (int64_t)(958936 * scale));
    Rng rng(171);
    // circuit-like: diagonal + power-law(2.0) extra degree capped at 353; 60 % of the columns within +-64 of the diagonal
    std::vector<int64_t> deg((size_t)n);
    double sum = 0;
    for (auto &d : deg) {
        d = rng.powerlaw(2.0, 1.0, 353.0);
        sum += (double)d;
    }
    const double f = (double)(nnz - n) * 1.03 / sum;
    std::vector<uint64_t> key;
    key.reserve((size_t)(nnz * 1.1));
    for (int64_t i = 0; i < n; ++i) {
        key.push_back((uint64_t)i * (uint64_t)n + (uint64_t)i);
        const int64_t d = std::min<int64_t>(353, (int64_t)std::floor((double)deg[(size_t)i] * f + rng.uniform()));
        for (int64_t k = 0; k < d; ++k) {
            int64_t j = rng.uniform() < 0.6 ? i + (int64_t)rng.below(129) - 64 : (int64_t)rng.below((uint64_t)n);
            j = std::min(n - 1, std::max<int64_t>(0, j));
            key.push_back((uint64_t)i * (uint64_t)n + (uint64_t)j);
        }
    }
    finish_keys(key, n, n, nnz, 171);
    return emit(key, n, n, 171, out);
}

int gen_mc2depi(double scale, pem_coo *out)
{
    const int64_t n = std::max<int64_t>(64, (int64_t)(525825 * scale));
    const int64_t nnz = scale == 1.0 ? 2100225 : std::min<int64_t>(4 * n - 8, std::max<int64_t>(128, (int64_t)(2100225 * scale)));
    const int64_t w = (int64_t)std::sqrt((double)n);
    std::vector<uint64_t> key;
    key.reserve((size_t)(4 * n));
    for (int64_t i = 0; i < n; ++i) {   // banded epidemiology-like: columns {i-1, i, i+1, i+floor(sqrt(n))} clipped
        key.push_back((uint64_t)i * (uint64_t)n + (uint64_t)i);
        if (i > 0) key.push_back((uint64_t)i * (uint64_t)n + (uint64_t)(i - 1));
        if (i + 1 < n) key.push_back((uint64_t)i * (uint64_t)n + (uint64_t)(i + 1));
        if (i + w < n) key.push_back((uint64_t)i * (uint64_t)n + (uint64_t)(i + w));
    }
    finish_keys(key, n, n, nnz, 526);
    return emit(key, n, n, 526, out);
}

// webbase-1M: see the head of the file.  Calibrated (scale 1): nnz 3 105 536, A^2 flop / C nnz printed by every run.
int gen_webbase(double scale, pem_coo *out)
{
    const int64_t n = std::max<int64_t>(64, (int64_t)(1000005 * scale)), nnz = std::max<int64_t>(128, (int64_t)(3105536 * scale));
    const uint64_t seed = 1000;
    Rng rng(seed);
    constexpr double HOST_ALPHA = 2.0, HOST_MAX = 250.0;
    constexpr double C0 = 0.90, C1 = 0.65;       // index page t links to a page of its host with probability Ct
    constexpr double Q0 = 0.90, Q1 = 0.70;       // a page links to index page t of its host with probability Qt
    constexpr double X_LOCAL = 0.50, X_GLOBAL = 0.35;
    std::vector<int64_t> hstart;
    for (int64_t at = 0; at < n;) {
        hstart.push_back(at);
        at += std::max<int64_t>(1, rng.powerlaw(HOST_ALPHA, 1.0, HOST_MAX));
    }
    hstart.push_back(n);
    std::vector<uint64_t> key;
    key.reserve((size_t)(nnz * 1.15));
    std::vector<int32_t> src;                     // source page of every link so far: a random element is an out-degree-proportional page
    src.reserve((size_t)(nnz * 1.15));
    auto link = [&](int64_t i, int64_t j) {
        key.push_back((uint64_t)i * (uint64_t)n + (uint64_t)j);
        src.push_back((int32_t)i);
    };
    for (size_t h = 0; h + 1 < hstart.size(); ++h) {
        const int64_t s0 = hstart[h], sz = hstart[h + 1] - s0;
        const int nidx = sz >= 8 ? 2 : sz >= 2 ? 1 : 0;
        for (int t = 0; t < nidx; ++t) {
            const double c = t == 0 ? C0 : C1, q = t == 0 ? Q0 : Q1;
            for (int64_t pg = s0; pg < s0 + sz; ++pg) {
                if (pg == s0 + t) continue;
                if (rng.uniform() < q) link(pg, s0 + t);
                if (rng.uniform() < c) link(s0 + t, pg);
            }
        }
        for (int64_t pg = s0; pg < s0 + sz; ++pg)
            for (int k = rng.poisson(X_LOCAL); k > 0; --k) link(pg, s0 + (int64_t)rng.below((uint64_t)sz));
    }
    const size_t nlocal = src.size();
    for (int64_t pg = 0; pg < n; ++pg)
        for (int k = rng.poisson(X_GLOBAL); k > 0; --k) link(pg, src[(size_t)rng.below(nlocal)]);
    // directory pages: the real matrix's largest rows (4 700 entries)
    const int ndir = (int)std::max<int64_t>(1, (int64_t)(40 * std::min(1.0, scale * 4)));
    for (int d = 0; d < ndir; ++d) {
        const int64_t pg = (int64_t)rng.below((uint64_t)n);
        const int64_t deg = std::min<int64_t>(n / 2, d == 0 ? 4700 : rng.powerlaw(1.3, 300.0, 4700.0));
        for (int64_t k = 0; k < deg; ++k) link(pg, (int64_t)rng.below((uint64_t)n));
    }
    finish_keys(key, n, n, nnz, seed);
    return emit(key, n, n, seed, out);
}

// cage15: see the head of the file.  Built row by row (no global sort): an edge {u, v} is kept by a hash both ends agree on.
int gen_cage15(double scale, pem_coo *out)
{
    const int64_t n = std::max<int64_t>(1024, (int64_t)(5154859 * scale)), want = std::max<int64_t>(4096, (int64_t)(99199551 * scale));
    const uint64_t seed = 5150;
    constexpr int CMAX = 7, BLK = 64;
    constexpr int64_t SCALE2 = 997;
    constexpr double P0 = 0.662, PVAR = 0.45, PERIOD = 20000.0;
    auto keep_p = [&](int64_t i) { return P0 * (1.0 + PVAR * std::sin(2.0 * M_PI * (double)i / PERIOD + 1.0)); };
    // node relabelling inside blocks of BLK rows: perm[old] = new
    std::vector<int32_t> perm((size_t)n), inv((size_t)n);
    {
        Rng rng(seed);
        for (int64_t b0 = 0; b0 < n; b0 += BLK) {
            const int64_t e = std::min(n, b0 + BLK);
            for (int64_t x = b0; x < e; ++x) perm[(size_t)x] = (int32_t)x;
            for (int64_t x = e - 1; x > b0; --x) std::swap(perm[(size_t)x], perm[(size_t)(b0 + (int64_t)rng.below((uint64_t)(x - b0 + 1)))]);
        }
        for (int64_t x = 0; x < n; ++x) inv[(size_t)perm[(size_t)x]] = (int32_t)x;
    }
    int64_t offs[4 * CMAX];
    int noff = 0;
    for (int c = 1; c <= CMAX; ++c) {
        offs[noff++] = c;
        offs[noff++] = -c;
        offs[noff++] = c * SCALE2;
        offs[noff++] = -c * SCALE2;
    }
    const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::vector<uint64_t>> part(nt);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
            std::vector<uint64_t> &k = part[t];
            const int64_t r0 = n * t / nt, r1 = n * (t + 1) / nt;
            k.reserve((size_t)((r1 - r0) * 20));
            int64_t rowbuf[4 * CMAX + 1];
            for (int64_t r = r0; r < r1; ++r) {   // r = new label; rows come out in order
                const int64_t i = inv[(size_t)r];
                int m = 0;
                rowbuf[m++] = r;                  // diagonal
                const double pi = keep_p(i);
                for (int o = 0; o < noff; ++o) {
                    const int64_t j = i + offs[o];
                    if (j < 0 || j >= n) continue;
                    const int64_t lo = std::min(i, j), hi = std::max(i, j);
                    if (hash01((uint64_t)lo, (uint64_t)hi, seed) < 0.5 * (pi + keep_p(j))) rowbuf[m++] = perm[(size_t)j];
                }
                std::sort(rowbuf, rowbuf + m);
                for (int x = 0; x < m; ++x) k.push_back((uint64_t)r * (uint64_t)n + (uint64_t)rowbuf[x]);
            }
        });
    for (auto &t : th) t.join();
    std::vector<uint64_t> key;
    size_t total = 0;
    for (auto &p : part) total += p.size();
    key.reserve(total + 1024);
    for (auto &p : part) {
        key.insert(key.end(), p.begin(), p.end());
        std::vector<uint64_t>().swap(p);
    }
    // already sorted and duplicate-free; to exactly `want` entries
    if ((int64_t)key.size() != want) finish_keys(key, n, n, want, seed);
    return emit(key, n, n, seed, out);
}

}  // namespace

extern "C" const char *pem_standin_names(void) { return "cage4 scircuit webbase-1M mc2depi cage15"; }

extern "C" int pem_standin_generate(const char *name, double scale, pem_coo *out)
{
    if (!name || !out || !(scale > 0.0) || scale > 1.0) return -1;
    memset(out, 0, sizeof *out);
    const std::string s(name);
    if (s == "cage4") return gen_cage4(out);
    if (s == "scircuit") return gen_scircuit(scale, out);
    if (s == "webbase-1M") return gen_webbase(scale, out);
    if (s == "mc2depi") return gen_mc2depi(scale, out);
    if (s == "cage15") return gen_cage15(scale, out);
    return -2;
}

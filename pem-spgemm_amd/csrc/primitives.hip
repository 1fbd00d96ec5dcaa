// primitives.hip -- device-wide primitives for gfx950 (wave64), written once for the path.
// Replaces thrust::exclusive_scan / thrust::sort / thrust::stable_sort / unique /
// reduce_by_key call sites (spgemm.cu:869-927, 990-1061, 1168, 1242, 1288) and the warp
// scan of NSPARSE/utils_cuda_scan.h:19-35.  No thrust, no rocPRIM: plain HIP.
#include "pem_internal.h"
#include <algorithm>
#include <cstdarg>

namespace pem {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

// ------------------------------------------------------------------------------------------
// device memory arena (pem_internal.h)
// ------------------------------------------------------------------------------------------
std::shared_ptr<Arena> &current_arena()
{
    static thread_local std::shared_ptr<Arena> a;
    return a;
}

static constexpr size_t ARENA_ALIGN = 256;
static constexpr size_t ARENA_MIN_SLAB = size_t(64) << 20;     // small requests share 64 MiB slabs
static constexpr size_t ARENA_MAX_GROWTH = size_t(4) << 30;    // ... a slab is never padded by more than this

Arena::~Arena()
{
    // the owners of every block hold a reference to the arena, so nothing is in use any more
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    for (auto &s : slabs) (void)hipFree(s.base);
}

pem_status Arena::add_slab(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc(%zu bytes) failed: %s (arena holds %zu bytes, %zu in use)", bytes, hipGetErrorString(e), slab_bytes, in_use);
        return PEM_E_NOMEM;
    }
    slabs.push_back(Slab{static_cast<char *>(p), bytes});
    slab_bytes += bytes;
    ++n_driver;
    free_blocks[static_cast<char *>(p)] = bytes;
    return PEM_OK;
}

void *Arena::take(size_t bytes)
{
    auto best = free_blocks.end();
    for (auto it = free_blocks.begin(); it != free_blocks.end(); ++it)
        if (it->second >= bytes && (best == free_blocks.end() || it->second < best->second)) best = it;
    if (best == free_blocks.end()) return nullptr;
    char *p = best->first;
    const size_t have = best->second;
    free_blocks.erase(best);
    if (have > bytes) free_blocks[p + bytes] = have - bytes;
    used_blocks[p] = bytes;
    in_use += bytes;
    if (in_use > peak) peak = in_use;
    ++n_block;
    return p;
}

void *Arena::alloc(size_t bytes)
{
    bytes = (bytes + ARENA_ALIGN - 1) & ~(ARENA_ALIGN - 1);
    if (bytes == 0) bytes = ARENA_ALIGN;
    std::lock_guard<std::mutex> lock(mu);
    if (void *p = take(bytes)) return p;
    // grow: the request itself, padded so that a run of growing requests (a first pass sizes its buffers one after the
    // other) does not end up as one driver call each
    size_t pad = slab_bytes / 2;
    if (pad > ARENA_MAX_GROWTH) pad = ARENA_MAX_GROWTH;
    size_t want = bytes + pad;
    if (want < ARENA_MIN_SLAB) want = ARENA_MIN_SLAB;
    if (add_slab(want) != PEM_OK) {
        if (want == bytes || add_slab(bytes) != PEM_OK) return nullptr;   // retry without the padding
    }
    return take(bytes);
}

void Arena::free(void *vp)
{
    if (!vp) return;
    char *p = static_cast<char *>(vp);
    std::lock_guard<std::mutex> lock(mu);
    auto u = used_blocks.find(p);
    if (u == used_blocks.end()) return;   // not ours (cannot happen: DevBuf remembers its arena)
    size_t bytes = u->second;
    used_blocks.erase(u);
    in_use -= bytes;
    // coalesce with the neighbours -- but never across a slab boundary (slabs may happen to abut)
    auto slab_of = [&](char *q) -> const Slab * {
        for (auto &s : slabs)
            if (q >= s.base && q < s.base + s.size) return &s;
        return nullptr;
    };
    const Slab *sl = slab_of(p);
    auto next = free_blocks.lower_bound(p);
    if (next != free_blocks.end() && next->first == p + bytes && sl && next->first < sl->base + sl->size) {
        bytes += next->second;
        next = free_blocks.erase(next);
    }
    if (next != free_blocks.begin()) {
        auto prev = std::prev(next);
        if (prev->first + prev->second == p && sl && prev->first >= sl->base) {
            prev->second += bytes;
            return;
        }
    }
    free_blocks[p] = bytes;
}

pem_status Arena::reserve(size_t bytes)
{
    bytes = (bytes + ARENA_ALIGN - 1) & ~(ARENA_ALIGN - 1);
    std::lock_guard<std::mutex> lock(mu);
    for (auto &f : free_blocks)
        if (f.second >= bytes) return PEM_OK;
    return add_slab(bytes);
}

pem_status Arena::reserve_many(const size_t *sizes, int n)
{
    std::lock_guard<std::mutex> lock(mu);
    std::vector<size_t> want;
    for (int i = 0; i < n; ++i)
        if (sizes[i]) want.push_back((sizes[i] + ARENA_ALIGN - 1) & ~(ARENA_ALIGN - 1));
    std::sort(want.begin(), want.end(), [](size_t a, size_t b) { return a > b; });
    std::vector<size_t> avail;
    for (auto &f : free_blocks) avail.push_back(f.second);
    size_t missing = 0;
    for (size_t w : want) {            // best fit over a copy of the free list, largest request first
        long best = -1;
        for (size_t i = 0; i < avail.size(); ++i)
            if (avail[i] >= w && (best < 0 || avail[i] < avail[(size_t)best])) best = (long)i;
        if (best >= 0)
            avail[(size_t)best] -= w;
        else
            missing += w;
    }
    return missing ? add_slab(missing) : PEM_OK;
}

pem_status arena_phase(const std::shared_ptr<Arena> &arena, std::initializer_list<PhaseWant> wants)
{
    size_t sizes[32];
    int n = 0;
    for (auto &w : wants)
        if (w.bytes > w.buf->cap && n < 32) sizes[n++] = w.bytes;
    return n ? arena->reserve_many(sizes, n) : PEM_OK;
}

void Arena::trim()
{
    std::lock_guard<std::mutex> lock(mu);
    ++generation;
    for (size_t i = 0; i < slabs.size();) {
        auto f = free_blocks.find(slabs[i].base);
        if (f != free_blocks.end() && f->second == slabs[i].size) {
            free_blocks.erase(f);
            (void)hipFree(slabs[i].base);
            slab_bytes -= slabs[i].size;
            slabs.erase(slabs.begin() + (long)i);
        } else {
            ++i;
        }
    }
}

Arena::Stats Arena::stats()
{
    std::lock_guard<std::mutex> lock(mu);
    size_t largest = 0;
    for (auto &f : free_blocks)
        if (f.second > largest) largest = f.second;
    return Stats{slab_bytes, in_use, peak, largest, n_driver, n_block};
}

// ------------------------------------------------------------------------------------------
// kernel spans
// ------------------------------------------------------------------------------------------
static hipEvent_t take_event(pem_ctx *ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

KernelSpan::KernelSpan(pem_ctx *c, const char *name) : ctx(c)
{
    if (!ctx->profiling) return;
    for (size_t i = 0; i < ctx->stats.size(); ++i)
        if (ctx->stats[i].name == name) { stat = (int)i; break; }
    if (stat < 0) {
        ctx->stats.push_back(KernelStat{name, 0, 0.0});
        stat = (int)ctx->stats.size() - 1;
    }
    e0 = take_event(ctx);
    e1 = take_event(ctx);
    (void)hipEventRecord(e0, ctx->stream);
}
KernelSpan::~KernelSpan()
{
    if (stat < 0) return;
    (void)hipEventRecord(e1, ctx->stream);
    ctx->pending.push_back(PendingSpan{stat, e0, e1});
}

pem_status resolve_kernel_spans(pem_ctx *ctx)
{
    if (ctx->pending.empty()) return PEM_OK;
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &s : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.e0, s.e1) == hipSuccess) {
            ctx->stats[s.stat].calls += 1;
            ctx->stats[s.stat].total_ms += ms;
        }
        ctx->event_pool.push_back(s.e0);
        ctx->event_pool.push_back(s.e1);
    }
    ctx->pending.clear();
    return PEM_OK;
}

pem_status zero_flags(pem_ctx *ctx)
{
    PEM_HIP(hipMemsetAsync(ctx->d_flags, 0, sizeof(int) * NUM_FLAGS, ctx->stream));
    return PEM_OK;
}

pem_status launch_status(pem_ctx *ctx)
{
    if (ctx->launch_err == hipSuccess) return PEM_OK;
    set_error("kernel launch refused by the runtime: %s: %s", ctx->launch_name ? ctx->launch_name : "?", hipGetErrorString(ctx->launch_err));
    ctx->launch_err = hipSuccess;
    ctx->launch_name = nullptr;
    return PEM_E_HIP;
}

pem_status read_flags(pem_ctx *ctx, int *host_flags)
{
    PEM_TRY(launch_status(ctx));
    int *h = reinterpret_cast<int *>(ctx->h_scalars + 48);
    PEM_HIP(hipMemcpyAsync(h, ctx->d_flags, sizeof(int) * NUM_FLAGS, hipMemcpyDeviceToHost, ctx->stream));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < NUM_FLAGS; ++i) host_flags[i] = h[i];
    return PEM_OK;
}

pem_status read_scalars(pem_ctx *ctx, const int64_t *d_src, int count, int64_t *host_dst)
{
    PEM_HIP(hipMemcpyAsync(ctx->h_scalars, d_src, sizeof(int64_t) * count, hipMemcpyDeviceToHost, ctx->stream));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < count; ++i) host_dst[i] = ctx->h_scalars[i];
    return PEM_OK;
}

// ------------------------------------------------------------------------------------------
// exclusive scan: reduce-then-scan, 3 launches, 2 reads + 1 write of the data
// ------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_TILE = SCAN_THREADS * 4;      // int4 per thread
constexpr int SCAN_TILES_PER_BLOCK = 8;
constexpr int SCAN_BLOCK_ITEMS = SCAN_TILE * SCAN_TILES_PER_BLOCK;

template <typename T> __device__ __forceinline__ T wave_inclusive_scan(T v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

template <typename T> __device__ __forceinline__ T wave_reduce_sum(T v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_reduce_kernel(const int *__restrict__ in, size_t n,
                                                                   long long *__restrict__ bsum)
{
    __shared__ long long wsum[SCAN_THREADS / 64];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK_ITEMS;
    size_t end = base + SCAN_BLOCK_ITEMS < n ? base + SCAN_BLOCK_ITEMS : n;
    long long s = 0;
    if (end - base == (size_t)SCAN_BLOCK_ITEMS && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {   // full block: 16-byte loads
        const int4 *in4 = reinterpret_cast<const int4 *>(in + base);
#pragma unroll
        for (int k = 0; k < SCAN_BLOCK_ITEMS / 4 / SCAN_THREADS; ++k) {
            const int4 q = in4[k * SCAN_THREADS + threadIdx.x];
            s += (long long)q.x + q.y + q.z + q.w;
        }
    } else {
        for (size_t i = base + threadIdx.x; i < end; i += SCAN_THREADS) s += in[i];
    }
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; ++w) t += wsum[w];
        bsum[blockIdx.x] = t;
    }
}

// single block: in-place exclusive scan of the block sums; total -> bsum[nblk], *total64, overflow flag
__global__ void __launch_bounds__(1024) scan_bsums_kernel(long long *__restrict__ bsum, int nblk,
                                                          long long *__restrict__ total64, int *__restrict__ flags)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        int i = base + threadIdx.x;
        long long v = i < nblk ? bsum[i] : 0;
        long long inc = wave_inclusive_scan(v);
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        long long carry = carry_s;
        if (i < nblk) bsum[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        long long t = carry_s;
        bsum[nblk] = t;
        if (total64) *total64 = t;
        if (t > 0x7FFFFFFFLL) flags[FLAG_OVERFLOW] = 1;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const int *in, int *out, size_t n,
                                                                  const long long *__restrict__ bsum, int nblk)
{
    __shared__ int wsum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long carry = bsum[blockIdx.x];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK_ITEMS;
    for (int tile = 0; tile < SCAN_TILES_PER_BLOCK; ++tile, base += SCAN_TILE) {
        if (base >= n) break;
        size_t i0 = base + (size_t)threadIdx.x * 4;
        int v[4];
        if (i0 + 4 <= n) {
            int4 q = *reinterpret_cast<const int4 *>(in + i0);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n) ? in[i0 + k] : 0;
        }
        int tsum = v[0] + v[1] + v[2] + v[3];
        int inc = wave_inclusive_scan(tsum);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0, btotal = 0;
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; ++w) {
            int s = wsum[w];
            if (w < wave) woff += s;
            btotal += s;
        }
        int ex = (int)carry + woff + inc - tsum;
        int o0 = ex, o1 = o0 + v[0], o2 = o1 + v[1], o3 = o2 + v[2];
        if (i0 + 4 <= n) {
            *reinterpret_cast<int4 *>(out + i0) = make_int4(o0, o1, o2, o3);
        } else {
            if (i0 < n) out[i0] = o0;
            if (i0 + 1 < n) out[i0 + 1] = o1;
            if (i0 + 2 < n) out[i0 + 2] = o2;
        }
        carry += btotal;
        __syncthreads();
    }
    if (blockIdx.x == nblk - 1 && threadIdx.x == 0) out[n] = (int)bsum[nblk];
}

__global__ void scan_empty_kernel(int *out, long long *total64)
{
    out[0] = 0;
    if (total64) *total64 = 0;
}

// short arrays (row counts of a slice, histogram tails): one 1024-thread block does the whole scan in one launch
__global__ void __launch_bounds__(1024) scan_small_kernel(const int *in, int *out, size_t n, long long *__restrict__ total64,
                                                          int *__restrict__ flags)
{
    // eight items per thread and trip (two 16-byte loads): half the trips, and barriers, of a four-item loop
    __shared__ long long wsum[16];   // 64-bit sums: a total beyond int32 must be reported, not wrapped
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long carry = 0;
    for (size_t base = 0; base < n; base += 8192) {
        const size_t i0 = base + (size_t)threadIdx.x * 8;
        int v[8];
        if (i0 + 8 <= n) {
            const int4 q0 = *reinterpret_cast<const int4 *>(in + i0), q1 = *reinterpret_cast<const int4 *>(in + i0 + 4);
            v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
            v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (i0 + k < n) ? in[i0 + k] : 0;
        }
        long long tsum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) tsum += v[k];
        const long long inc = wave_inclusive_scan(tsum);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        long long woff = 0, btotal = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const long long c = wsum[w];
            if (w < wave) woff += c;
            btotal += c;
        }
        int o[8];
        o[0] = (int)(carry + woff + inc - tsum);
#pragma unroll
        for (int k = 1; k < 8; ++k) o[k] = o[k - 1] + v[k - 1];
        if (i0 + 8 <= n) {
            *reinterpret_cast<int4 *>(out + i0) = make_int4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<int4 *>(out + i0 + 4) = make_int4(o[4], o[5], o[6], o[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i0 + k < n) out[i0 + k] = o[k];
        }
        carry += btotal;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[n] = (int)carry;
        if (total64) *total64 = carry;
        if (carry > 0x7FFFFFFFLL) flags[FLAG_OVERFLOW] = 1;
    }
}

// mid-size arrays (tile-row counts of a plan, 256-tile group counts): ONE launch of up to SCAN_MID_BLOCKS blocks.  Every
// block scans 2048 items, publishes its total, and sums the totals of all earlier blocks -- at most 127 words, two per
// lane of one wave, read in one round trip.
// Which 2048 items a block takes is decided by a TICKET (one atomic per block), not by blockIdx: HIP promises nothing
// about dispatch order, and these scans run while row-sort workgroups from the auxiliary streams hold most of the CUs.
// A block that holds ticket k only ever waits for tickets < k, and whoever drew those is already running and publishes
// without waiting for anybody: the chain cannot deadlock whatever the order of dispatch.  in == out is allowed: a block
// reads its own items before it writes them and nobody else reads them (there is no "re-sum the input" fallback any
// more -- the one this kernel had was wrong for in-place scans).  Should the wait ever run out (it cannot, see above),
// FLAG_INTERNAL makes the host fail the call instead of returning a wrong scan.  The last block to finish clears the
// words for the next scan.  A single 1024-thread block took 17 us for 62 k items and 35 us for 75 k.
constexpr int SCAN_MID_ITEMS = 2048, SCAN_MID_BLOCKS = 128;
constexpr unsigned long long SCAN_MID_VALID = 1ull << 63;

// (blockIdx.y: up to two independent arrays scanned by one launch -- step 1 scans its per-row tile counts and its per-256-pairs
// first-pair counts back to back; each array has its own look-back words, and blocks past an array's end leave at once)
struct ScanMidArgs {
    const int *in[2];
    int *out[2];
    size_t n[2];
    long long *total64[2];
    int nblk[2];
};
__global__ void __launch_bounds__(256) scan_mid_kernel(ScanMidArgs args, unsigned long long *state_all, int *__restrict__ flags, int stall_ticket)
{
    __shared__ long long wsum[4];
    __shared__ long long s_excl;
    __shared__ int s_ticket;
    const int y = blockIdx.y;
    const int nblk = args.nblk[y];
    if ((int)blockIdx.x >= nblk) return;
    const int *in = args.in[y];
    int *out = args.out[y];
    const size_t n = args.n[y];
    long long *total64 = args.total64[y];
    unsigned long long *state = state_all + (size_t)y * (SCAN_MID_BLOCKS + 2);
    int *ctl = reinterpret_cast<int *>(state + SCAN_MID_BLOCKS);   // done, ticket
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = atomicAdd(&ctl[1], 1);
    __syncthreads();
    const int blk = s_ticket;
    const size_t i0 = (size_t)blk * SCAN_MID_ITEMS + (size_t)tid * 8;
    int v[8];
    if (i0 + 8 <= n) {
        const int4 q0 = *reinterpret_cast<const int4 *>(in + i0), q1 = *reinterpret_cast<const int4 *>(in + i0 + 4);
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
        v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i0 + k < n) ? in[i0 + k] : 0;
    }
    long long tsum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tsum += v[k];
    const long long inc = wave_inclusive_scan(tsum);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    long long woff = 0, btotal = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) woff += wsum[w];
        btotal += wsum[w];
    }
    if (wave == 0) {
        if (blk == stall_ticket)                    // test hook: every later ticket has to sit out a long wait
            for (int k = 0; k < 50; ++k) __builtin_amdgcn_s_sleep(127);   // ~0.2 ms
        // (62 bits of total: the block's own sum can exceed 2^31 when the scan as a whole overflows -- flagged below)
        if (lane == 0) __hip_atomic_store(&state[blk], SCAN_MID_VALID | (unsigned long long)btotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long excl = 0;
        bool ok = false;
        for (long long polls = 0; polls < (1ll << 24) && !ok; ++polls) {
            const unsigned long long a = lane < blk ? __hip_atomic_load(&state[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SCAN_MID_VALID;
            const unsigned long long b = 64 + lane < blk ? __hip_atomic_load(&state[64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SCAN_MID_VALID;
            if (__ballot(!(a & SCAN_MID_VALID) || !(b & SCAN_MID_VALID)) == 0) {
                excl = wave_reduce_sum((long long)(a & ~SCAN_MID_VALID) + (long long)(b & ~SCAN_MID_VALID));
                ok = true;
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (!ok && lane == 0) flags[FLAG_INTERNAL] = 1;
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    const long long carry = s_excl;
    int o[8];
    o[0] = (int)(carry + woff + inc - tsum);
#pragma unroll
    for (int k = 1; k < 8; ++k) o[k] = o[k - 1] + v[k - 1];
    if (i0 + 8 <= n) {
        *reinterpret_cast<int4 *>(out + i0) = make_int4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<int4 *>(out + i0 + 4) = make_int4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + k < n) out[i0 + k] = o[k];
    }
    if (tid == 0) {
        if (blk == nblk - 1) {
            const long long total = carry + btotal;
            out[n] = (int)total;
            if (total64) *total64 = total;
            if (total > 0x7FFFFFFFLL) flags[FLAG_OVERFLOW] = 1;
        }
        // the last block to get here has seen every other block read what it needed: clear the words for the next scan
        if (atomicAdd(&ctl[0], 1) == nblk - 1) {
            for (int b = 0; b < nblk; ++b) __hip_atomic_store(&state[b], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static pem_status scan_mid_state(pem_ctx *ctx)
{
    if (!ctx->scan_state.p) {   // look-back words + completion counter (two sets: scan_mid_kernel takes two arrays), zero between scans
        PEM_TRY(ctx->scan_state.reserve(sizeof(unsigned long long) * 2 * (SCAN_MID_BLOCKS + 2)));
        PEM_HIP(hipMemsetAsync(ctx->scan_state.p, 0, sizeof(unsigned long long) * 2 * (SCAN_MID_BLOCKS + 2), ctx->stream));
    }
    return PEM_OK;
}

// two independent arrays, each in place or out of place: one launch where both are mid-size (8192 < n <= 262144), else two calls
pem_status exclusive_scan_i32_two(pem_ctx *ctx, const int *in_a, int *out_a, size_t na, int64_t *d_total_a, const int *in_b, int *out_b, size_t nb,
                                  int64_t *d_total_b)
{
    auto mid = [](size_t n) { return n > 8192 && n <= (size_t)SCAN_MID_ITEMS * SCAN_MID_BLOCKS; };
    const bool aligned = !((reinterpret_cast<uintptr_t>(in_a) | reinterpret_cast<uintptr_t>(out_a) | reinterpret_cast<uintptr_t>(in_b) |
                            reinterpret_cast<uintptr_t>(out_b)) & 15);
    if (ctx->dbg_scan_force || !aligned || !mid(na) || !mid(nb)) {
        PEM_TRY(exclusive_scan_i32(ctx, in_a, out_a, na, d_total_a));
        return exclusive_scan_i32(ctx, in_b, out_b, nb, d_total_b);
    }
    PEM_TRY(scan_mid_state(ctx));
    ScanMidArgs a = {};
    a.in[0] = in_a;
    a.out[0] = out_a;
    a.n[0] = na;
    a.total64[0] = reinterpret_cast<long long *>(d_total_a);
    a.nblk[0] = (int)((na + SCAN_MID_ITEMS - 1) / SCAN_MID_ITEMS);
    a.in[1] = in_b;
    a.out[1] = out_b;
    a.n[1] = nb;
    a.total64[1] = reinterpret_cast<long long *>(d_total_b);
    a.nblk[1] = (int)((nb + SCAN_MID_ITEMS - 1) / SCAN_MID_ITEMS);
    PEM_LAUNCH(ctx, scan_mid_kernel, dim3((unsigned)std::max(a.nblk[0], a.nblk[1]), 2), 256, a, ctx->scan_state.as<unsigned long long>(), ctx->d_flags, -1);
    return PEM_OK;
}

pem_status exclusive_scan_i32(pem_ctx *ctx, const int *in, int *out, size_t n, int64_t *d_total64)
{
    if (n == 0) {
        PEM_LAUNCH(ctx, scan_empty_kernel, 1, 1, out, reinterpret_cast<long long *>(d_total64));
        return PEM_OK;
    }
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) {
        set_error("exclusive_scan_i32: unaligned pointer");
        return PEM_E_INVALID;
    }
    const int force = ctx->dbg_scan_force;   // test hook: 1 one block, 2 chained single launch, 3 three launches
    if (force == 2 && n > (size_t)SCAN_MID_ITEMS * SCAN_MID_BLOCKS) {
        set_error("exclusive_scan_i32: the chained scan takes at most %d items", SCAN_MID_ITEMS * SCAN_MID_BLOCKS);
        return PEM_E_INVALID;
    }
    if (force ? force == 1 : n <= 8192) {
        PEM_LAUNCH(ctx, scan_small_kernel, 1, 1024, in, out, n, reinterpret_cast<long long *>(d_total64), ctx->d_flags);
        return PEM_OK;
    }
    if (force ? force == 2 : n <= (size_t)SCAN_MID_ITEMS * SCAN_MID_BLOCKS) {
        PEM_TRY(scan_mid_state(ctx));
        ScanMidArgs a = {};
        a.in[0] = in;
        a.out[0] = out;
        a.n[0] = n;
        a.total64[0] = reinterpret_cast<long long *>(d_total64);
        a.nblk[0] = (int)((n + SCAN_MID_ITEMS - 1) / SCAN_MID_ITEMS);
        PEM_LAUNCH(ctx, scan_mid_kernel, (unsigned)a.nblk[0], 256, a, ctx->scan_state.as<unsigned long long>(), ctx->d_flags, ctx->dbg_scan_stall_ticket);
        return PEM_OK;
    }
    int nblk = (int)((n + SCAN_BLOCK_ITEMS - 1) / SCAN_BLOCK_ITEMS);
    PEM_TRY(ctx->scan_bsum.reserve(sizeof(long long) * ((size_t)nblk + 1)));
    long long *bsum = ctx->scan_bsum.as<long long>();
    PEM_LAUNCH(ctx, scan_reduce_kernel, nblk, SCAN_THREADS, in, n, bsum);
    PEM_LAUNCH(ctx, scan_bsums_kernel, 1, 1024, bsum, nblk, reinterpret_cast<long long *>(d_total64), ctx->d_flags);
    PEM_LAUNCH(ctx, scan_apply_kernel, nblk, SCAN_THREADS, in, out, n, bsum, nblk);
    return PEM_OK;
}

// Two independent arrays of the same length scanned by one set of launches (blockIdx.y picks the array).
struct ScanPair {
    const int *in[2];
    int *out[2];
    long long *bsum[2];
    long long *total64[2];
};

__global__ void __launch_bounds__(SCAN_THREADS) scan2_reduce_kernel(ScanPair sp, size_t n)
{
    __shared__ long long wsum[SCAN_THREADS / 64];
    const int *in = sp.in[blockIdx.y];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK_ITEMS;
    size_t end = base + SCAN_BLOCK_ITEMS < n ? base + SCAN_BLOCK_ITEMS : n;
    long long s = 0;
    if (end - base == (size_t)SCAN_BLOCK_ITEMS && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {   // full block: 16-byte loads
        const int4 *in4 = reinterpret_cast<const int4 *>(in + base);
#pragma unroll
        for (int k = 0; k < SCAN_BLOCK_ITEMS / 4 / SCAN_THREADS; ++k) {
            const int4 q = in4[k * SCAN_THREADS + threadIdx.x];
            s += (long long)q.x + q.y + q.z + q.w;
        }
    } else {
        for (size_t i = base + threadIdx.x; i < end; i += SCAN_THREADS) s += in[i];
    }
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; ++w) t += wsum[w];
        sp.bsum[blockIdx.y][blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(1024) scan2_bsums_kernel(ScanPair sp, int nblk, int *__restrict__ flags)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    long long *bsum = sp.bsum[blockIdx.x];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        int i = base + threadIdx.x;
        long long v = i < nblk ? bsum[i] : 0;
        long long inc = wave_inclusive_scan(v);
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        long long carry = carry_s;
        if (i < nblk) bsum[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        long long t = carry_s;
        bsum[nblk] = t;
        if (sp.total64[blockIdx.x]) *sp.total64[blockIdx.x] = t;
        if (t > 0x7FFFFFFFLL) flags[FLAG_OVERFLOW] = 1;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) scan2_apply_kernel(ScanPair sp, size_t n, int nblk)
{
    __shared__ int wsum[SCAN_THREADS / 64];
    const int *in = sp.in[blockIdx.y];
    int *out = sp.out[blockIdx.y];
    const long long *bsum = sp.bsum[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long carry = bsum[blockIdx.x];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK_ITEMS;
    for (int tile = 0; tile < SCAN_TILES_PER_BLOCK; ++tile, base += SCAN_TILE) {
        if (base >= n) break;
        size_t i0 = base + (size_t)threadIdx.x * 4;
        int v[4];
        if (i0 + 4 <= n) {
            int4 q = *reinterpret_cast<const int4 *>(in + i0);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n) ? in[i0 + k] : 0;
        }
        int tsum = v[0] + v[1] + v[2] + v[3];
        int inc = wave_inclusive_scan(tsum);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0, btotal = 0;
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; ++w) {
            int c = wsum[w];
            if (w < wave) woff += c;
            btotal += c;
        }
        int o0 = (int)carry + woff + inc - tsum, o1 = o0 + v[0], o2 = o1 + v[1], o3 = o2 + v[2];
        if (i0 + 4 <= n) {
            *reinterpret_cast<int4 *>(out + i0) = make_int4(o0, o1, o2, o3);
        } else {
            if (i0 < n) out[i0] = o0;
            if (i0 + 1 < n) out[i0 + 1] = o1;
            if (i0 + 2 < n) out[i0 + 2] = o2;
        }
        carry += btotal;
        __syncthreads();
    }
    if (blockIdx.x == nblk - 1 && threadIdx.x == 0) out[n] = (int)bsum[nblk];
}

pem_status exclusive_scan_i32_pair(pem_ctx *ctx, int *a, int *b, size_t n, int64_t *d_total_a, int64_t *d_total_b)
{
    if (n <= (size_t)SCAN_MID_ITEMS * SCAN_MID_BLOCKS) {   // short: two single-launch scans
        PEM_TRY(exclusive_scan_i32(ctx, a, a, n, d_total_a));
        return exclusive_scan_i32(ctx, b, b, n, d_total_b);
    }
    int nblk = (int)((n + SCAN_BLOCK_ITEMS - 1) / SCAN_BLOCK_ITEMS);
    PEM_TRY(ctx->scan_bsum.reserve(sizeof(long long) * 2 * ((size_t)nblk + 2)));
    ScanPair sp;
    sp.in[0] = a; sp.in[1] = b;
    sp.out[0] = a; sp.out[1] = b;
    sp.bsum[0] = ctx->scan_bsum.as<long long>();
    sp.bsum[1] = sp.bsum[0] + nblk + 2;
    sp.total64[0] = reinterpret_cast<long long *>(d_total_a);
    sp.total64[1] = reinterpret_cast<long long *>(d_total_b);
    PEM_LAUNCH(ctx, scan2_reduce_kernel, dim3(nblk, 2), SCAN_THREADS, sp, n);
    PEM_LAUNCH(ctx, scan2_bsums_kernel, 2, 1024, sp, nblk, ctx->d_flags);
    PEM_LAUNCH(ctx, scan2_apply_kernel, dim3(nblk, 2), SCAN_THREADS, sp, n, nblk);
    return PEM_OK;
}

// ------------------------------------------------------------------------------------------
// LSD radix sort, 8 bits per pass, stable.  Per pass: block histograms -> device scan ->
// scatter.  A block = 4 waves x 16 rounds x 64 keys; ranks come from wave64 ballots
// (match-any on the digit), so there are no atomics in the scatter.
// ------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
#ifndef PEM_RS_ROUNDS
#define PEM_RS_ROUNDS 16
#endif
constexpr int RS_ROUNDS = PEM_RS_ROUNDS;
constexpr int RS_WAVE_ITEMS = RS_ROUNDS * 64;
constexpr int RS_BLOCK_ITEMS = RS_WAVE_ITEMS * RS_WAVES;

__global__ void __launch_bounds__(RS_THREADS) rs_hist_kernel(const uint64_t *__restrict__ keys, size_t n, int shift,
                                                             int *__restrict__ hist, int nblk)
{
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_BLOCK_ITEMS + threadIdx.x;
    // (all of the thread's keys in flight together, then the counting: a load per trip of a run-time loop was a chain of
    // RS_BLOCK_ITEMS / RS_THREADS memory round trips)
    uint64_t k[RS_BLOCK_ITEMS / RS_THREADS];
#pragma unroll
    for (int j = 0; j < RS_BLOCK_ITEMS / RS_THREADS; ++j) {
        const size_t i = base + (size_t)j * RS_THREADS;
        k[j] = i < n ? keys[i] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < RS_BLOCK_ITEMS / RS_THREADS; ++j)
        if (base + (size_t)j * RS_THREADS < n) atomicAdd(&h[(k[j] >> shift) & 255], 1);
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// lanes holding the same digit as this lane (restricted to `valid` lanes)
__device__ __forceinline__ unsigned long long match_digit(unsigned d, unsigned long long valid)
{
    unsigned long long peers = valid;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        unsigned long long m = __ballot((d >> b) & 1);
        peers &= ((d >> b) & 1) ? m : ~m;
    }
    return peers;
}

// The scatter goes through LDS: ranks inside the block first (ballot match per round, per-wave digit counters), the items are
// laid out digit by digit in a block-sized LDS image, and the image is then written out IN ORDER -- consecutive lanes hold
// consecutive items of a digit's run (16 items on average), so a wave's store touches a handful of lines.  Writing every item
// straight to its global position (rounds 1-3) made ~50 partial-line requests per store instruction and ran at the L2's request
// rate, not at any bandwidth: 47 us per pass for 75 MB.
__global__ void __launch_bounds__(RS_THREADS) rs_scatter_kernel(const uint64_t *__restrict__ kin, uint64_t *__restrict__ kout,
                                                                const uint32_t *__restrict__ vin, uint32_t *__restrict__ vout,
                                                                size_t n, int shift, const int *__restrict__ goff, int nblk)
{
    __shared__ int wcnt[RS_WAVES][256];
    __shared__ int lstart[256];                 // where digit d's run starts in the block image
    __shared__ int gdelta[256];                 // global position of the run's first item minus lstart
    __shared__ int wsum[RS_WAVES];
    __shared__ uint64_t skey[RS_BLOCK_ITEMS];
    __shared__ uint32_t sval[RS_BLOCK_ITEMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < RS_WAVES; ++w) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const size_t bbase = (size_t)blockIdx.x * RS_BLOCK_ITEMS;
    const size_t wbase = bbase + (size_t)wave * RS_WAVE_ITEMS;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint64_t k[RS_ROUNDS];
    uint32_t v[RS_ROUNDS];
    unsigned info[RS_ROUNDS];
    // every load of the wave first, all in flight together: with three waves per SIMD (a block is 4 096 items) a load per round,
    // waited for in the round, was a chain of 2 x 16 memory round trips -- 30 of the kernel's 47 us
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t idx = wbase + (size_t)r * 64 + lane;
        k[r] = idx < n ? kin[idx] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t idx = wbase + (size_t)r * 64 + lane;
        v[r] = idx < n ? vin[idx] : 0u;
    }
    // phase 1: per-wave digit counts
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        size_t idx = wbase + (size_t)r * 64 + lane;
        bool valid = idx < n;
        unsigned d = (unsigned)(k[r] >> shift) & 255u;
        unsigned long long vm = __ballot(valid);
        unsigned long long peers = match_digit(d, vm);
        const int below = __popcll(peers & lt), all = __popcll(peers);
        info[r] = (unsigned)below | ((unsigned)all << 8);                   // rank among the round's equal digits | their number
        if (valid && below == 0) wcnt[wave][d] += all;                      // lowest lane of each digit group
    }
    __syncthreads();
    // phase 2: thread d: the block's count of digit d -> exclusive scan over the digits (run starts in the image); the wave
    // counters become running positions in the image
    {
        int c[RS_WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            c[w] = wcnt[w][threadIdx.x];
            tot += c[w];
        }
        int inc = tot;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            int v = __shfl_up(inc, dlt, 64);
            if (lane >= dlt) inc += v;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        int run = before + inc - tot;
        lstart[threadIdx.x] = run;
        gdelta[threadIdx.x] = goff[(size_t)threadIdx.x * nblk + blockIdx.x] - run;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            wcnt[w][threadIdx.x] = run;
            run += c[w];
        }
    }
    __syncthreads();
    // phase 3: into the image (the ranks inside a round were noted in phase 1)
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        size_t idx = wbase + (size_t)r * 64 + lane;
        bool valid = idx < n;
        unsigned d = (unsigned)(k[r] >> shift) & 255u;
        const int below = (int)(info[r] & 255u), all = (int)(info[r] >> 8);
        int start = valid ? wcnt[wave][d] : 0;
        int pos = start + below;
        if (valid && below == 0) wcnt[wave][d] = start + all;
        if (valid) {
            skey[pos] = k[r];
            sval[pos] = v[r];
        }
    }
    __syncthreads();
    // phase 4: the image, in order
    const int nvalid = (int)(bbase + RS_BLOCK_ITEMS <= n ? (size_t)RS_BLOCK_ITEMS : n - bbase);
    for (int i = threadIdx.x; i < nvalid; i += RS_THREADS) {
        const uint64_t key = skey[i];
        const int g = i + gdelta[(unsigned)(key >> shift) & 255u];
        kout[g] = key;
        vout[g] = sval[i];
    }
}

pem_status radix_sort_u64_u32(pem_ctx *ctx, uint64_t *k0, uint64_t *k1, uint32_t *v0, uint32_t *v1, size_t n, int nbits,
                              uint64_t **keys_out, uint32_t **vals_out, int first_bit)
{
    *keys_out = k0;
    *vals_out = v0;
    if (n <= 1 || nbits <= 0) return PEM_OK;
    if (n > 0x7FFFFFFFull) {
        set_error("radix_sort: %zu items exceed int32 positions", n);
        return PEM_E_OVERFLOW;
    }
    int nblk = (int)((n + RS_BLOCK_ITEMS - 1) / RS_BLOCK_ITEMS);
    size_t hcount = (size_t)256 * nblk;
    PEM_TRY(ctx->sort_hist.reserve(sizeof(int) * (hcount + 4)));
    int *hist = ctx->sort_hist.as<int>();
    uint64_t *kin = k0, *kout = k1;
    uint32_t *vin = v0, *vout = v1;
    for (int shift = first_bit; shift < nbits; shift += 8) {
        PEM_LAUNCH(ctx, rs_hist_kernel, nblk, RS_THREADS, kin, n, shift, hist, nblk);
        PEM_TRY(exclusive_scan_i32(ctx, hist, hist, hcount, nullptr));
        PEM_LAUNCH(ctx, rs_scatter_kernel, nblk, RS_THREADS, kin, kout, vin, vout, n, shift, hist, nblk);
        uint64_t *tk = kin; kin = kout; kout = tk;
        uint32_t *tv = vin; vin = vout; vout = tv;
    }
    *keys_out = kin;
    *vals_out = vin;
    return PEM_OK;
}

}  // namespace pem

// test hook (include/pem_spgemm.h): the device scan on a caller's array, with the regime forced and one block stalled
extern "C" pem_status pem_debug_scan_i32(pem_ctx *ctx, const int32_t *in, int64_t n, int regime, int in_place, int stall_ticket, int32_t *out,
                                         int64_t *total)
{
    if (!ctx || n < 0 || (n > 0 && !in) || !out || regime < 0 || regime > 3) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    pem::DevBuf dIn, dOut;
    const size_t sz = sizeof(int) * ((size_t)n + 4);
    PEM_TRY(dIn.reserve(sz));
    PEM_TRY(dOut.reserve(sz));
    PEM_TRY(pem::zero_flags(ctx));
    if (n) PEM_HIP(hipMemcpyAsync(dIn.p, in, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    ctx->dbg_scan_force = regime;
    ctx->dbg_scan_stall_ticket = stall_ticket;
    int *dst = in_place ? dIn.as<int>() : dOut.as<int>();
    pem_status s = pem::exclusive_scan_i32(ctx, dIn.as<int>(), dst, (size_t)n, ctx->d_scalars + 8);
    ctx->dbg_scan_force = 0;
    ctx->dbg_scan_stall_ticket = -1;
    PEM_TRY(s);
    int64_t t = 0;
    PEM_TRY(pem::read_scalars(ctx, ctx->d_scalars + 8, 1, &t));
    int hf[pem::NUM_FLAGS];
    PEM_TRY(pem::read_flags(ctx, hf));
    PEM_HIP(hipMemcpy(out, dst, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToHost));
    if (total) *total = t;
    if (hf[pem::FLAG_INTERNAL]) {
        pem::set_error("device scan: a block waited for an earlier ticket beyond the poll budget");
        return PEM_E_HIP;
    }
    return PEM_OK;
}

__global__ void dbg_noop_kernel(int *flags)
{
    if (flags == nullptr) __builtin_trap();
}
// test hook (include/pem_test.h): a launch the runtime refuses must surface as PEM_E_HIP at the next synchronisation point
extern "C" pem_status pem_debug_refused_launch(pem_ctx *ctx)
{
    if (!ctx) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    PEM_LAUNCH(ctx, dbg_noop_kernel, 1, 2048, ctx->d_flags);
    return PEM_OK;
}

extern "C" const char *pem_last_error(void) { return pem::last_error(); }
extern "C" const char *pem_version(void) { return "pem-spgemm_amd 0.1 (gfx950)"; }

// ivlmerge.hpp — merge of an interval list that is ordered by contig and by start, on the device.
// The sequential rule (sdust's result list, src/sdust/sdust.c:94-98, with dist = 0; `bedtools merge -d dist` of the
// panel scripts): an interval whose start is at most `dist` past the largest finish seen so far in its contig extends
// the current output interval, otherwise it opens a new one.  With K = contig << 32 | finish, "the largest finish so
// far in the contig" is the maximum K over everything before (a plain max-scan: a later contig outranks every finish
// of an earlier one) whenever that maximum is of the same contig.  Five small launches: tile-local max-scan, scan of
// the tile maxima, head flags, add-scan of the flags, emit.
#pragma once
#include "common.hpp"
#include "scan.hpp"

namespace cnivl {
namespace {

constexpr int ST_THREADS = 256, ST_ITEMS = 4, ST_TILE = ST_THREADS * ST_ITEMS;

__device__ __forceinline__ unsigned long long st_shfl_up(unsigned long long v, int d)
{
    return ((unsigned long long)(unsigned)__shfl_up((int)(v >> 32), d) << 32) | (unsigned)__shfl_up((int)v, d);
}

// kprev[j] = max K over the elements of j's tile before j (0: none); tile_max[b] = max K of tile b
__global__ __launch_bounds__(ST_THREADS) void st_local(const cornetto_ivl_t *v, int64_t n, unsigned long long *kprev, unsigned long long *tile_max)
{
    __shared__ unsigned long long wmax[ST_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * ST_TILE + (int64_t)t * ST_ITEMS;
    unsigned long long k[ST_ITEMS], run = 0;
#pragma unroll
    for (int i = 0; i < ST_ITEMS; ++i) {
        k[i] = 0;
        if (j0 + i < n) k[i] = ((unsigned long long)(unsigned)v[j0 + i].ctg << 32) | (unsigned)v[j0 + i].finish;
        run = k[i] > run ? k[i] : run;
    }
    unsigned long long inc = run;                     // inclusive max-scan of the per-thread maxima over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = st_shfl_up(inc, d);
        if (lane >= d && o > inc) inc = o;
    }
    if (lane == 63) wmax[wv] = inc;
    __syncthreads();
    unsigned long long before = st_shfl_up(inc, 1);   // maximum over the earlier threads of the tile
    if (lane == 0) before = 0;
    for (int i = 0; i < wv; ++i) before = wmax[i] > before ? wmax[i] : before;
#pragma unroll
    for (int i = 0; i < ST_ITEMS; ++i) {
        if (j0 + i < n) kprev[j0 + i] = before;
        before = k[i] > before ? k[i] : before;
    }
    if (t == ST_THREADS - 1) tile_max[blockIdx.x] = before;
}

// exclusive max-scan of the tile maxima (one workgroup; a few hundred tiles per million intervals)
__global__ __launch_bounds__(1024) void st_tiles(unsigned long long *tile_max, int64_t n_tiles)
{
    __shared__ unsigned long long wmax[16];
    __shared__ unsigned long long carry_s;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_tiles; base += 1024) {
        const unsigned long long mine = base + t < n_tiles ? tile_max[base + t] : 0ull;
        unsigned long long inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = st_shfl_up(inc, d);
            if (lane >= d && o > inc) inc = o;
        }
        if (lane == 63) wmax[wv] = inc;
        __syncthreads();
        unsigned long long before = st_shfl_up(inc, 1);
        if (lane == 0) before = 0;
        for (int i = 0; i < wv; ++i) before = wmax[i] > before ? wmax[i] : before;
        const unsigned long long carry = carry_s;
        before = carry > before ? carry : before;
        if (base + t < n_tiles) tile_max[base + t] = before;
        __syncthreads();
        if (t == 1023) carry_s = (mine > before ? mine : before);
        __syncthreads();
    }
}

// head[j] = 1 when interval j starts a new output interval; kprev[j] becomes the INCLUSIVE maximum
__global__ void st_heads(const cornetto_ivl_t *v, int64_t n, unsigned long long *kprev, const unsigned long long *tile_before, uint32_t *head, int32_t dist)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned long long tb = tile_before[j / ST_TILE];
    unsigned long long kp = kprev[j];
    kp = tb > kp ? tb : kp;
    const cornetto_ivl_t x = v[j];
    const bool same = kp != 0 && (int32_t)(kp >> 32) == x.ctg;
    head[j] = (!same || (int64_t)x.start > (int64_t)(int32_t)(kp & 0xFFFFFFFFull) + dist) ? 1u : 0u;
    const unsigned long long k = ((unsigned long long)(unsigned)x.ctg << 32) | (unsigned)x.finish;
    kprev[j] = k > kp ? k : kp;
}

__global__ void st_emit(const cornetto_ivl_t *v, int64_t n, const unsigned long long *kinc, const uint32_t *head, const uint32_t *rank, cornetto_ivl_t *out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t hd = head[j];
    const uint32_t g = rank[j] + hd - 1u;             // heads before j, plus j itself when it is one: index of j's group
    if (hd) {
        out[g].ctg = v[j].ctg;
        out[g].start = v[j].start;
    }
    if (j == n - 1 || head[j + 1]) out[g].finish = (int32_t)(kinc[j] & 0xFFFFFFFFull);
}

// ---- the same merge in ONE launch (round 4) --------------------------------------------------------------------------------------
// A tile (1024 intervals) scans its maxima, publishes its own maximum, looks back over the tiles in front of it for the maximum K
// before it (decoupled look-back, as scan.hpp: tiles are numbered in the order they start), decides its heads, publishes their
// number, looks back once more for the rank of its first head, and emits.  The number of intervals is read from the device
// (`d_n`): the caller sizes the grid by an estimate (`n_cap`) and does not have to wait for the count.  States are never
// cleared: a tile's flag carries the epoch of the call.  Per tile 4 x u64: own maximum | inclusive maximum | epoch << 2 | kind
// (1 own, 2 inclusive) | the head count's state (kind << 62 | epoch << 32 | value).
struct StArgs {
    const cornetto_ivl_t *v;
    const unsigned long long *d_n;
    int64_t n_cap;
    int32_t dist;
    unsigned long long *st;
    uint32_t *ticket;
    uint32_t ticket_base, epoch;
    cornetto_ivl_t *out;
    unsigned long long *d_count;
};

__device__ __forceinline__ unsigned long long st_wave_max(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)(v >> 32), d) << 32) | (unsigned)__shfl_xor((int)v, d);
        v = o > v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(ST_THREADS) void st_fused(StArgs A)
{
    __shared__ unsigned long long wmax[ST_THREADS / 64];
    __shared__ uint32_t wcnt[ST_THREADS / 64];
    __shared__ unsigned long long s_in;
    __shared__ uint32_t s_tile, s_base;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) s_tile = atomicAdd(A.ticket, 1u) - A.ticket_base;
    __syncthreads();
    const int64_t tile = s_tile;
    if (tile >= (A.n_cap + ST_TILE - 1) / ST_TILE) return;   // (a ticket base out of step with the counter: never an index)
    const int64_t n = (int64_t)__hip_atomic_load(A.d_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // More rows than the caller sized the grid, the input and the output for: NO tile touches anything (rows beyond n_cap were never
    // gathered, heads would land beyond the output block); *d_count keeps the caller's marker and the caller takes the long way.
    // Every tile has drawn its ticket above, so the host's ticket base stays in step with the device's counter.
    if (n > A.n_cap) return;
    if (tile * ST_TILE >= n) return;                 // (nothing in front of a tile that works waits for one that does not)
    const int64_t j0 = tile * ST_TILE + (int64_t)t * ST_ITEMS;
    cornetto_ivl_t x[ST_ITEMS + 1];
    unsigned long long k[ST_ITEMS], run = 0;
#pragma unroll
    for (int i = 0; i <= ST_ITEMS; ++i) {
        x[i] = cornetto_ivl_t{0, 0, 0};
        if (j0 + i < n) x[i] = A.v[j0 + i];
    }
#pragma unroll
    for (int i = 0; i < ST_ITEMS; ++i) {
        k[i] = j0 + i < n ? ((unsigned long long)(unsigned)x[i].ctg << 32) | (unsigned)x[i].finish : 0ull;
        run = k[i] > run ? k[i] : run;
    }
    unsigned long long inc = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = st_shfl_up(inc, d);
        if (lane >= d && o > inc) inc = o;
    }
    if (lane == 63) wmax[wv] = inc;
    __syncthreads();
    unsigned long long before = st_shfl_up(inc, 1);   // maximum over the earlier threads of the tile
    if (lane == 0) before = 0;
    unsigned long long tmax = 0;
#pragma unroll
    for (int i = 0; i < ST_THREADS / 64; ++i) {
        if (i < wv) before = wmax[i] > before ? wmax[i] : before;
        tmax = wmax[i] > tmax ? wmax[i] : tmax;
    }
    unsigned long long *const st = A.st;
    const unsigned long long tag = (unsigned long long)A.epoch << 2;
    if (wv == 0) {
        unsigned long long M = 0;
        if (tile > 0) {
            if (lane == 0) {
                __hip_atomic_store(&st[4 * tile], tmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __atomic_thread_fence(__ATOMIC_RELEASE);
                __hip_atomic_store(&st[4 * tile + 2], tag | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int64_t pos = tile - 1;; pos -= 64) {
                const int64_t idx = pos - lane;
                unsigned long long f;
                for (;;) {
                    f = idx >= 0 ? __hip_atomic_load(&st[4 * idx + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tag | 2ull);
                    const bool ready = (f >> 2) == (unsigned long long)A.epoch && (f & 3ull) != 0;
                    if (__builtin_amdgcn_ballot_w64(!ready) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                unsigned long long val = 0;
                if (idx >= 0) val = __hip_atomic_load(&st[4 * idx + ((f & 3ull) == 2 ? 1 : 0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long incl = __builtin_amdgcn_ballot_w64((f & 3ull) == 2);
                if (incl) {
                    const int first = __builtin_ctzll(incl);
                    const unsigned long long m = st_wave_max(lane <= first ? val : 0ull);
                    M = m > M ? m : M;
                    break;
                }
                const unsigned long long m = st_wave_max(val);
                M = m > M ? m : M;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&st[4 * tile + 1], M > tmax ? M : tmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store(&st[4 * tile + 2], tag | 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_in = M;
        }
    }
    __syncthreads();
    // heads (st_heads) and the inclusive maxima
    unsigned long long kp = s_in > before ? s_in : before, kinc[ST_ITEMS];
    uint32_t hd[ST_ITEMS + 1], hc = 0;
#pragma unroll
    for (int i = 0; i <= ST_ITEMS; ++i) {
        const bool same = kp != 0 && (int32_t)(kp >> 32) == x[i].ctg;
        hd[i] = (j0 + i < n && (!same || (int64_t)x[i].start > (int64_t)(int32_t)(kp & 0xFFFFFFFFull) + A.dist)) ? 1u : 0u;
        if (i < ST_ITEMS) {
            hc += hd[i];
            kp = k[i] > kp ? k[i] : kp;
            kinc[i] = kp;
        }
    }
    uint32_t pre = cnscan::wave_incl(hc, lane);
    if (lane == 63) wcnt[wv] = pre;
    __syncthreads();
    pre -= hc;
    uint32_t tcnt = 0;
#pragma unroll
    for (int i = 0; i < ST_THREADS / 64; ++i) {
        if (i < wv) pre += wcnt[i];
        tcnt += wcnt[i];
    }
    if (wv == 0) {
        const unsigned long long ctag = (unsigned long long)(A.epoch & 0x3FFFFFFFu) << 32;
        uint32_t excl = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&st[4 * tile + 3], (1ull << 62) | ctag | tcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t pos = tile - 1;; pos -= 64) {
                const int64_t idx = pos - lane;
                unsigned long long c;
                for (;;) {
                    c = idx >= 0 ? __hip_atomic_load(&st[4 * idx + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((2ull << 62) | ctag);
                    const bool ready = (c >> 62) != 0 && (c & (0x3FFFFFFFull << 32)) == ctag;
                    if (__builtin_amdgcn_ballot_w64(!ready) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                const unsigned long long incl = __builtin_amdgcn_ballot_w64((c >> 62) == 2);
                const uint32_t val = (uint32_t)c;
                if (incl) {
                    const int first = __builtin_ctzll(incl);
                    excl += cnscan::wave_sum(lane <= first ? val : 0u);
                    break;
                }
                excl += cnscan::wave_sum(val);
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&st[4 * tile + 3], (2ull << 62) | ctag | (excl + tcnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base = excl;
            if (tile == (n - 1) / ST_TILE && A.d_count) *A.d_count = (unsigned long long)excl + tcnt;
        }
    }
    __syncthreads();
    uint32_t rank = s_base + pre;                     // heads in front of my first interval
#pragma unroll
    for (int i = 0; i < ST_ITEMS; ++i) {
        if (j0 + i >= n) break;
        const uint32_t g = rank + hd[i] - 1u;
        if (hd[i]) {
            A.out[g].ctg = x[i].ctg;
            A.out[g].start = x[i].start;
        }
        if (j0 + i == n - 1 || hd[i + 1]) A.out[g].finish = (int32_t)(kinc[i] & 0xFFFFFFFFull);
        rank += hd[i];
    }
}

// d_in[0 .. *d_n) -> d_out[0 .. *d_count) in one launch; the grid covers n_cap intervals (the caller checks *d_n <= n_cap afterwards:
// with more, *d_count is not written and d_out is incomplete)
static inline int merge_fused(cornetto_accel_t *h, const char *name, const cornetto_ivl_t *d_in, const unsigned long long *d_n, int64_t n_cap, int32_t dist,
                              cornetto_ivl_t *d_out, unsigned long long *d_count)
{
    if (n_cap <= 0) return CORNETTO_OK;
    const int64_t nt = (n_cap + ST_TILE - 1) / ST_TILE;
    const size_t need = 64 + (size_t)nt * 32;
    const bool fresh = h->dev[WS_STITCH].bytes < need;
    uint8_t *ws = (uint8_t *)cn_ws(h, WS_STITCH, need);
    if (!ws) return cn_fail(h, CORNETTO_E_NOMEM, "merge: workspace allocation failed");
    const int rc = [&]() -> int {
        if (fresh) {
            CN_HIP(h, hipMemsetAsync(ws, 0, h->dev[WS_STITCH].bytes, h->stream));
            h->st_tickets = 0;
            h->st_epoch = 0;
        }
        uint32_t epoch = (h->st_epoch + 1) & 0x3FFFFFFFu;
        if (epoch == 0) epoch = 1;
        StArgs A{d_in, d_n, n_cap, dist, reinterpret_cast<unsigned long long *>(ws + 64), reinterpret_cast<uint32_t *>(ws), h->st_tickets, epoch, d_out, d_count};
        CN_LAUNCH(h, name, st_fused<<<dim3((unsigned)nt), dim3(ST_THREADS), 0, h->stream>>>(A));
        h->st_epoch = epoch;                         // (the device's ticket counter advances iff the kernel was queued)
        h->st_tickets += (uint32_t)nt;
        return CORNETTO_OK;
    }();
    if (rc != CORNETTO_OK) h->dev[WS_STITCH].bytes = 0;   // the next call gets new, cleared memory and counters that start again
    return rc;
}

// bytes of device workspace merge() needs for n intervals
static inline size_t ws_bytes(size_t n)
{
    const size_t nt = (n + ST_TILE - 1) / ST_TILE;
    return n * 16 + (nt + 1) * 8 + ((n + 4095) / 4096 + 1) * 4 + 64;
}

// d_in[0..n) -> d_out[0..*d_count): merged list; d_count is a device u64 (also readable after the stream is synchronised)
static inline int merge(cornetto_accel_t *h, const char *name, const cornetto_ivl_t *d_in, int64_t n_in, int32_t dist, uint8_t *ws,
                        cornetto_ivl_t *d_out, unsigned long long *d_count)
{
    if (n_in <= 0) return CORNETTO_OK;
    const size_t n = (size_t)n_in, nt = (n + ST_TILE - 1) / ST_TILE;
    unsigned long long *d_k = (unsigned long long *)ws, *d_tile = d_k + n;
    uint32_t *d_head = (uint32_t *)(d_tile + nt + 1), *d_rank = d_head + n, *d_hp = d_rank + n;
    const unsigned nbn = (unsigned)((n + 255) / 256);
    CN_LAUNCH(h, name, st_local<<<dim3((unsigned)nt), dim3(ST_THREADS), 0, h->stream>>>(d_in, n_in, d_k, d_tile));
    CN_LAUNCH(h, name, st_tiles<<<dim3(1), dim3(1024), 0, h->stream>>>(d_tile, (int64_t)nt));
    CN_LAUNCH(h, name, st_heads<<<dim3(nbn), dim3(256), 0, h->stream>>>(d_in, n_in, d_k, d_tile, d_head, dist));
    CN_TRY(cnscan::exclusive_u32(h, name, d_head, n_in, 1, d_rank, d_hp, d_count));
    CN_LAUNCH(h, name, st_emit<<<dim3(nbn), dim3(256), 0, h->stream>>>(d_in, n_in, d_k, d_head, d_rank, d_out));
    return CORNETTO_OK;
}

}  // namespace
}  // namespace cnivl

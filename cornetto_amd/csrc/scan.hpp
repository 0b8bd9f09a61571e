// scan.hpp — device-wide exclusive prefix sum of 32-bit counters (strided input), used to turn per-tile /
// per-chunk counts into ordered output offsets without a host round trip.  ONE launch (round 4; three before: tile-local scan,
// scan of the tile totals, offset add — at a 395 Mb share a step is made of such launches, ~10 us of stream time each): every
// workgroup scans its 4096 items, publishes its total, and looks back over the states of the tiles in front of it until it meets
// one whose inclusive prefix is known ("decoupled look-back").  A tile's number is the order in which it STARTED (a ticket), so
// everything it waits for is already running.  CORNETTO_SCAN=3 asks for the three launches (A/B, tests).
#pragma once
#include "common.hpp"

namespace cnscan {
namespace {   // internal linkage: every translation unit that includes this gets its own copy

constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 16;                       // per thread
constexpr int SC_TILE = SC_THREADS * SC_ITEMS;     // 4096

__device__ __forceinline__ uint32_t wave_incl(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// out[i] = sum_{k < i, same tile} in[k * stride]; partial[tile] = tile total
__global__ __launch_bounds__(SC_THREADS) void scan_local(const uint32_t *in, int64_t n, int stride, uint32_t *out, uint32_t *partial)
{
    __shared__ uint32_t wtot[SC_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)t * SC_ITEMS;
    uint32_t v[SC_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        const int64_t i = base + k;
        v[k] = i < n ? in[i * stride] : 0u;
        s += v[k];
    }
    const uint32_t inc = wave_incl(s, lane);
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    uint32_t pre = inc - s;
#pragma unroll
    for (int w = 0; w < SC_THREADS / 64; ++w)
        if (w < wv) pre += wtot[w];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        const int64_t i = base + k;
        if (i < n) out[i] = pre;
        pre += v[k];
    }
    if (t == SC_THREADS - 1) partial[blockIdx.x] = pre;
}

// exclusive scan of up to 1024 * per partials by one workgroup, in place; total -> *total
__global__ __launch_bounds__(1024) void scan_partials(uint32_t *partial, int64_t np, unsigned long long *total)
{
    __shared__ uint32_t sh[1024];
    const int t = threadIdx.x;
    const int64_t per = (np + 1023) / 1024;
    const int64_t lo = (int64_t)t * per, hi = lo + per < np ? lo + per : np;
    uint32_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += partial[i];
    sh[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t o = t >= d ? sh[t - d] : 0u;
        __syncthreads();
        sh[t] += o;
        __syncthreads();
    }
    uint32_t run = sh[t] - s;
    for (int64_t i = lo; i < hi; ++i) {
        const uint32_t x = partial[i];
        partial[i] = run;
        run += x;
    }
    if (t == 1023 && total) *total = sh[t];
}

__global__ __launch_bounds__(SC_THREADS) void scan_add(uint32_t *out, int64_t n, const uint32_t *partial)
{
    const uint32_t add = partial[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SC_TILE;
    for (int k = threadIdx.x; k < SC_TILE; k += SC_THREADS) {
        const int64_t i = base + k;
        if (i < n) out[i] += add;
    }
}

// The same for up to four counters that sit side by side in one record (in[i * stride + q], q < m): blockIdx.y = q, three launches for all
// of them instead of three each (a 395 Mb share is made of launches: 12 -> 3 in front of telofind's gather).
struct Outs4 {
    uint32_t *o[4];
};
__global__ __launch_bounds__(SC_THREADS) void scan_local_m(const uint32_t *in, int64_t n, int stride, Outs4 outs, uint32_t *partial, int64_t np)
{
    __shared__ uint32_t wtot[SC_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, q = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)t * SC_ITEMS;
    uint32_t *const out = outs.o[q];
    uint32_t v[SC_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        const int64_t i = base + k;
        v[k] = i < n ? in[i * stride + q] : 0u;
        s += v[k];
    }
    const uint32_t inc = wave_incl(s, lane);
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    uint32_t pre = inc - s;
#pragma unroll
    for (int w = 0; w < SC_THREADS / 64; ++w)
        if (w < wv) pre += wtot[w];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        const int64_t i = base + k;
        if (i < n) out[i] = pre;
        pre += v[k];
    }
    if (t == SC_THREADS - 1) partial[(int64_t)q * np + blockIdx.x] = pre;
}

__global__ __launch_bounds__(1024) void scan_partials_m(uint32_t *partial_all, int64_t np, unsigned long long *total)
{
    __shared__ uint32_t sh[1024];
    const int t = threadIdx.x, q = blockIdx.x;
    uint32_t *const partial = partial_all + (int64_t)q * np;
    const int64_t per = (np + 1023) / 1024;
    const int64_t lo = (int64_t)t * per, hi = lo + per < np ? lo + per : np;
    uint32_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += partial[i];
    sh[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t o = t >= d ? sh[t - d] : 0u;
        __syncthreads();
        sh[t] += o;
        __syncthreads();
    }
    uint32_t run = sh[t] - s;
    for (int64_t i = lo; i < hi; ++i) {
        const uint32_t x = partial[i];
        partial[i] = run;
        run += x;
    }
    if (t == 1023 && total) total[q] = sh[t];
}

__global__ __launch_bounds__(SC_THREADS) void scan_add_m(Outs4 outs, int64_t n, const uint32_t *partial, int64_t np)
{
    const int q = blockIdx.y;
    uint32_t *const out = outs.o[q];
    const uint32_t add = partial[(int64_t)q * np + blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SC_TILE;
    for (int k = threadIdx.x; k < SC_TILE; k += SC_THREADS) {
        const int64_t i = base + k;
        if (i < n) out[i] += add;
    }
}

// ---- single pass -------------------------------------------------------------------------------------------------------------
// state of tile t of counter q: [63:62] 1 = the tile's own total, 2 = the inclusive prefix up to and including it; [61:32] the epoch
// of the call that wrote it; [31:0] the value
struct LbArgs {
    const uint32_t *in;
    int64_t n, np;
    int stride, m;
    Outs4 outs;
    unsigned long long *state;     // [m * np]
    uint32_t *ticket;
    uint32_t ticket_base, epoch;
    unsigned long long *total;     // [m] or null
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__global__ __launch_bounds__(SC_THREADS) void scan_lookback(LbArgs A)
{
    __shared__ uint32_t wtot[SC_THREADS / 64];
    __shared__ uint32_t s_gid, s_excl;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) s_gid = atomicAdd(A.ticket, 1u) - A.ticket_base;
    __syncthreads();
    const int64_t gid = s_gid;
    if (gid >= (int64_t)A.m * A.np) return;          // (a ticket base out of step with the counter — the host resets both on any error: never an index)
    const int q = (int)(gid / A.np);
    const int64_t tile = gid - (int64_t)q * A.np;
    const int64_t base = tile * SC_TILE + (int64_t)t * SC_ITEMS;
    uint32_t *const out = A.outs.o[q];
    uint32_t v[SC_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        const int64_t i = base + k;
        v[k] = i < A.n ? A.in[i * A.stride + q] : 0u;
        s += v[k];
    }
    const uint32_t inc = wave_incl(s, lane);
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    uint32_t pre = inc - s, bt = 0;
#pragma unroll
    for (int w = 0; w < SC_THREADS / 64; ++w) {
        if (w < wv) pre += wtot[w];
        bt += wtot[w];
    }
    if (wv == 0) {
        unsigned long long *const st = A.state + (int64_t)q * A.np;
        const unsigned long long tag = (unsigned long long)(A.epoch & 0x3FFFFFFFu) << 32;
        uint32_t excl = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&st[tile], (1ull << 62) | tag | bt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t pos = tile - 1;; pos -= 64) {
                const int64_t idx = pos - lane;
                unsigned long long x;
                for (;;) {
                    x = idx >= 0 ? __hip_atomic_load(&st[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((2ull << 62) | tag);
                    const bool ready = (x >> 62) != 0 && (x & (0x3FFFFFFFull << 32)) == tag;
                    if (__builtin_amdgcn_ballot_w64(!ready) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                const unsigned long long incl = __builtin_amdgcn_ballot_w64((x >> 62) == 2);
                uint32_t val = (uint32_t)x;
                if (incl) {
                    const int first = __builtin_ctzll(incl);          // the nearest tile whose prefix is complete
                    excl += wave_sum(lane <= first ? val : 0u);
                    break;
                }
                excl += wave_sum(val);
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&st[tile], (2ull << 62) | tag | (excl + bt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
            if (A.total && tile == A.np - 1) A.total[q] = (unsigned long long)excl + bt;
        }
    }
    __syncthreads();
    pre += s_excl;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        const int64_t i = base + k;
        if (i < A.n) out[i] = pre;
        pre += v[k];
    }
}

static inline int scan_mode()
{
    static const int mode = CN_DEV_INT("CORNETTO_SCAN", 1);
    return mode;
}

static inline int lookback_launch(cornetto_accel_t *h, const char *name, const uint32_t *in, int64_t n, int stride, int m, const Outs4 &o,
                                  unsigned long long *d_total)
{
    const int64_t np = (n + SC_TILE - 1) / SC_TILE;
    const size_t need = 64 + (size_t)m * (size_t)np * 8;
    const bool fresh = h->dev[WS_SCAN].bytes < need;
    uint8_t *ws = (uint8_t *)cn_ws(h, WS_SCAN, need);
    if (!ws) return cn_fail(h, CORNETTO_E_NOMEM, "scan: workspace allocation failed");
    const int rc = [&]() -> int {
        if (fresh) {                                 // new memory: no state of any epoch in it, the ticket counter starts again
            CN_HIP(h, hipMemsetAsync(ws, 0, h->dev[WS_SCAN].bytes, h->stream));
            h->scan_tickets = 0;
            h->scan_epoch = 0;
        }
        uint32_t epoch = (h->scan_epoch + 1) & 0x3FFFFFFFu;
        if (epoch == 0) epoch = 1;
        LbArgs A{in, n, np, stride, m, o, reinterpret_cast<unsigned long long *>(ws + 64), reinterpret_cast<uint32_t *>(ws), h->scan_tickets, epoch, d_total};
        CN_LAUNCH(h, name, scan_lookback<<<dim3((unsigned)(m * np)), dim3(SC_THREADS), 0, h->stream>>>(A));
        h->scan_epoch = epoch;                       // the device's ticket counter advances iff the kernel was queued: the host's copy only then
        h->scan_tickets += (uint32_t)(m * np);
        return CORNETTO_OK;
    }();
    if (rc != CORNETTO_OK) h->dev[WS_SCAN].bytes = 0;   // counters possibly out of step: the next call gets new, cleared memory (cn_ws frees the block)
    return rc;
}

// outs[q][i] = exclusive prefix of in[i * stride + q] for q < m (<= 4); d_total (optional): m grand totals (u64 each).
// `partial` must hold m * ceil(n / 4096) u32.
static inline int exclusive_u32_multi(cornetto_accel_t *h, const char *name, const uint32_t *in, int64_t n, int stride, int m, uint32_t *const *outs,
                                      uint32_t *partial, unsigned long long *d_total)
{
    if (n <= 0 || m <= 0) return CORNETTO_OK;
    const int64_t np = (n + SC_TILE - 1) / SC_TILE;
    Outs4 o{};
    for (int q = 0; q < m && q < 4; ++q) o.o[q] = outs[q];
    if (scan_mode() != 3) return lookback_launch(h, name, in, n, stride, m < 4 ? m : 4, o, d_total);
    CN_LAUNCH(h, name, scan_local_m<<<dim3((unsigned)np, (unsigned)m), dim3(SC_THREADS), 0, h->stream>>>(in, n, stride, o, partial, np));
    CN_LAUNCH(h, name, scan_partials_m<<<dim3((unsigned)m), dim3(1024), 0, h->stream>>>(partial, np, d_total));
    CN_LAUNCH(h, name, scan_add_m<<<dim3((unsigned)np, (unsigned)m), dim3(SC_THREADS), 0, h->stream>>>(o, n, partial, np));
    return CORNETTO_OK;
}

// out[i] (u32) = exclusive prefix of in[i*stride]; d_total (optional, device u64) = grand total.
// `partial` must hold ceil(n / 4096) u32.
static inline int exclusive_u32(cornetto_accel_t *h, const char *name, const uint32_t *in, int64_t n, int stride, uint32_t *out,
                                uint32_t *partial, unsigned long long *d_total)
{
    if (n <= 0) return CORNETTO_OK;
    if (scan_mode() != 3) {
        Outs4 o{};
        o.o[0] = out;
        return lookback_launch(h, name, in, n, stride, 1, o, d_total);
    }
    const int64_t np = (n + SC_TILE - 1) / SC_TILE;
    CN_LAUNCH(h, name, scan_local<<<dim3((unsigned)np), dim3(SC_THREADS), 0, h->stream>>>(in, n, stride, out, partial));
    CN_LAUNCH(h, name, scan_partials<<<dim3(1), dim3(1024), 0, h->stream>>>(partial, np, d_total));
    CN_LAUNCH(h, name, scan_add<<<dim3((unsigned)np), dim3(SC_THREADS), 0, h->stream>>>(out, n, partial));
    return CORNETTO_OK;
}

}  // namespace
}  // namespace cnscan

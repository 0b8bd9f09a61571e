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

// cov.hip — the (no)boringbits window stage for gfx950: replaces get_regs() (src/boringbits_main.c:322-378),
// the mean-depth totals of get_depths() (:283-285,:293-294) and the selection loops of print_fun_bits /
// print_boring_bits (:425-445, :463-481).
//
// The reference sums window_size (2500) positions for every window, step window_inc (50): each base is
// read 50 times.  Here each base is read ONCE from HBM (4 B/base: u16 depth + u16 mq):
//
//  cov_blocks   a 256-thread workgroup owns 256 consecutive `inc`-sized blocks of one contig.  The raw
//               u16 data of the tile is staged into LDS with coalesced 16-byte loads, then thread b sums
//               block b out of LDS (odd dword stride for inc = 50: conflict free), also the first
//               r = w % inc elements ("head").  A workgroup-wide inclusive scan (u32, wrapping like the
//               reference's int accumulators) turns block sums into tile-local prefix sums; the tile total
//               is kept exactly in u64 for the mean.
//  cov_tilescan one workgroup: exclusive scan of tile totals (u32 offsets for the prefix, u64 grand totals).
//  cov_windows  one thread per window j: sum = G[a+q-1] - G[a-1] + head[a+q] with G = local prefix +
//               tile offset, q = w / inc; integer division by (end-st) truncating toward zero (:360-361);
//               predicate depth<lo || depth>hi || mq/(double)depth < (double)low_mq (:439) in IEEE double;
//               selected windows are compacted with one atomic reservation per tile and a (base,count)
//               table, so the output order is the reference's print order whatever the dispatch order.
#include <algorithm>
#include <cmath>

#include "common.hpp"
#include "scan.hpp"
#include "tiles.hpp"
#include "ivlmerge.hpp"
#include "internal.hpp"

namespace {

constexpr int CB_THREADS = 256;
#ifndef CN_CB_PARTS
#define CN_CB_PARTS 2
#endif
constexpr int CB_PARTS = CN_CB_PARTS;   // a tile goes through LDS in this many parts
constexpr int CB_MAX_INC_LDS = 128;   // inc <= this: LDS staging path (256 / CB_PARTS x inc x 2 B <= 32 KiB)

struct CbArgs {
    const uint16_t *depth, *mq;
    const int64_t *ctg_off;    // element offsets
    const int32_t *ctg_len;
    const int2 *tiles;         // {ctg, first block of the tile within the contig}
    int32_t inc, r;            // block size, head size (w % inc)
    uint2 *pre;                // [tile*256 + b] = local inclusive prefix {depth, mq} of the block sums
    uint2 *head;               // [tile*256 + b] = {depth, mq} sums of the block's first r positions (r > 0 only: w not a multiple of inc)
    int64_t n_tiles;
    const int4 *tmeta;         // (or NULL) per tile {ctg, first block, contig length, -} {contig offset lo, hi, -, -}: one scalar load instead of three, two of them dependent
    uint2 *tile_tot32;         // wrapping tile totals {depth, mq}
    ulonglong2 *tile_tot64;    // exact tile totals
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// the same scan in six DPP additions (row shifts, then the two row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}
typedef unsigned short cb_us2 __attribute__((ext_vector_type(2)));
// acc + both halves of a dword of two u16 values: one instruction (v_dot2_u32_u16 with a vector of ones)
__device__ __forceinline__ uint32_t add_u16x2(uint32_t acc, uint32_t pair)
{
    const cb_us2 ones = {1, 1};
    return __builtin_amdgcn_udot2(__builtin_bit_cast(cb_us2, pair), ones, acc, false);
}

typedef unsigned int cb_u4 __attribute__((ext_vector_type(4)));

// INC: the block size when it is known at compile time (50, the default step), else 0.  NT (INC known only): tiles per workgroup — the
// loads of ALL of them are issued before the first one is summed: beside another stream's resident kernel one workgroup fits on a CU, and
// with one tile its load and its sum phases alternate with nothing in flight meanwhile
template <bool STAGE, int INC, int NT = 1>
__global__ __launch_bounds__(CB_THREADS) void cov_blocks(CbArgs A)
{
    __builtin_amdgcn_s_setprio(CN_STREAM_PRIO);   // short streaming kernel: issue ahead of a long compute-bound kernel of another stream

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t wsum[2][CB_THREADS / 64];
    __shared__ unsigned long long wsum64[2][CB_THREADS / 64];

    const int t = threadIdx.x;
    const int inc = INC ? INC : A.inc, r = A.r;
    constexpr int PBc = CB_THREADS / CB_PARTS;
    constexpr int NVIc = INC ? (PBc * INC / 8 + CB_THREADS - 1) / CB_THREADS : 4;
    cb_u4 pre_all[NT][2][CB_PARTS][NVIc];
    if (STAGE && INC) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int64_t tix = (int64_t)blockIdx.x * NT + ti;
            if (tix >= A.n_tiles) continue;
            int2 tl;
            int ln;
            int64_t of;
            if (A.tmeta) {
                const int4 m0 = A.tmeta[2 * tix], m1 = A.tmeta[2 * tix + 1];
                tl = make_int2(m0.x, m0.y); ln = m0.z; of = (int64_t)(((unsigned long long)(uint32_t)m1.y << 32) | (uint32_t)m1.x);
            } else {
                tl = A.tiles[tix]; ln = A.ctg_len[tl.x]; of = A.ctg_off[tl.x];
            }
            const int64_t e0l = (int64_t)tl.y * inc;
            const int nvec = PBc * inc / 8;
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int part = 0; part < CB_PARTS; ++part) {
                    const int64_t pe0 = e0l + (int64_t)part * PBc * inc;
                    const int64_t left = ((int64_t)ln - pe0) * 2;
                    const uint32_t nrec = left <= 0 ? 0u : (uint32_t)(left < (int64_t)nvec * 16 ? (left + 15) & ~15LL : (int64_t)nvec * 16);
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((which ? A.mq : A.depth) + of + pe0), 0, (int)nrec, 0x00020000);
#pragma unroll
                    for (int k = 0; k < NVIc; ++k) pre_all[ti][which][part][k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (t + k * CB_THREADS) * 16, 0, 0);
                }
        }
    }
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int64_t tix = (int64_t)blockIdx.x * NT + ti;
    if (tix >= A.n_tiles) break;
    if (ti) __syncthreads();                            // (the tile before is through with the staging buffer and the wave sums)
    int2 tile;
    int len;
    int64_t off;
    if (A.tmeta) {
        const int4 m0 = A.tmeta[2 * tix], m1 = A.tmeta[2 * tix + 1];
        tile = make_int2(m0.x, m0.y); len = m0.z; off = (int64_t)(((unsigned long long)(uint32_t)m1.y << 32) | (uint32_t)m1.x);
    } else {
        tile = A.tiles[tix]; len = A.ctg_len[tile.x]; off = A.ctg_off[tile.x];
    }
    const int64_t e0 = (int64_t)tile.y * inc;           // first element of the tile within the contig
    const int64_t p0 = e0 + (int64_t)t * inc;           // first element of my block

    uint32_t fd = 0, fq = 0, hd = 0, hq = 0;
    unsigned long long xd = 0, xq = 0;                  // exact block sums
    if (STAGE) {
        // One array at a time, and of each array one half of the tile (CB_THREADS / CB_PARTS blocks) at a time, through the
        // same LDS buffer (256 / CB_PARTS * inc * 2 bytes = 12.8 KB for inc = 50): several workgroups fit into the
        // LDS that another stream's resident kernel leaves free on a CU, and their phases overlap.
        uint16_t *sv = reinterpret_cast<uint16_t *>(smem);
        constexpr int PB = CB_THREADS / CB_PARTS;       // blocks per part
        const int nvec = PB * inc / 8;                  // 16-byte vectors per part (PB * inc * 2 bytes: multiple of 16)
        const int base = (t % PB) * inc;
        const int nval = p0 >= len ? 0 : (int)(len - p0 < inc ? len - p0 : inc);   // elements of my block inside the contig
        const int nh = r < nval ? r : nval;
        // INC known (<= 64: at most 4 vectors per thread and part): the loads of all four phases (2 arrays x 2 parts: the whole
        // tile, 4 x 12.8 KB at INC = 50) are issued before the first LDS store — beside another stream's resident kernel only
        // one or two of these workgroups fit on a CU, and what bounds the kernel then is the bytes it keeps in flight.
        constexpr int NVI = NVIc;
        auto &pre = pre_all[ti];
#pragma unroll
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int part = 0; part < CB_PARTS; ++part) {
                const int64_t pe0 = e0 + (int64_t)part * PB * inc;      // first element of this part within the contig
                // The part as a raw buffer that ends with the contig (rounded up to a whole vector, as far as a plain load of
                // the vector holding the last element reads): loads past it return zeros, so no load needs a bounds test.
                const int64_t left = ((int64_t)len - pe0) * 2;
                const uint32_t nrec = left <= 0 ? 0u : (uint32_t)(left < (int64_t)nvec * 16 ? (left + 15) & ~15LL : (int64_t)nvec * 16);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((which ? A.mq : A.depth) + off + pe0), 0, (int)nrec, 0x00020000);
                if (which | part) __syncthreads();      // everyone is done reading the previous part
                // all loads of a thread in flight before the first LDS store
                constexpr int NV = 4;
                if (INC) {
#pragma unroll
                    for (int k = 0; k < NVI; ++k) {
                        const int v = t + k * CB_THREADS;
                        if (v < nvec) reinterpret_cast<cb_u4 *>(sv)[v] = pre[which][part][k];
                    }
                } else
                for (int v0 = t; v0 < nvec; v0 += NV * CB_THREADS) {
                    cb_u4 a[NV];
#pragma unroll
                    for (int k = 0; k < NV; ++k) a[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (v0 + k * CB_THREADS) * 16, 0, 0);
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        const int v = v0 + k * CB_THREADS;
                        if (v < nvec) reinterpret_cast<cb_u4 *>(sv)[v] = a[k];
                    }
                }
                __syncthreads();
                if (t / PB == part) {
                    uint32_t f = 0, hh = 0;
                    if (INC && (INC & 1) == 0 && nval == inc) {   // whole dwords, unrolled
                        const uint32_t *wd = reinterpret_cast<const uint32_t *>(sv + base);
#pragma unroll
                        for (int i = 0; i < (INC ? INC : 2) / 2; ++i) f = add_u16x2(f, wd[i]);
                    } else if (nval == inc && (inc & 1) == 0) {
                        const uint32_t *wd = reinterpret_cast<const uint32_t *>(sv + base);
                        for (int i = 0; i < inc / 2; ++i) f = add_u16x2(f, wd[i]);
                    } else {
#pragma clang loop vectorize(disable) unroll(disable)
                        for (int i = 0; i < nval; ++i) f += sv[base + i];
                    }
#pragma clang loop vectorize(disable) unroll(disable)
                    for (int i = 0; i < nh; ++i) hh += sv[base + i];
                    if (which) { fq = f; hq = hh; } else { fd = f; hd = hh; }
                }
            }
        }
        xd = fd;
        xq = fq;
    } else {
        // general path for large increments: direct loads (each thread walks its own block)
        const uint16_t *d = A.depth + off, *q = A.mq + off;
        for (int64_t p = p0; p < p0 + inc && p < len; ++p) {
            const uint32_t a = d[p], b = q[p];
            xd += a;
            xq += b;
            if (p - p0 < r) {
                hd += a;
                hq += b;
            }
        }
        fd = (uint32_t)xd;
        fq = (uint32_t)xq;
    }

    // workgroup inclusive scan of (fd, fq), wrapping; exact totals in u64.  On the staging path (inc <= 128) a tile total is
    // below 256 x 128 x 65535 < 2^32: the u32 scan does not wrap inside a tile and its last value is the exact total.
    const int lane = t & 63, wv = t >> 6;
    uint32_t sdv, sqv;
    unsigned long long td = 0, tq = 0;
    if (STAGE) {
        sdv = wave_incl_scan_dpp(fd);
        sqv = wave_incl_scan_dpp(fq);
    } else {
        sdv = wave_incl_scan(fd, lane);
        sqv = wave_incl_scan(fq, lane);
        td = xd;
        tq = xq;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            td += ((unsigned long long)__shfl_xor((unsigned)(td >> 32), d) << 32) | __shfl_xor((unsigned)td, d);
            tq += ((unsigned long long)__shfl_xor((unsigned)(tq >> 32), d) << 32) | __shfl_xor((unsigned)tq, d);
        }
    }
    if (lane == 63) {
        wsum[0][wv] = sdv;
        wsum[1][wv] = sqv;
    }
    if (!STAGE && lane == 0) {
        wsum64[0][wv] = td;
        wsum64[1][wv] = tq;
    }
    __syncthreads();
    uint32_t pd = 0, pq = 0;
#pragma unroll
    for (int i = 0; i < CB_THREADS / 64; ++i)
        if (i < wv) {
            pd += wsum[0][i];
            pq += wsum[1][i];
        }
    A.pre[(size_t)tix * CB_THREADS + t] = make_uint2(sdv + pd, sqv + pq);
    if (A.r) A.head[(size_t)tix * CB_THREADS + t] = make_uint2(hd, hq);
    if (t == CB_THREADS - 1) {
        A.tile_tot32[tix] = make_uint2(sdv + pd, sqv + pq);
        ulonglong2 x;
        if (STAGE) {
            x.x = sdv + pd;
            x.y = sqv + pq;
        } else {
            x.x = wsum64[0][0] + wsum64[0][1] + wsum64[0][2] + wsum64[0][3];
            x.y = wsum64[1][0] + wsum64[1][1] + wsum64[1][2] + wsum64[1][3];
        }
        A.tile_tot64[tix] = x;
    }
  }
}

// exact grand totals of depth / mq for the mean: one 64-bit atomic pair per workgroup
__global__ __launch_bounds__(256) void cov_total64(const ulonglong2 *tot64, int64_t n, unsigned long long *grand /* [2] */)
{
    __shared__ unsigned long long sa[4], sb[4];
    unsigned long long a = 0, b = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const ulonglong2 v = tot64[i];
        a += v.x;
        b += v.y;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += ((unsigned long long)__shfl_xor((unsigned)(a >> 32), d) << 32) | __shfl_xor((unsigned)a, d);
        b += ((unsigned long long)__shfl_xor((unsigned)(b >> 32), d) << 32) | __shfl_xor((unsigned)b, d);
    }
    if ((threadIdx.x & 63) == 0) {
        sa[threadIdx.x >> 6] = a;
        sb[threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&grand[0], sa[0] + sa[1] + sa[2] + sa[3]);
        atomicAdd(&grand[1], sb[0] + sb[1] + sb[2] + sb[3]);
    }
}

struct CwArgs {
    const uint2 *pre, *head;   // what cov_blocks left (8 B per block each: a window reads two prefixes, the head only when w % inc != 0)
    const uint32_t *toff_d, *toff_q;   // exclusive prefix of the tile totals (wrapping)
    const int64_t *blk_off;    // first global block of each contig (multiple of 256)
    const int32_t *ctg_len;
    const int32_t *n_reg;      // windows per contig
    const int2 *tiles;         // {ctg, first window}
    int32_t w, inc, q, r;
    uint32_t w_magic, w_shift;   // floor(n / w) = umulhi(n, w_magic) >> w_shift for 0 <= n < 2^31 (w >= 2; w_magic = 0: divide)
    // selection
    int32_t mode;              // 0 = all windows of one contig into regs; 1 = fun; 2 = boring
    int32_t lo, hi, edge, min_len;
    double low_mq;
    cornetto_reg_t *regs;
    cornetto_regrec_t *sel;    // selected windows, each tile's at the front of the tile's own 256 entries of the raw array
    cornetto_regpk_t *pk;      // ... or, if not NULL, their packed 8-byte form (whenever the means fit: w <= 32768; cov_order expands them)
    uint2 *tile_res;           // per tile {base, count}
};

__device__ __forceinline__ uint2 cw_prefix(const CwArgs &A, int64_t x)   // inclusive global prefix at block x
{
    const uint2 b = A.pre[x];
    return make_uint2(b.x + A.toff_d[x >> 8], b.y + A.toff_q[x >> 8]);
}

__global__ __launch_bounds__(256) void cov_windows(CwArgs A)
{
    __builtin_amdgcn_s_setprio(CN_STREAM_PRIO);   // short streaming kernel: issue ahead of a long compute-bound kernel of another stream

    __shared__ uint32_t wcnt[4];
    const int t = threadIdx.x;
    const int2 tile = A.tiles[blockIdx.x];
    const int ctg = tile.x;
    const int len = A.ctg_len[ctg];
    const int j = tile.y + t;
    bool valid = j < A.n_reg[ctg];
    int st = 0, end = 0, depth = 0, mq = 0;
    bool sel = false;
    if (valid) {
        st = j * A.inc;                                  // :347
        end = st + A.w;
        if (end > len) end = len;                        // :349-351
        const int64_t a = A.blk_off[ctg] + j;
        // blocks of a contig start at a multiple of 256 = a tile boundary of the prefix, and the global
        // prefix is continuous across contigs, so G[a+q-1] - G[a-1] is the sum over exactly [a, a+q).
        // q == 0 (inc > w: sparse windows, each inside its own block): the window is the head of block a alone
        uint32_t sd = 0, sq = 0;
        if (A.q > 0) {
            const uint2 hi = cw_prefix(A, a + A.q - 1);
            uint2 lo = make_uint2(0, 0);
            if (a > 0) lo = cw_prefix(A, a - 1);
            sd = hi.x - lo.x;
            sq = hi.y - lo.y;
        }
        if (A.r) {
            const uint2 hb = A.head[a + A.q];
            sd += hb.x;
            sq += hb.y;
        }
        // :360-361 (positions >= len contributed zero).  Nearly every window is w long and its sums are below 2^31: the division is a
        // multiplication by a reciprocal the host made for w (exact for 0 <= n < 2^31); anything else divides
        if (end - st == A.w && A.w_magic && (int32_t)(sd | sq) >= 0) {
            depth = (int32_t)(__umulhi(sd, A.w_magic) >> A.w_shift);
            mq = (int32_t)(__umulhi(sq, A.w_magic) >> A.w_shift);
        } else {
            depth = (int32_t)sd / (end - st);
            mq = (int32_t)sq / (end - st);
        }
        if (A.mode == 0) {
            A.regs[j] = cornetto_reg_t{st, end, depth, mq};
        } else {
            // :439.  mq >= depth > 0 gives a quotient >= 1, which is not below a threshold <= 1: no division for those (the test is
            // the reference's IEEE double division wherever it can decide)
            bool fun = depth < A.lo || depth > A.hi;
            if (!fun && !(A.low_mq <= 1.0 && depth > 0 && mq >= depth)) fun = ((double)mq / (double)depth) < A.low_mq;
            if (A.mode == 1) sel = len >= A.min_len && fun;                                               // :428 else-branch
            else sel = len > A.min_len && st > A.edge && end < len - A.edge && !fun;                      // :467,:473-474
        }
    }
    if (A.mode == 0) return;
    // ordered compaction inside the tile: ballot + popcount, one reservation per tile
    const unsigned long long bal = __ballot(sel);
    const int lane = t & 63, wv = t >> 6;
    if (lane == 0) wcnt[wv] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < wv) pre += wcnt[i];
        tot += wcnt[i];
    }
    // a tile's selected windows go to the tile's own 256 entries of the raw array (no reservation: one counter for 250 000 tiles was a third of
    // the kernel's time — 2 ns per atomic on one address); the ordering pass packs the segments by the scan of their counts
    if (t == 0) A.tile_res[blockIdx.x] = make_uint2(blockIdx.x * 256u, tot);
    if (sel) {
        const size_t idx = (size_t)blockIdx.x * 256u + pre + __popcll(bal & ((1ull << lane) - 1ull));
        if (A.pk) A.pk[idx] = cornetto_regpk_t{st, (uint16_t)depth, (uint16_t)mq};   // means of uint16 values: they fit
        else A.sel[idx] = cornetto_regrec_t{ctg, st, end, depth, mq};
    }
}

// packed selection: the first record of every contig = the ordered offset of its first tile (a contig without tiles: that of the next contig
// that has one, or the total)
__global__ void cov_ctg_first(const uint32_t *ooff, const int32_t *first_tile, int32_t n_ctg, uint32_t n_tiles, const unsigned long long *total, uint32_t *out)
{
    const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_ctg) return;
    const uint32_t t = (uint32_t)first_tile[i];
    out[i] = t < n_tiles ? ooff[t] : (uint32_t)*total;
}

// tile segments (reservation order) -> (contig, window) order: one wavefront per tile; INTS = 4-byte words per record
template <int INTS>
__global__ __launch_bounds__(256) void cov_order(const int32_t *raw, const uint2 *tres, const uint32_t *ooff, int64_t n_tiles, int32_t *dst, const unsigned long long *total,
                                                 uint32_t cap)
{
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    if (*total > cap) return;   // the selection outgrew the block (the host reruns at the exact size): segments beyond it were never written
    const uint2 r = tres[t];
    const int32_t *src = raw + (size_t)r.x * INTS;
    int32_t *d = dst + (size_t)ooff[t] * INTS;
    const uint32_t nint = r.y * (uint32_t)INTS;
    for (uint32_t i = threadIdx.x & 63; i < nint; i += 64) d[i] = src[i];
}

// the same with packed segments expanded to full records: the contig is the tile's, end = min(st + w, length) (:348-351)
__global__ __launch_bounds__(256) void cov_order_expand(const cornetto_regpk_t *raw, const uint2 *tres, const uint32_t *ooff, int64_t n_tiles, cornetto_regrec_t *dst,
                                                        const unsigned long long *total, uint32_t cap, const int2 *tiles, const int32_t *ctg_len, int32_t w)
{
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    if (*total > cap) return;
    const uint2 r = tres[t];
    const int32_t ctg = tiles[t].x, len = ctg_len[ctg];
    const cornetto_regpk_t *src = raw + r.x;
    cornetto_regrec_t *d = dst + ooff[t];
    for (uint32_t i = threadIdx.x & 63; i < r.y; i += 64) {
        const cornetto_regpk_t p = src[i];
        const int64_t e = (int64_t)p.st + w;
        d[i] = cornetto_regrec_t{ctg, p.st, e > len ? len : (int32_t)e, (int32_t)p.depth, (int32_t)p.mq_depth};
    }
}

int32_t n_reg_host(int32_t length, int32_t w, int32_t inc)
{
    int32_t n = (length - w + inc - 1) / inc + 1;   // src/boringbits_main.c:338, C truncation
    return n < 1 ? 1 : n;                           // :339
}

// The two asserts of get_regs() that can fire (src/boringbits_main.c:353 inside the loop, :368 behind it; :369 repeats :353 for the last window).
// Every window but the last starts in front of the last one, so only the last window decides: 0 = the reference computes, 353 / 368 = the
// line of the assert that ends it.  With 1 <= inc <= w and length >= 1 neither fires; with inc > w the windows are sparse and the last one
// must still reach the contig's end.
int32_t regs_assert_host(int32_t length, int32_t w, int32_t inc)
{
    const int64_t st = (int64_t)(n_reg_host(length, w, inc) - 1) * inc;
    int64_t end = st + w;
    if (end > length) end = length;
    if (!(st < end)) return 353;
    return end == length ? 0 : 368;
}

}  // namespace

extern "C" {

int32_t cornetto_n_reg(int32_t length, int32_t window_size, int32_t window_inc)
{
    if (window_inc <= 0) return 0;
    return n_reg_host(length, window_size, window_inc);
}

int32_t cornetto_regs_assert(int32_t length, int32_t window_size, int32_t window_inc)
{
    if (window_inc <= 0) return 353;
    return regs_assert_host(length, window_size, window_inc);
}

int32_t cornetto_cov_threshold(float factor, int32_t mean)
{
    return (int32_t)round(factor * mean);   // float product promoted to double by round(): src/boringbits_main.c:518-519
}

int cornetto_cov_prepare(cornetto_accel_t *h, cornetto_cov_t *c, int32_t w, int32_t inc, uint64_t sums[3])
{
    if (!h || !c || !sums) return cn_fail(h, CORNETTO_E_ARG, "cov_prepare: bad argument");
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    const int rc = cn_cov_prepare_impl(h, c, w, inc, sums);
    if (rc == CORNETTO_OK) cn_timing_end(h);
    return rc;
}

}  // extern "C"

// tiles.hpp: window tile j of contig i
struct CwTileFill {
    int2 *out;
    __device__ void operator()(int64_t t, int ci, int64_t j) const { out[t] = make_int2(ci, (int)(j * 256)); }
};
// tiles.hpp: block tile j of contig i, and the same with the contig's length and offset in one 32-byte record (CbArgs::tmeta)
struct CbTileFill {
    int2 *tiles;
    int4 *tmeta;
    const int32_t *len;
    const int64_t *off;
    __device__ void operator()(int64_t t, int ci, int64_t j) const
    {
        const int b = (int)(j * CB_THREADS);
        const unsigned long long o = (unsigned long long)off[ci];
        tiles[t] = make_int2(ci, b);
        tmeta[2 * t] = make_int4(ci, b, len[ci], 0);
        tmeta[2 * t + 1] = make_int4((int)(uint32_t)o, (int)(uint32_t)(o >> 32), 0, 0);
    }
};

int cn_cov_prepare_impl(cornetto_accel_t *h, cornetto_cov_t *c, int32_t w, int32_t inc, uint64_t sums[3])
{
    if (inc < 1) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "cov_prepare: needs window_inc >= 1 (got -i %d: the reference divides by it)", inc);
    // get_regs() runs over every contig before anything is printed, and its asserts (:353, :368) end the process: the first contig, in input
    // order, whose last window is empty or stops short of the contig's end (possible only with w < inc or w < 1) decides
    for (int32_t i = 0; i < c->n; ++i) {
        const int32_t line = c->len[i] < 1 ? 0 : regs_assert_host(c->len[i], w, inc);
        if (line)
            return cn_fail(h, CORNETTO_E_ASSERT, "src/boringbits_main.c:%d: get_regs: Assertion `%s' failed. (contig %d of length %d, -w %d -i %d)", line,
                           line == 353 ? "st<end" : "end == length", i, c->len[i], w, inc);
    }
    const int32_t q = w / inc, r = w % inc;
    sums[0] = sums[1] = 0;
    sums[2] = (uint64_t)c->total;
    CN_TRACE("cov_prepare: enter");
    if (c->w != w || c->inc != inc || !c->d_blk) {
        // (re)build the decomposition for these window sizes; cached for later calls.  A failure anywhere inside (an allocation, a copy) leaves
        // the cache invalid — the next call with the same sizes rebuilds instead of indexing with offsets that were never uploaded.
        struct Invalidate {
            cornetto_cov_t *c;
            bool keep = false;
            ~Invalidate() { if (!keep) c->w = c->inc = -1; }
        } built{c};
        c->w = w;
        c->inc = inc;
        c->cw_mode = -1;
        c->blk_off.assign(c->n + 1, 0);
        c->n_reg.assign(c->n, 0);
        for (int32_t i = 0; i < c->n; ++i) {
            if (c->len[i] < 1) return cn_fail(h, CORNETTO_E_ARG, "cov_prepare: contig %d is empty (a bedgraph cannot produce that)", i);
            c->n_reg[i] = n_reg_host(c->len[i], w, inc);
            c->blk_off[i + 1] = c->blk_off[i] + cn_align_up((int64_t)c->n_reg[i] + q + 1, CB_THREADS);
        }
        c->n_blk = c->blk_off[c->n];
        // the block tiles on the device (tiles.hpp): tile j of contig i = its blocks j CB_THREADS ..., with what cov_blocks needs of the contig
        const int64_t ntl = cntiles::prefix(h, c->cb_pref, c->n, [&](int32_t i) { return (c->blk_off[i + 1] - c->blk_off[i]) / CB_THREADS; });
        if (ntl < 0) return cn_fail(h, CORNETTO_E_NOMEM, "cov_prepare: device allocation failed");
        c->n_cb_tiles = ntl;
        if (c->d_blk) { (void)hipFree(c->d_blk); c->d_blk = nullptr; }
        if (c->d_blk_off) { (void)hipFree(c->d_blk_off); c->d_blk_off = nullptr; }
        if (c->d_cb_tiles) { (void)hipFree(c->d_cb_tiles); c->d_cb_tiles = nullptr; }
        if (c->d_cb_tmeta) { (void)hipFree(c->d_cb_tmeta); c->d_cb_tmeta = nullptr; }
        if (c->d_n_reg) { (void)hipFree(c->d_n_reg); c->d_n_reg = nullptr; }
        const size_t nt = (size_t)ntl;
        if (nt == 0) { built.keep = true; return CORNETTO_OK; }
        // prefixes [n_blk] uint2, heads [n_blk] uint2, then two arrays of tile offsets [nt] u32 each
        if (cn_obj_malloc(h, (void **)&c->d_blk, (size_t)c->n_blk * sizeof(uint4) + 2 * nt * sizeof(uint32_t)) != hipSuccess ||
            cn_obj_malloc(h, (void **)&c->d_blk_off, (size_t)(c->n + 1) * 8) != hipSuccess ||
            cn_obj_malloc(h, (void **)&c->d_cb_tiles, nt * sizeof(int2)) != hipSuccess ||
            cn_obj_malloc(h, (void **)&c->d_cb_tmeta, 2 * nt * sizeof(int4)) != hipSuccess ||
            cn_obj_malloc(h, (void **)&c->d_n_reg, (size_t)c->n * 4) != hipSuccess)
            return cn_fail(h, CORNETTO_E_NOMEM, "cov_prepare: device allocation failed");
        cntiles::fill<<<dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->stream>>>(c->cb_pref.dev, c->n, ntl, CbTileFill{c->d_cb_tiles, c->d_cb_tmeta, c->d_len, c->d_off});
        CN_HIP(h, hipGetLastError());
        CN_HIP(h, hipMemcpyAsync(c->d_blk_off, c->blk_off.data(), (size_t)(c->n + 1) * 8, hipMemcpyHostToDevice, h->stream));   // (members of the object: alive
        CN_HIP(h, hipMemcpyAsync(c->d_n_reg, c->n_reg.data(), (size_t)c->n * 4, hipMemcpyHostToDevice, h->stream));            //  while the copies run)
        built.keep = true;
    }
    CN_TRACE("cov_prepare: tables");
    const size_t nt = (size_t)c->n_cb_tiles;
    if (nt == 0) return CORNETTO_OK;
    uint2 *d_t32 = (uint2 *)cn_ws(h, WS_CB_T32, nt * sizeof(uint2) + ((nt + 4095) / 4096 + 1) * 4 * 2);
    ulonglong2 *d_t64 = (ulonglong2 *)cn_ws(h, WS_CB_T64, nt * sizeof(ulonglong2));
    unsigned long long *d_grand = (unsigned long long *)cn_ws(h, WS_CB_GRAND, 16);
    unsigned long long *p_grand = (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
    if (!d_t32 || !d_t64 || !d_grand || !p_grand) return cn_fail(h, CORNETTO_E_NOMEM, "cov_prepare: workspace allocation failed");
    uint32_t *d_part = reinterpret_cast<uint32_t *>(d_t32 + nt);
    uint2 *d_pre = reinterpret_cast<uint2 *>(c->d_blk), *d_head = d_pre + c->n_blk;
    uint32_t *d_toff_d = reinterpret_cast<uint32_t *>(d_head + c->n_blk), *d_toff_q = d_toff_d + nt;
    CN_TRACE("cov_prepare: workspaces");
    CN_HIP(h, hipMemsetAsync(d_grand, 0, 16, h->stream));
    static const int cb_tmeta = CN_DEV_INT("CORNETTO_COV_TMETA", 1);
    CbArgs A{c->d_depth, c->d_mq, c->d_off, c->d_len, c->d_cb_tiles, inc, r, d_pre, d_head, (int64_t)nt, cb_tmeta ? c->d_cb_tmeta : nullptr, d_t32, d_t64};
    if (inc <= CB_MAX_INC_LDS) {
        const size_t lds = (size_t)CB_THREADS / CB_PARTS * inc * sizeof(uint16_t);
        static_assert(CB_THREADS / CB_PARTS * 50 / 8 <= 4 * CB_THREADS, "cov_blocks<true, INC>: at most 4 vectors per thread and part");
        if (A.inc == 50) {                               // the default step (-i 50)
            static const int cb_nt = CN_DEV_INT("CORNETTO_COV_TILES", 1);   // (2: two tiles per workgroup, all loads up front — 138 registers: does not fit beside the resident sdust waves, 8.4 instead of 3.9 ms in the step)
            if (cb_nt == 2) {
                CN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&cov_blocks<true, 50, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                CN_LAUNCH(h, "cov_blocks", cov_blocks<true, 50, 2><<<dim3((unsigned)((nt + 1) / 2)), dim3(CB_THREADS), lds, h->stream>>>(A));
            } else {
                CN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&cov_blocks<true, 50>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                CN_LAUNCH(h, "cov_blocks", cov_blocks<true, 50><<<dim3((unsigned)nt), dim3(CB_THREADS), lds, h->stream>>>(A));
            }
        } else {
            CN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&cov_blocks<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CN_LAUNCH(h, "cov_blocks", cov_blocks<true, 0><<<dim3((unsigned)nt), dim3(CB_THREADS), lds, h->stream>>>(A));
        }
    } else {
        CN_LAUNCH(h, "cov_blocks", cov_blocks<false, 0><<<dim3((unsigned)nt), dim3(CB_THREADS), 0, h->stream>>>(A));
    }
    {
        uint32_t *const outs[2] = {d_toff_d, d_toff_q};
        CN_TRY(cnscan::exclusive_u32_multi(h, "cov_tilescan", reinterpret_cast<const uint32_t *>(d_t32), (int64_t)nt, 2, 2, outs, d_part, nullptr));
    }
    const unsigned nb64 = (unsigned)std::min<size_t>(1024, (nt + 255) / 256);
    CN_LAUNCH(h, "cov_total64", cov_total64<<<dim3(nb64), dim3(256), 0, h->stream>>>(d_t64, (int64_t)nt, d_grand));
    CN_HIP(h, hipMemcpyAsync(p_grand, d_grand, 16, hipMemcpyDeviceToHost, h->stream));
    CN_TRACE("cov_prepare: queued");
    CN_HIP(h, hipStreamSynchronize(h->stream));
    CN_TRACE("cov_prepare: done");
    // (a coverage read from bedgraph text with negative depth values: the totals take the values, the arrays their uint16 — common.hpp sum_corr; the
    // sums are then two's complement numbers, negative where the negative values outweigh the others)
    sums[0] = p_grand[0] - c->sum_corr[0];
    sums[1] = p_grand[1] - c->sum_corr[1];
    c->sums[0] = sums[0]; c->sums[1] = sums[1]; c->sums[2] = (uint64_t)c->total;
    return CORNETTO_OK;
}

extern "C" {

// mode 0: all windows of contig only_ctg into regs_host; mode 1/2: selection into a malloc'd array
// keep_on_device: *recs is the ordered DEVICE array (workspace WS_CW_SEL, valid until the next call), nothing is copied back
static int64_t cov_est_key(const cornetto_cov_t *c, int mode, int32_t lo, int32_t hi, float low_mq, int32_t edge, int32_t min_len)
{
    uint32_t qb;
    memcpy(&qb, &low_mq, 4);
    uint64_t k = 1469598103934665603ull;
    for (uint64_t v : {(uint64_t)(uint32_t)mode, (uint64_t)(uint32_t)lo, (uint64_t)(uint32_t)hi, (uint64_t)qb, (uint64_t)(uint32_t)edge, (uint64_t)(uint32_t)min_len,
                       (uint64_t)(uint32_t)c->w, (uint64_t)(uint32_t)c->inc})
        k = (k ^ v) * 1099511628211ull;
    return (int64_t)(k >> 1);
}

// spec (packed form only): queue everything, the result copy sized by the last count for these parameters, and return WITHOUT synchronising
static int cov_run_windows(cornetto_accel_t *h, cornetto_cov_t *c, int32_t only_ctg, int mode, int32_t lo, int32_t hi,
                           float low_mq, int32_t edge, int32_t min_len, cornetto_reg_t *regs_host, cornetto_regrec_t **recs,
                           int64_t *n_recs, bool keep_on_device = false, cornetto_regpk_t **pk_out = nullptr, int64_t **ctg_first = nullptr,
                           CnCovSpec *spec = nullptr)
{
    const bool packed = pk_out != nullptr || spec != nullptr;
    CN_TRACE("cov_select: enter");
    const int64_t est_key = cov_est_key(c, mode, lo, hi, low_mq, edge, min_len);
    if (spec) {
        spec->queued = false;
        if (c->cw_est_key != est_key || c->cw_est_cnt < 0) return CORNETTO_OK;   // nothing to size the copy by: the caller takes the exact call
    }
    const size_t rec_bytes = packed ? sizeof(cornetto_regpk_t) : sizeof(cornetto_regrec_t);
    if (!c->d_blk) return cn_fail(h, CORNETTO_E_ARG, "cov: cornetto_cov_prepare() has not been called");
    const int32_t w = c->w, inc = c->inc, q = w / inc, r = w % inc;
    // A window mean fits 16 bits as long as the reference's `int` sums cannot wrap: w x 65535 < 2^31.  Then the raw array (256 entries per
    // window tile: 250 000 tiles for 3 Gbp) holds the 8-byte form for either record type — 0.5 GB instead of 1.3 GB of workspace for the
    // full records — and the ordering pass expands them.  Larger windows keep full records; the packed interface refuses them.
    const bool raw_packed = w <= 32768;
    if (packed && !raw_packed) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "cov_select_packed: window size %d > 32768 (a wrapped sum's mean does not fit 16 bits)", w);
    const size_t raw_bytes = raw_packed ? sizeof(cornetto_regpk_t) : sizeof(cornetto_regrec_t);
    if (c->cw_mode != mode || c->cw_min_len != min_len || c->cw_only != only_ctg) {   // window tiles, cached
        // on the device (tiles.hpp): window tile j of contig i = its windows 256 j ...; the contigs the selection leaves out have none
        const int64_t ntl = cntiles::prefix(h, c->cw_pref, c->n, [&](int32_t i) -> int64_t {
            if (only_ctg >= 0 && i != only_ctg) return 0;
            if (mode == 1 && !(c->len[i] >= min_len)) return 0;   // short contigs print one '.' line and no windows
            if (mode == 2 && !(c->len[i] > min_len)) return 0;
            return ((int64_t)c->n_reg[i] + 255) / 256;
        });
        if (ntl < 0) return cn_fail(h, CORNETTO_E_NOMEM, "cov: device allocation failed");
        if (c->d_cw_tiles) { (void)hipFree(c->d_cw_tiles); c->d_cw_tiles = nullptr; }
        if (c->d_cw_first) { (void)hipFree(c->d_cw_first); c->d_cw_first = nullptr; }
        c->n_cw_tiles = ntl;
        if (ntl > 0) {
            // the first tile of every contig (a contig without tiles: the first tile of the next one that has — or their number)
            c->cw_first.resize((size_t)c->n + 1);
            for (int32_t i = 0; i <= c->n; ++i) c->cw_first[i] = (int32_t)c->cw_pref.host[i];
            if (cn_obj_malloc(h, (void **)&c->d_cw_tiles, (size_t)ntl * sizeof(int2)) != hipSuccess || cn_obj_malloc(h, (void **)&c->d_cw_first, ((size_t)c->n + 1) * 4) != hipSuccess)
                return cn_fail(h, CORNETTO_E_NOMEM, "cov: device allocation failed");
            cntiles::fill<<<dim3((unsigned)((ntl + 255) / 256)), dim3(256), 0, h->stream>>>(c->cw_pref.dev, c->n, ntl, CwTileFill{c->d_cw_tiles});
            CN_HIP(h, hipGetLastError());
            CN_HIP(h, hipMemcpyAsync(c->d_cw_first, c->cw_first.data(), (size_t)c->n * 4, hipMemcpyHostToDevice, h->stream));   // (a member of the object)
        }
        c->cw_mode = mode;
        c->cw_min_len = min_len;
        c->cw_only = only_ctg;
    }
    const size_t nt = (size_t)c->n_cw_tiles;
    if (nt == 0) return CORNETTO_OK;                 // (spec: not queued — the exact call makes the empty result)
    const size_t nt_blk = (size_t)c->n_cb_tiles;
    const uint2 *d_pre = reinterpret_cast<const uint2 *>(c->d_blk), *d_head = d_pre + c->n_blk;
    const uint32_t *d_toff_d = reinterpret_cast<const uint32_t *>(d_head + c->n_blk), *d_toff_q = d_toff_d + nt_blk;
    unsigned long long *d_cnt = (unsigned long long *)cn_ws(h, WS_CW_CNT, 16);
    // per tile: {base,count} (8 B) + ordered offset (4 B) + scan partials
    uint2 *d_tres = (uint2 *)cn_ws(h, WS_CW_TRES, nt * 12 + ((nt + 4095) / 4096 + 1) * 4);
    unsigned long long *p_cnt = spec ? spec->p_cnt : (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
    if (!d_cnt || !d_tres || !p_cnt) return cn_fail(h, CORNETTO_E_NOMEM, "cov: workspace allocation failed");
    uint32_t *d_ooff = reinterpret_cast<uint32_t *>(d_tres + nt), *d_part = d_ooff + nt;
    CwArgs A{};
    A.pre = d_pre; A.head = d_head; A.toff_d = d_toff_d; A.toff_q = d_toff_q; A.blk_off = c->d_blk_off; A.ctg_len = c->d_len; A.n_reg = c->d_n_reg;
    A.tiles = c->d_cw_tiles; A.w = w; A.inc = inc; A.q = q; A.r = r; A.mode = mode; A.lo = lo; A.hi = hi; A.edge = edge;
    if (w >= 2) {                                   // floor(n / w) for 0 <= n < 2^31: M = floor(2^(31 + L) / w) + 1, L = ceil(log2 w)
        int L = 0;
        while ((1ll << L) < (long long)w) ++L;
        A.w_magic = (uint32_t)(((1ull << (31 + L)) / (unsigned long long)w) + 1ull);
        A.w_shift = (uint32_t)(L - 1);
    }
    A.min_len = min_len; A.low_mq = (double)low_mq;   // float promoted exactly as in `x < low_mq_cov_thresh`
    A.tile_res = d_tres;
    if (mode == 0) {
        const size_t nr = (size_t)c->n_reg[only_ctg];
        cornetto_reg_t *d_regs = (cornetto_reg_t *)cn_ws(h, WS_CW_REGS, nr * sizeof(cornetto_reg_t));
        if (!d_regs) return cn_fail(h, CORNETTO_E_NOMEM, "cov: workspace allocation failed");
        A.regs = d_regs;
        CN_LAUNCH(h, "cov_windows", cov_windows<<<dim3((unsigned)nt), dim3(256), 0, h->stream>>>(A));
        CN_HIP(h, hipMemcpyAsync(regs_host, d_regs, nr * sizeof(cornetto_reg_t), hipMemcpyDeviceToHost, h->stream));
        CN_HIP(h, hipStreamSynchronize(h->stream));
        return CORNETTO_OK;
    }
    // raw (256 entries per tile, a tile's selected windows at the front of its own) and ordered copies share one workspace: [nt * 256] + [cap]
    const size_t n_raw = nt * 256;
    if (n_raw > 0xffffff00ull) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "cov: %zu window tiles", nt);
    size_t cap = std::max<size_t>(1 << 16, nt * 256 / 8);
    if (h->dev[WS_CW_SEL].bytes > n_raw * raw_bytes) cap = std::max(cap, (h->dev[WS_CW_SEL].bytes - n_raw * raw_bytes) / rec_bytes);   // keep what an earlier call grew to
    unsigned long long cnt = 0;
    cornetto_regrec_t *d_raw = nullptr, *d_dst = nullptr;
    uint32_t *p_cf = nullptr, *d_cf = nullptr;
    if (packed) {
        p_cf = (uint32_t *)cn_pin(h, PIN_CW, ((size_t)c->n + 1) * 4);
        d_cf = (uint32_t *)cn_ws(h, WS_CW_CF, ((size_t)c->n + 1) * 4);
        if (!p_cf || !d_cf) return cn_fail(h, CORNETTO_E_NOMEM, "cov_select: allocation failed");
    }
    CN_TRACE("cov_select: tables + small workspaces");
    // selection, ordering and (packed) the first record of every contig are queued in one go: the ordering kernels read the count on the
    // device, so the host meets it once — together with the per-contig offsets — and only then sizes the result
    for (int attempt = 0; attempt < 2; ++attempt) {
        cap = std::min<size_t>(cap, 0x7fffffff);
        d_raw = (cornetto_regrec_t *)cn_ws(h, WS_CW_SEL, n_raw * raw_bytes + cap * rec_bytes);
        if (!d_raw) return cn_fail(h, CORNETTO_E_NOMEM, "cov: workspace allocation of %zu bytes failed", n_raw * raw_bytes + cap * rec_bytes);
        d_dst = reinterpret_cast<cornetto_regrec_t *>(reinterpret_cast<uint8_t *>(d_raw) + n_raw * raw_bytes);
        A.sel = d_raw;
        A.pk = raw_packed ? reinterpret_cast<cornetto_regpk_t *>(d_raw) : nullptr;
        CN_LAUNCH(h, "cov_windows", cov_windows<<<dim3((unsigned)nt), dim3(256), 0, h->stream>>>(A));
        // tiles are in (contig, window) order: exclusive scan of their counts = final position of each segment; its total = the number selected
        CN_TRY(cnscan::exclusive_u32(h, "cov_order", reinterpret_cast<const uint32_t *>(d_tres) + 1, (int64_t)nt, 2, d_ooff, d_part, d_cnt));
        const unsigned nb = (unsigned)((nt + 3) / 4);
        hipEvent_t ea = cn_event(h), eb = cn_event(h);
        (void)hipEventRecord(ea, h->stream);
        if (packed) cov_order<2><<<dim3(nb), dim3(256), 0, h->stream>>>(reinterpret_cast<const int32_t *>(d_raw), d_tres, d_ooff, (int64_t)nt, reinterpret_cast<int32_t *>(d_dst), d_cnt, (uint32_t)cap);
        else if (raw_packed) cov_order_expand<<<dim3(nb), dim3(256), 0, h->stream>>>(reinterpret_cast<const cornetto_regpk_t *>(d_raw), d_tres, d_ooff, (int64_t)nt, d_dst, d_cnt, (uint32_t)cap, c->d_cw_tiles, c->d_len, w);
        else cov_order<5><<<dim3(nb), dim3(256), 0, h->stream>>>(reinterpret_cast<const int32_t *>(d_raw), d_tres, d_ooff, (int64_t)nt, reinterpret_cast<int32_t *>(d_dst), d_cnt, (uint32_t)cap);
        (void)hipEventRecord(eb, h->stream);
        h->recs.push_back(cornetto_accel::Rec{"cov_order", ea, eb});
        if (hipGetLastError() != hipSuccess) return cn_fail(h, CORNETTO_E_HIP, "cov_select: ordering failed");
        if (packed) {
            // the first record of every contig = the ordered offset of its first tile, picked on the device (4 B per contig to the host)
            cov_ctg_first<<<dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, h->stream>>>(d_ooff, c->d_cw_first, c->n, (uint32_t)nt, d_cnt, d_cf);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(p_cf, d_cf, (size_t)c->n * 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess)
                return cn_fail(h, CORNETTO_E_HIP, "cov_select: first records of the contigs failed");
        }
        CN_HIP(h, hipMemcpyAsync(p_cnt, d_cnt, 8, hipMemcpyDeviceToHost, h->stream));
        if (spec) {
            // the copy goes out now, for the count of last time plus head room (at most the block); whoever synchronises the stream checks
            size_t n_copy = std::min<size_t>(cap, (size_t)c->cw_est_cnt + (size_t)c->cw_est_cnt / 16 + 1024);
            if (const int f = CN_DEV_INT("CORNETTO_STEP_EST_FORCE", 0)) n_copy = std::min<size_t>(cap, (size_t)std::max(1, f));   // (tests: an estimate that does not hold)
            cornetto_regpk_t *o = (cornetto_regpk_t *)cn_result_alloc(n_copy * rec_bytes);
            if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "cov_select: host allocation failed");
            if (cn_result_d2h(h, o, d_dst, n_copy * rec_bytes) != hipSuccess) {
                cn_result_quiesce(h);
                cornetto_free(o);
                return cn_fail(h, CORNETTO_E_HIP, "cov_select: copy back failed");
            }
            spec->queued = true; spec->o = o; spec->n_copy = n_copy; spec->cap = cap; spec->p_cf = p_cf; spec->key = est_key;
            return CORNETTO_OK;
        }
        CN_HIP(h, hipStreamSynchronize(h->stream));
        cnt = p_cnt[0];
        if (cnt <= cap) break;
        if (attempt == 1 || cnt > 0x7fffffffull) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "cov: %llu selected windows", cnt);
        cap = (size_t)cnt;   // exact rerun, never a truncated answer
    }
    CN_TRACE("cov_select: kernels done, count known");
    cornetto_regrec_t *o = keep_on_device ? d_dst : (cornetto_regrec_t *)cn_result_alloc((cnt ? cnt : 1) * rec_bytes);
    CN_TRACE("cov_select: result block");
    if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "cov_select: host allocation failed");
    if (cnt) {
        if (!keep_on_device && (cn_result_d2h(h, o, d_dst, (size_t)cnt * rec_bytes) != hipSuccess || (!h->lazy && hipStreamSynchronize(h->stream) != hipSuccess))) {
            cn_result_quiesce(h);
            cornetto_free(o);
            return cn_fail(h, CORNETTO_E_HIP, "cov_select: copy back failed");
        }
        if (packed) {
            int64_t *cf = (int64_t *)malloc(((size_t)c->n + 1) * sizeof(int64_t));
            if (!cf) { cn_result_quiesce(h); cornetto_free(o); return cn_fail(h, CORNETTO_E_NOMEM, "cov_select: host allocation failed"); }
            for (int32_t i = 0; i < c->n; ++i) cf[i] = (int64_t)p_cf[i];   // (contigs without tiles have no records: the offset of the next one that has)
            cf[c->n] = (int64_t)cnt;
            *ctg_first = cf;
        }
    } else if (packed) {
        int64_t *cf = (int64_t *)calloc((size_t)c->n + 1, sizeof(int64_t));
        if (!cf) { if (!keep_on_device) cornetto_free(o); return cn_fail(h, CORNETTO_E_NOMEM, "cov_select: host allocation failed"); }
        *ctg_first = cf;
    }
    if (packed) {
        *pk_out = reinterpret_cast<cornetto_regpk_t *>(o);
        c->cw_est_key = est_key;
        c->cw_est_cnt = (int64_t)cnt;
    } else {
        *recs = o;
    }
    *n_recs = (int64_t)cnt;
    return CORNETTO_OK;
}

int cornetto_cov_regs(cornetto_accel_t *h, const cornetto_cov_t *c, int32_t ctg, cornetto_reg_t *regs)
{
    if (!h || !c || !regs || ctg < 0 || ctg >= c->n) return cn_fail(h, CORNETTO_E_ARG, "cov_regs: bad argument");
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    int rc = cov_run_windows(h, const_cast<cornetto_cov_t *>(c), ctg, 0, 0, 0, 0.f, 0, 0, regs, nullptr, nullptr);
    cn_timing_end(h);
    return rc;
}

int cornetto_cov_select(cornetto_accel_t *h, const cornetto_cov_t *c, int32_t lo, int32_t hi, float low_mq, int32_t edge_len,
                        int32_t min_ctg_len, int boring, cornetto_regrec_t **recs, int64_t *n_recs)
{
    if (!h || !c || !recs || !n_recs) return cn_fail(h, CORNETTO_E_ARG, "cov_select: bad argument");
    *recs = nullptr;
    *n_recs = 0;
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    int rc = cov_run_windows(h, const_cast<cornetto_cov_t *>(c), -1, boring ? 2 : 1, lo, hi, low_mq, edge_len, min_ctg_len, nullptr, recs, n_recs);
    cn_timing_end(h);
    if (rc == CORNETTO_OK && !*recs) {   // nothing to scan (no contig qualifies): an empty, freeable result
        *recs = (cornetto_regrec_t *)malloc(sizeof(cornetto_regrec_t));
        if (!*recs) return cn_fail(h, CORNETTO_E_NOMEM, "cov_select: host allocation failed");
    }
    return rc;
}

int cornetto_cov_select_packed(cornetto_accel_t *h, const cornetto_cov_t *c, int32_t lo, int32_t hi, float low_mq, int32_t edge_len,
                               int32_t min_ctg_len, int boring, cornetto_regpk_t **recs, int64_t *n_recs, int64_t **ctg_first)
{
    if (!h || !c || !recs || !n_recs || !ctg_first) return cn_fail(h, CORNETTO_E_ARG, "cov_select_packed: bad argument");
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    const int rc = cn_cov_select_packed_impl(h, const_cast<cornetto_cov_t *>(c), lo, hi, low_mq, edge_len, min_ctg_len, boring, recs, n_recs, ctg_first);
    cn_timing_end(h);
    return rc;
}

}  // extern "C"

int cn_cov_spec_queue(cornetto_accel_t *h, cornetto_cov_t *c, int32_t lo, int32_t hi, float low_mq, int32_t edge_len, int32_t min_ctg_len, int boring,
                      unsigned long long *p_cnt, CnCovSpec *S)
{
    S->p_cnt = p_cnt;
    return cov_run_windows(h, c, -1, boring ? 2 : 1, lo, hi, low_mq, edge_len, min_ctg_len, nullptr, nullptr, nullptr, false, nullptr, nullptr, S);
}

int cn_cov_spec_finish(cornetto_accel_t *h, cornetto_cov_t *c, CnCovSpec *S, cornetto_regpk_t **recs, int64_t *n_recs, int64_t **ctg_first)
{
    if (!S->queued) return 1;
    S->queued = false;
    const unsigned long long cnt = S->p_cnt[0];
    int64_t *cf = cnt <= S->n_copy ? (int64_t *)malloc(((size_t)c->n + 1) * sizeof(int64_t)) : nullptr;
    if (!cf) {                                        // the count outgrew the copy (or no memory): nothing of this attempt is returned
        cn_result_quiesce(h);
        cornetto_free(S->o);
        S->o = nullptr;
        c->cw_est_key = -1;
        return cnt <= S->n_copy ? cn_fail(h, CORNETTO_E_NOMEM, "cov_select: host allocation failed") : 1;
    }
    for (int32_t i = 0; i < c->n; ++i) cf[i] = (int64_t)S->p_cf[i];
    cf[c->n] = (int64_t)cnt;
    *recs = S->o;
    *n_recs = (int64_t)cnt;
    *ctg_first = cf;
    S->o = nullptr;
    c->cw_est_key = S->key;
    c->cw_est_cnt = (int64_t)cnt;
    return CORNETTO_OK;
}

int cn_cov_select_packed_impl(cornetto_accel_t *h, cornetto_cov_t *c, int32_t lo, int32_t hi, float low_mq, int32_t edge_len, int32_t min_ctg_len, int boring,
                              cornetto_regpk_t **recs, int64_t *n_recs, int64_t **ctg_first)
{
    *recs = nullptr;
    *n_recs = 0;
    *ctg_first = nullptr;
    int rc = cov_run_windows(h, c, -1, boring ? 2 : 1, lo, hi, low_mq, edge_len, min_ctg_len, nullptr, nullptr, n_recs, false, recs, ctg_first);
    if (rc == CORNETTO_OK && !*recs) {   // nothing to scan (no contig qualifies): empty, freeable results
        *recs = (cornetto_regpk_t *)malloc(sizeof(cornetto_regpk_t));
        if (!*ctg_first) *ctg_first = (int64_t *)calloc((size_t)c->n + 1, sizeof(int64_t));
        if (!*recs || !*ctg_first) return cn_fail(h, CORNETTO_E_NOMEM, "cov_select_packed: host allocation failed");
    }
    return rc;
}

extern "C" {

namespace {
__global__ void cov_rec_spans(const cornetto_regrec_t *r, int64_t n, cornetto_ivl_t *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cornetto_ivl_t{r[i].ctg, r[i].st, r[i].end};
}
}  // namespace

int cornetto_cov_select_merged(cornetto_accel_t *h, const cornetto_cov_t *c, int32_t lo, int32_t hi, float low_mq, int32_t edge_len,
                               int32_t min_ctg_len, int boring, int32_t merge_dist, int32_t min_len, cornetto_ivl_t **ivls, int64_t *n_ivls)
{
    if (!h || !c || !ivls || !n_ivls || merge_dist < 0) return cn_fail(h, CORNETTO_E_ARG, "cov_select_merged: bad argument");
    *ivls = nullptr;
    *n_ivls = 0;
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    cornetto_regrec_t *d_recs = nullptr;
    int64_t n = 0;
    CN_TRY(cov_run_windows(h, const_cast<cornetto_cov_t *>(c), -1, boring ? 2 : 1, lo, hi, low_mq, edge_len, min_ctg_len, nullptr, &d_recs, &n, true));
    cornetto_ivl_t *o = nullptr;
    int64_t n_out = 0;
    if (n > 0) {
        // the selected windows are in (contig, start) order already: spans -> merge on the device -> only the merged list comes back
        uint8_t *ws = (uint8_t *)cn_ws(h, WS_CW_MERGE, cnivl::ws_bytes((size_t)n) + 2 * (size_t)n * sizeof(cornetto_ivl_t));
        unsigned long long *d_cnt = (unsigned long long *)cn_ws(h, WS_CW_CNT, 16);
        unsigned long long *p_cnt = (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
        if (!ws || !d_cnt || !p_cnt) return cn_fail(h, CORNETTO_E_NOMEM, "cov_select_merged: workspace allocation failed");
        cornetto_ivl_t *d_in = (cornetto_ivl_t *)(ws + cnivl::ws_bytes((size_t)n)), *d_out = d_in + n;
        CN_LAUNCH(h, "cov_merge", cov_rec_spans<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream>>>(d_recs, n, d_in));
        CN_TRY(cnivl::merge(h, "cov_merge", d_in, n, merge_dist, ws, d_out, d_cnt + 1));
        CN_HIP(h, hipMemcpyAsync(p_cnt, d_cnt + 1, 8, hipMemcpyDeviceToHost, h->stream));
        CN_HIP(h, hipStreamSynchronize(h->stream));
        const int64_t m = (int64_t)p_cnt[0];
        o = (cornetto_ivl_t *)cn_result_alloc(((size_t)m ? (size_t)m : 1) * sizeof(cornetto_ivl_t));
        if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "cov_select_merged: host allocation failed");
        if (m > 0 && (hipMemcpyAsync(o, d_out, (size_t)m * sizeof(cornetto_ivl_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                      hipStreamSynchronize(h->stream) != hipSuccess)) {
            cornetto_free(o);
            return cn_fail(h, CORNETTO_E_HIP, "cov_select_merged: copy back failed");
        }
        for (int64_t i = 0; i < m; ++i)                      // awk '($3-$2)>=min_len' of the scripts: the merged list is short
            if (o[i].finish - o[i].start >= min_len) o[n_out++] = o[i];
    }
    cn_timing_end(h);
    if (!o) {
        o = (cornetto_ivl_t *)malloc(sizeof(cornetto_ivl_t));
        if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "cov_select_merged: host allocation failed");
    }
    *ivls = o;
    *n_ivls = n_out;
    return CORNETTO_OK;
}

}  // extern "C"

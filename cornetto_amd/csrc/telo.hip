// telo.hip — telofind (motif run finding, both strands) and telowin (1000/200 mark-density windows) for
// gfx950.  Replaces find() src/find_telomere.c:44-74 and process_scaffold() src/telomere_windows.c:28-43
// of the reference; see include/cornetto_accel.h for the boundary.
//
// tf_scan   one pass over the bases (1 B/base from HBM).  A 256-thread workgroup owns a tile of 254 x 64
//           positions of one contig; every thread runs a shift-and automaton for the motif and for its
//           reverse complement over its own 64 positions (+ halo) out of registers, yielding two 64-bit
//           match masks.  Neighbouring masks are exchanged through LDS; for a motif without a proper
//           border (TTAGGG, CCCTAA, every real telomere unit) matches cannot overlap, so the reference's
//           greedy runs are exactly  head = m[p] & !m[p-k],  tail = m[p] & !m[p+k]  
//           and the i-th head pairs with the i-th tail.  Heads/tails are compacted with a packed
//           4-counter block scan into a small fixed row per tile (no atomics: a returning atomic per tile
//           on one counter line costs ~12 ns and was 95 % of the kernel); device scans of the per-tile
//           counts then give every tile its place in the dense, contig-ordered lists (tf_gather), and
//           tf_pair writes the final records in the reference's print order.  A tile that overflows its row
//           (never for real telomere motifs) triggers a second pass that writes straight to the scanned
//           offsets.  The same pass writes the telowin mark bitmap (1 bit/base) = union of [p,p+k) over matches.
// tf_greedy for motifs WITH a border (AAAA, ACACA ...) only: the sequential greedy rule of
//           src/find_telomere.c:49-58 over the compacted match list, one thread per (contig,strand).
// tw_fill   marks [start,end) of explicit hits in the bitmap (src/telomere_windows.c:75-79).
// tw_scan   one thread per 200-bp window start: popcount of <=1000 bitmap bits, double-precision
//           car/den >= threshold exactly as :36-37, passing windows appended.
#include <algorithm>
#include <cmath>
#include <string>

#include "common.hpp"
#include "scan.hpp"
#include "tiles.hpp"
#include "internal.hpp"

namespace {

constexpr int TF_THREADS = 256;
constexpr int TF_SEG = 64;
constexpr int TF_TILE = (TF_THREADS - 2) * TF_SEG;  // 16256 positions; threads 0 and 255 are halo only
constexpr int MAX_MOTIF = 32;
constexpr int TF_ROW = 32;   // list entries a tile can hold in its fixed row (single-pass mode)

struct TfArgs {
    const uint8_t *bases;
    const int64_t *ctg_off;
    const int32_t *ctg_len;
    const int2 *tiles;   // {ctg, first position}
    const uint2 *lut;    // [256] {fwd mask, rev mask}
    const uint8_t *mot;  // tf_scan<-1> (motifs longer than 32 bytes): the motif, then its reverse complement, k bytes each
    int32_t k;
    int32_t bordered;    // 0: heads/tails + bitmap; 1: all matches (lists 0 and 2)
    int32_t mode;        // 0: count only; 1: write at tile_off (dense lists); 2: write into fixed rows of TF_ROW
    unsigned long long *bitmap;
    const int64_t *bm_off;   // first bit of every contig in the bitmap (multiples of 64: the contigs back to back, whatever lies between them in memory)
    int32_t *list0, *list1, *list2, *list3;   // mode 1: dense lists; mode 2: [tile][TF_ROW] rows
    const uint32_t *off0, *off1, *off2, *off3;   // mode 1: exclusive scan of the per-tile counts
    uint4 *tile_cnt;
    uint32_t *ovf;       // mode 2: max per-tile count when it exceeds TF_ROW
    int64_t n_tiles;
};

__device__ __forceinline__ unsigned long long shfl_up64(unsigned long long v, int d)
{
    unsigned lo = __shfl_up((unsigned)v, d), hi = __shfl_up((unsigned)(v >> 32), d);
    return ((unsigned long long)hi << 32) | lo;
}

// union of [p, p+k) over set bits p of (prev:cur), restricted to cur's 64 positions (k <= 64): the covered width doubles
// per step (1, 2, 4, ... then the rest), on the 128-bit pair
__device__ __forceinline__ unsigned long long smear(unsigned long long cur, unsigned long long prev, int k)
{
    unsigned long long lo = prev, hi = cur;            // bit i of lo = position i - 64
    int w = 1;                                          // every set bit p covers [p, p + w) so far
    while (w < k) {
        const int s = w < k - w ? w : k - w;            // shift by s: covers [p, p + w + s)
        hi |= (hi << s) | (lo >> (64 - s));
        lo |= lo << s;
        w += s;
    }
    return hi;
}

// inclusive scan over the wave in six DPP additions (row shifts, then the two row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
    return (uint32_t)x;
}

// TF_NT consecutive tiles per workgroup (round 5).  One tile per workgroup loaded its 80 bytes per thread, waited for them, then computed: beside
// the resident sdust waves only two such waves fit on a SIMD (their registers), and with all of a wave's loads ahead of all of its work the
// scan was bound by the round trips — 1.6 ms in the bench step against 0.71 alone.  Now the 16 bytes a thread has just used up are re-loaded
// with the NEXT tile's 16 bytes at the same place (the same registers): the loads of tile i + 1 travel while tile i is computed — 1.28 ms in
// the step (0.84 alone: the loop costs there).  Measured twice: with sdust on a second host thread and chunks of 1536 bases what this stream
// gained the other lost (step 6.86-6.95 against 6.78-6.80 ms); with the step on one host thread and chunks of 1792 bases the telomere scan was
// what the step waited for at the share the coverage kernel wants, and the step is 6.32-6.34 (share 76) against 6.49-6.57 (share 72) ms.
constexpr int TF_NT = 4;

template <int H>  // halo bytes behind the 64 positions of a thread; motif length k <= H + 1
__global__ __launch_bounds__(TF_THREADS) void tf_scan(TfArgs A)
{
    __builtin_amdgcn_s_setprio(CN_STREAM_PRIO);   // short streaming kernel: issue ahead of a long compute-bound kernel of another stream

    __shared__ uint2 lut[256];
    __shared__ unsigned long long shF[TF_THREADS], shR[TF_THREADS];
    __shared__ unsigned long long wtot[TF_THREADS / 64];

    const int t = threadIdx.x;
    lut[t] = A.lut[t];
    constexpr int HH = H < 0 ? 0 : H;
    constexpr int NW = (TF_SEG + HH + 3) / 4;      // dwords touched
    constexpr int NV = H < 0 ? 1 : (NW + 3) / 4;   // 16-byte loads
    uint32_t w[NV * 4];
    const int64_t ti0 = (int64_t)blockIdx.x * TF_NT;
    // what the tiles of this workgroup are, through LDS: read inside the loop from memory these values would come by vector loads (the kernel
    // stores, so nothing is invariant to the compiler), in order with the tiles' bytes on the one counter both are waited for with
    __shared__ int m_ctg[TF_NT + 1], m_y[TF_NT + 1], m_len[TF_NT + 1];
    __shared__ int64_t m_off[TF_NT + 1], m_bm[TF_NT + 1];
    if (t <= TF_NT) {
        const bool there = t < TF_NT && ti0 + t < A.n_tiles;
        const int2 tl = there ? A.tiles[ti0 + t] : make_int2(-1, 0);
        m_ctg[t] = tl.x;
        m_y[t] = tl.y;
        m_len[t] = there ? A.ctg_len[tl.x] : 0;          // (no tile: no thread has positions, everybody loads the array's first bytes)
        m_off[t] = there ? A.ctg_off[tl.x] : 0;
        m_bm[t] = there && A.bm_off ? A.bm_off[tl.x] : 0;
    }
    // a thread without positions in its tile (in front of the contig, behind its end) loads from the start of the array: any bytes will do
    auto src_at = [&](int y, int len, int64_t off) {
        const int s = y + (t - 1) * TF_SEG;
        return reinterpret_cast<const uint4 *>(s >= 0 && s < len ? A.bases + off + s : A.bases);
    };
    if constexpr (H >= 0) {
        const int2 tl = A.tiles[ti0];
        const uint4 *src = src_at(tl.y, A.ctg_len[tl.x], A.ctg_off[tl.x]);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const uint4 v = src[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    }
    __syncthreads();

    for (int it = 0; it < TF_NT; ++it) {
    const int64_t ti = ti0 + it;
    const int ctg = __builtin_amdgcn_readfirstlane(m_ctg[it]);
    if (ctg < 0) break;
    const int len = __builtin_amdgcn_readfirstlane(m_len[it]);
    const int64_t off = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(m_off[it] >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)m_off[it]));
    const int s0 = __builtin_amdgcn_readfirstlane(m_y[it]) + (t - 1) * TF_SEG;
    const uint4 *nsrc = nullptr;                    // the next tile's bytes (behind the last tile: the array's first bytes, loaded and dropped)
    if constexpr (H >= 0) nsrc = src_at(m_y[it + 1], m_len[it + 1], m_off[it + 1]);

    unsigned long long Mf = 0, Mr = 0;
    if constexpr (H < 0) {
        // motifs longer than 32 bytes (not a telomere unit; the reference takes any string): every start position compared
        // byte by byte with the motif and with its reverse complement, the sequence upper-cased as src/find_telomere.c:76-81
        // does ('a'..'z' only), stopping at the first difference of both
        if (s0 >= 0 && s0 < len) {
            const uint8_t *seq = A.bases + off;
            const int k = A.k;
            for (int b = 0; b < TF_SEG && s0 + b + k <= len; ++b) {
                bool mf = true, mr = true;
                for (int j = 0; j < k && (mf || mr); ++j) {
                    uint8_t u = seq[s0 + b + j];
                    u = (u >= 'a' && u <= 'z') ? (uint8_t)(u - 32) : u;
                    mf = mf && u == A.mot[j];
                    mr = mr && u == A.mot[k + j];
                }
                Mf |= (unsigned long long)mf << b;
                Mr |= (unsigned long long)mr << b;
            }
        }
    } else {
        // (every thread computes — one without positions on whatever it loaded; its masks are cleared below)
        constexpr uint32_t INJ = 1u << (31 - HH);
        uint32_t Sf = 0, Sr = 0, Af = 0, Bf = 0, Ar = 0, Br = 0;
        // the 16 bytes behind byte e are through: the next tile's come into their place
        auto refill = [&](int e, int last, uint32_t &S0, uint32_t &S1) __attribute__((always_inline)) {
            if ((e & 15) == 15 || e == last) {
                const int i = e >> 4;
                // (the load is tied behind the last use of the registers it fills: left to itself the compiler issues all of them at the top of
                // the tile into twenty registers of their own, and a SIMD that holds the sdust waves has room for one such wave instead of two;
                // the automaton's state as well: what the table gave for these 16 bytes is used up here, not kept for a chain of steps at the tile's end)
                unsigned long long pa = reinterpret_cast<unsigned long long>(nsrc);
                asm volatile("" : "+v"(pa), "+v"(w[4 * i]), "+v"(w[4 * i + 1]), "+v"(w[4 * i + 2]), "+v"(w[4 * i + 3]), "+v"(S0), "+v"(S1));
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(pa)[i];      // (a global load, not a flat one)
                w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
            }
        };
        if constexpr (H == 7) {
            // Motifs of up to 8 bytes (every telomere unit): BOTH automata in one 32-bit state, each followed by an 8-bit delay
            // line — reverse strand: automaton bits 0-7 (match = bit 7), delay 8-15; forward: automaton 16-23, delay 24-31.  The
            // delay bits are ones in every table entry, so a match bit just travels upwards one bit per step, and eight steps of
            // matches are harvested at once: 3 vector instructions per byte (address, shift-or, and) instead of 8.  What leaves
            // the reverse delay line lands on bit 16, which the injection sets anyway; the forward one falls off the top.
            constexpr uint32_t INJ2 = (1u << 16) | 1u;
            uint32_t S = 0;
#pragma unroll
            for (int e = 0; e < TF_SEG + 8; ++e) {
                const uint32_t c = (w[e >> 2] >> (8 * (e & 3))) & 0xFFu;
                S = ((S << 1) | INJ2) & lut[c].x;
                if (e >= 15 && (e - 15) % 8 == 0) {     // delay bit 24 + d: a match that starts at e - 8 - d (d = 0..7)
                    const uint32_t fb = S >> 24, rb = (S >> 8) & 0xFFu;
                    if (e < 15 + 32) {
                        Af = (Af << 8) | fb;
                        Ar = (Ar << 8) | rb;
                    } else {
                        Bf = (Bf << 8) | fb;
                        Br = (Br << 8) | rb;
                    }
                }
                refill(e, TF_SEG + 8 - 1, S, Af);
            }
        } else {
#pragma unroll
        for (int e = 0; e < TF_SEG + HH; ++e) {
            const uint32_t c = (w[e >> 2] >> (8 * (e & 3))) & 0xFFu;
            const uint2 L = lut[c];
            Sf = ((Sf << 1) | INJ) & L.x;
            Sr = ((Sr << 1) | INJ) & L.y;
            if (e >= HH) {                             // bit 31 of S = "a match starts at e - H"
                if (e - HH < 32) {
                    Af = __builtin_amdgcn_alignbit(Af, Sf, 31);
                    Ar = __builtin_amdgcn_alignbit(Ar, Sr, 31);
                } else {
                    Bf = __builtin_amdgcn_alignbit(Bf, Sf, 31);
                    Br = __builtin_amdgcn_alignbit(Br, Sr, 31);
                }
            }
            if ((e & 7) == 7) asm volatile("" : "+v"(Sf), "+v"(Sr), "+v"(Af), "+v"(Ar), "+v"(Bf), "+v"(Br) : : "memory");      // (table reads are looked ahead eight bases, not a tile: registers)
            refill(e, TF_SEG + HH - 1, Sf, Sr);
        }
        }
        if (s0 >= 0 && s0 < len) {
        Mf = (unsigned long long)__brev(Af) | ((unsigned long long)__brev(Bf) << 32);
        Mr = (unsigned long long)__brev(Ar) | ((unsigned long long)__brev(Br) << 32);
        const int nvalid = len - A.k + 1 - s0;        // start positions p with p + k <= len
        if (nvalid < 64) {
            const unsigned long long m = nvalid <= 0 ? 0ull : ((1ull << nvalid) - 1ull);
            Mf &= m;
            Mr &= m;
        }
        }
    }
    shF[t] = Mf;
    shR[t] = Mr;
    __syncthreads();

    const bool inner = (t >= 1 && t <= TF_THREADS - 2);
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (inner) {
        const unsigned long long pF = shF[t - 1], nF = shF[t + 1], pR = shR[t - 1], nR = shR[t + 1];
        const int k = A.k;
        if (A.bordered) {
            q0 = Mf;
            q2 = Mr;
        } else {
            q0 = Mf & ~((Mf << k) | (pF >> (64 - k)));
            q1 = Mf & ~((Mf >> k) | (nF << (64 - k)));
            q2 = Mr & ~((Mr << k) | (pR >> (64 - k)));
            q3 = Mr & ~((Mr >> k) | (nR << (64 - k)));
            if (A.bitmap && s0 >= 0 && s0 < len)
                A.bitmap[(m_bm[it] + s0) >> 6] = smear(Mf | Mr, pF | pR, k);      // (the union of [p, p + k) over the matches of either strand: one smear, not two)
        }
    }
    // packed 4 x 16-bit exclusive scan over the workgroup (a tile holds < 2^15 heads per list)
    unsigned long long pk = (unsigned long long)__popcll(q0) | ((unsigned long long)__popcll(q1) << 16) |
                            ((unsigned long long)__popcll(q2) << 32) | ((unsigned long long)__popcll(q3) << 48);
    // (two 16-bit counters per half, each below 2^15 for a whole tile: the halves scan independently, no carry between them)
    const int lane = t & 63, wv = t >> 6;
    const unsigned long long inc = (unsigned long long)wave_incl_scan_dpp((uint32_t)pk) | ((unsigned long long)wave_incl_scan_dpp((uint32_t)(pk >> 32)) << 32);
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    unsigned long long wpre = 0, total = 0;
#pragma unroll
    for (int i = 0; i < TF_THREADS / 64; ++i) {
        if (i < wv) wpre += wtot[i];
        total += wtot[i];
    }
    const unsigned long long excl = wpre + inc - pk;
    if (t == 0) {
        uint32_t cnt[4], mx = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cnt[q] = (uint32_t)((total >> (16 * q)) & 0xFFFFu);
            mx = cnt[q] > mx ? cnt[q] : mx;
        }
        if (A.mode != 1) A.tile_cnt[ti] = make_uint4(cnt[0], cnt[1], cnt[2], cnt[3]);
        if (A.mode == 2 && mx > TF_ROW) atomicMax(A.ovf, mx);   // rare: second pass will write densely
    }
    if (!inner || A.mode == 0) continue;
    int32_t *lists[4] = {A.list0, A.list1, A.list2, A.list3};
    const uint32_t *offs[4] = {A.off0, A.off1, A.off2, A.off3};
    unsigned long long qs[4] = {q0, q1, q2, q3};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned long long m = qs[q];
        if (!m) continue;
        uint32_t idx = (uint32_t)((excl >> (16 * q)) & 0xFFFFu);
        if (A.mode == 1) {
            int32_t *dst = lists[q] + offs[q][ti];
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                dst[idx++] = s0 + b;
            }
        } else {
            int32_t *dst = lists[q] + (size_t)ti * TF_ROW;
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                if (idx < TF_ROW) dst[idx] = s0 + b;
                ++idx;
            }
        }
    }
    }
}

// fixed rows -> dense lists in tile (= contig, position) order.  One WAVE per tile (round 5; before: a 256-thread workgroup per tile, one wave per
// list with half its lanes idle — 776 000 waves for the 194 000 tiles of a 3.16 Gbp assembly, 0.5 ms beside the resident sdust waves for 12 MB of
// traffic): a lane moves two of the tile's 4 x TF_ROW entries
__global__ __launch_bounds__(256) void tf_gather(const int32_t *r0, const int32_t *r1, const int32_t *r2, const int32_t *r3,
                                                 const uint4 *tile_cnt, const uint32_t *o0, const uint32_t *o1, const uint32_t *o2,
                                                 const uint32_t *o3, int32_t *d0, int32_t *d1, int32_t *d2, int32_t *d3, uint4 caps, int64_t n_tiles)
{
    static_assert(TF_ROW == 32, "two entries per lane: 4 lists x 32 entries per tile");
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const uint4 c4 = tile_cnt[tile];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = (lane >> 5) + 2 * k, idx = lane & 31;            // lanes 0-31: lists 0 / 2, lanes 32-63: lists 1 / 3
        const uint32_t cnt = q == 0 ? c4.x : q == 1 ? c4.y : q == 2 ? c4.z : c4.w;
        const uint32_t cap = q == 0 ? caps.x : q == 1 ? caps.y : q == 2 ? caps.z : caps.w;    // entries list q has room for (lists sized by an estimate: telo spec)
        const int32_t *src = (q == 0 ? r0 : q == 1 ? r1 : q == 2 ? r2 : r3) + tile * TF_ROW;
        const uint32_t at = (q == 0 ? o0 : q == 1 ? o1 : q == 2 ? o2 : o3)[tile];
        int32_t *dst = (q == 0 ? d0 : q == 1 ? d1 : q == 2 ? d2 : d3) + at;
        if ((uint32_t)idx < cnt && (unsigned long long)at + (uint32_t)idx < cap) dst[idx] = src[idx];
    }
}

// list offsets at contig boundaries: ctg_off[q][c] = number of entries of list q before contig c (c = n: total)
__global__ void tf_ctgoff(const int32_t *ctg_tile0, int32_t n_ctg, int64_t n_tiles, const uint32_t *o0, const uint32_t *o1,
                          const uint32_t *o2, const uint32_t *o3, const unsigned long long *totals, uint32_t *ctg_off /* [4][n+1] */,
                          uint32_t *err)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_ctg) return;
    const int64_t t = ctg_tile0[c];
    uint32_t v[4];
    if (t < n_tiles) { v[0] = o0[t]; v[1] = o1[t]; v[2] = o2[t]; v[3] = o3[t]; }
    else { v[0] = (uint32_t)totals[0]; v[1] = (uint32_t)totals[1]; v[2] = (uint32_t)totals[2]; v[3] = (uint32_t)totals[3]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) ctg_off[(size_t)q * (n_ctg + 1) + c] = v[q];
    if (v[0] != v[1] || v[2] != v[3]) atomicOr(err, 1u);   // every run has one head and one tail inside its contig
}

// the i-th head of a (contig, strand) pairs with its i-th tail; records go out in the reference's print
// order: by contig, strand 0 then strand 1, by position (src/find_telomere.c:49-72)
__global__ void tf_pair(const int32_t *hf, const int32_t *tf, const int32_t *hr, const int32_t *tr, const uint32_t *ctg_off,
                        int32_t n_ctg, uint32_t n_f, uint32_t n_r, int32_t k, cornetto_hit_t *out)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_f + n_r) return;
    const int strand = g >= n_f;
    const uint32_t i = strand ? g - n_f : g;
    const uint32_t *cf = ctg_off, *cr = ctg_off + 2 * (size_t)(n_ctg + 1);   // head offsets of strand 0 / strand 1
    const uint32_t *mine = strand ? cr : cf;
    int lo = 0, hi = n_ctg;                      // largest c with mine[c] <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (mine[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const int c = lo;
    const uint32_t pos = strand ? cf[c + 1] + cr[c] + (i - cr[c]) : cf[c] + cr[c] + (i - cf[c]);
    const int32_t st = strand ? hr[i] : hf[i], en = (strand ? tr[i] : tf[i]) + k;
    out[pos] = cornetto_hit_t{c, strand, st, en};
}

// the same with the list totals read on the device (lists and output sized by an estimate: nothing is touched when a total outgrew its list)
__global__ void tf_pair_dev(const int32_t *hf, const int32_t *tf, const int32_t *hr, const int32_t *tr, const uint32_t *ctg_off, int32_t n_ctg,
                            const unsigned long long *totals, uint4 caps, int32_t k, cornetto_hit_t *out)
{
    const unsigned long long t0 = totals[0], t1 = totals[1], t2 = totals[2], t3 = totals[3];
    if (t0 > caps.x || t1 > caps.y || t2 > caps.z || t3 > caps.w || t0 != t1 || t2 != t3) return;
    const uint32_t n_f = (uint32_t)t0, n_r = (uint32_t)t2;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_f + n_r) return;
    const int strand = g >= n_f;
    const uint32_t i = strand ? g - n_f : g;
    const uint32_t *cf = ctg_off, *cr = ctg_off + 2 * (size_t)(n_ctg + 1);
    const uint32_t *mine = strand ? cr : cf;
    int lo = 0, hi = n_ctg;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (mine[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const int c = lo;
    const uint32_t pos = strand ? cf[c + 1] + cr[c] + (i - cr[c]) : cf[c] + cr[c] + (i - cf[c]);
    if (pos >= n_f + n_r) return;                 // (unequal heads and tails inside a contig: the error flag of tf_ctgoff says so)
    const int32_t st = strand ? hr[i] : hf[i], en = (strand ? tr[i] : tf[i]) + k;
    out[pos] = cornetto_hit_t{c, strand, st, en};
}

// The reference's sequential rule for one (contig, strand): src/find_telomere.c:49-58.
// matches: ascending start positions of ALL occurrences, in dense lists; tile tl holds cnt entries from off[tl].
struct GreedyArgs {
    const int32_t *list0, *list2;
    const uint32_t *off0, *off2;   // per-tile offsets into the dense match lists
    const uint4 *tile_cnt;
    const int32_t *ctg_tile0;   // [n_ctg + 1] first tile of each contig
    const int64_t *run_off;     // [2 * n_ctg] first output slot of (ctg, strand)
    int32_t n_ctg, k;
    int2 *runs;                 // {start, end}
    int32_t *n_runs;            // [2 * n_ctg]
};

__global__ void tf_greedy(GreedyArgs G)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 2 * G.n_ctg) return;
    const int ctg = id >> 1, strand = id & 1;
    const int32_t *list = strand ? G.list2 : G.list0;
    int2 *out = G.runs + G.run_off[id];
    int n = 0;
    long long pos = 0;          // next position the search may start from (:49 / :57)
    bool in_run = false;
    long long rs = 0, re = 0;   // current run [rs, re): re is also the only position that can extend it
    for (int tl = G.ctg_tile0[ctg]; tl < G.ctg_tile0[ctg + 1]; ++tl) {
        const uint4 c4 = G.tile_cnt[tl];
        const uint32_t base = strand ? G.off2[tl] : G.off0[tl], cnt = strand ? c4.z : c4.x;
        for (uint32_t i = 0; i < cnt; ++i) {
            const long long m = list[base + i];
            if (in_run) {
                if (m < re) continue;                    // inside the run, off phase: skipped by the reference
                if (m == re) { re += G.k; continue; }    // strncmp at :52 succeeds
                out[n++] = make_int2((int)rs, (int)re);  // run ended: strncmp failed at re
                pos = re + 1;
                in_run = false;
            }
            if (m >= pos) { in_run = true; rs = m; re = m + G.k; }
        }
    }
    if (in_run) out[n++] = make_int2((int)rs, (int)re);
    G.n_runs[id] = n;
}

// ---- telowin --------------------------------------------------------------------------------------
__global__ void tw_fill(const int4 *hits /* {bit offset lo, bit offset hi, start, end} */, int64_t n,
                        unsigned long long *bitmap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 hv = hits[i];
    const long long boff = ((long long)(uint32_t)hv.y << 32) | (uint32_t)hv.x;
    long long a = boff + hv.z, b = boff + hv.w;    // [a, b) in bits
    while (a < b) {
        const long long wi = a >> 6;
        const int lo = (int)(a & 63);
        const long long wend = (wi + 1) << 6;
        const int hi = (int)((b < wend ? b : wend) - (wi << 6));   // exclusive bit index within the word, 1..64
        unsigned long long m = (hi == 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
        atomicOr(&bitmap[wi], m);
        a = wend;
    }
}

struct TwArgs {
    const unsigned long long *bitmap;
    const int64_t *bit_off;   // [n_ctg] first bit of contig (multiple of 64)
    const int32_t *ctg_len;
    const int2 *tiles;        // {ctg, first window index}
    double thr;
    int4 *out;                // {ctg, start, end, car}
    unsigned long long *counter;
    uint32_t cap;
};

__device__ __forceinline__ int popc_range(const unsigned long long *bm, long long a, long long b)
{
    int c = 0;
    while (a < b) {
        const long long wi = a >> 6;
        const int lo = (int)(a & 63);
        const long long wend = (wi + 1) << 6;
        const int hi = (int)((b < wend ? b : wend) - (wi << 6));
        const unsigned long long m = (hi == 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
        c += __popcll(bm[wi] & m);
        a = wend;
    }
    return c;
}

// One thread per window start j (every 200 bases, :31); a window is five blocks of 200 bases (the last ones clipped at the contig's
// end, :36), so a workgroup counts its 260 blocks ONCE (4-5 bitmap words each) and a window adds five counts — the first version
// counted the 16-17 words of every window per thread: 0.11 G wave-instructions per 3.16 Gbp step, a quarter of that now.
__global__ __launch_bounds__(256) void tw_scan(TwArgs A)
{
    __builtin_amdgcn_s_setprio(CN_STREAM_PRIO);   // short streaming kernel: issue ahead of a long compute-bound kernel of another stream

    __shared__ int bc[256 + 4];
    __shared__ unsigned long long sw[816];            // the workgroup's 260 blocks of marks: 52 000 bits from any bit of a word
    const int2 tile = A.tiles[blockIdx.x];
    const int ctg = tile.x;
    const int len = A.ctg_len[ctg];
    const long long boff = A.bit_off[ctg];
    // the words once, side by side (round 5; before: every thread its own 4-5 words one after the other — a chain of dependent round trips
    // to memory per workgroup, 0.166 ms for 0.4 GB; 0.138 now), the counts from LDS
    const long long bit0 = boff + (long long)tile.y * 200, bit1 = boff + len < bit0 + 260 * 200 ? boff + len : bit0 + 260 * 200;
    const long long w0 = bit0 >> 6;
    const int nw = bit1 > bit0 ? (int)(((bit1 + 63) >> 6) - w0) : 0;
    for (int k = threadIdx.x; k < nw; k += 256) sw[k] = A.bitmap[w0 + k];
    __syncthreads();
    for (int k = threadIdx.x; k < 260; k += 256) {
        const long long lo = ((long long)tile.y + k) * 200;
        const long long hi = lo + 200 < len ? lo + 200 : len;
        bc[k] = lo < hi ? popc_range(sw, boff + lo - (w0 << 6), boff + hi - (w0 << 6)) : 0;
    }
    __syncthreads();
    const long long j = (long long)tile.y + threadIdx.x;
    const long long i = j * 200;                                   // WINDOW_SIZE / 5, :31
    // the loop of :31-41 visits i = 0, 200, ... up to and including the first i with i + 1000 >= len
    if (i > len) return;
    if (i > 0 && (i - 200) + 1000 >= len) return;
    const long long end = (i + 1000 < len) ? i + 1000 : len;
    const int den = (int)(end - i);                                // :36
    const int t = threadIdx.x;
    const int car = bc[t] + bc[t + 1] + bc[t + 2] + bc[t + 3] + bc[t + 4];
    if ((double)car / den >= A.thr) {                              // :37 (0/0 -> NaN -> false, as in C)
        const unsigned long long idx = atomicAdd(A.counter, 1ull);
        if (idx < A.cap) A.out[idx] = make_int4(ctg, (int)i, (int)end, car);
    }
}

// ---- host side --------------------------------------------------------------------------------------
inline uint8_t c_toupper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

std::string revcomp(const std::string &m)   // src/find_telomere.c:24-42
{
    std::string r(m.size(), 'N');
    for (size_t i = 0; i < m.size(); ++i) {
        char c = m[m.size() - 1 - i];
        r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
    }
    return r;
}

bool has_border(const std::string &m)
{
    for (size_t b = 1; b < m.size(); ++b)
        if (m.compare(0, b, m, m.size() - b, b) == 0) return true;
    return false;
}

struct WinLayout {
    std::vector<int64_t> bit_off;
    std::vector<int2> tiles;
    int64_t n_words = 0;
};

// tiles.hpp: piece j of sequence c = (c, j step)
struct TileFill {
    int2 *out;
    int32_t step;
    __device__ void operator()(int64_t t, int c, int64_t j) const { out[t] = make_int2(c, (int)(j * step)); }
};

WinLayout win_layout_from_lengths(const int32_t *len, int32_t n, const int64_t *byte_off /* may be null */)
{
    WinLayout L;
    int64_t bits = 0;
    for (int32_t c = 0; c < n; ++c) {
        int64_t bo = byte_off ? byte_off[c] : bits;
        L.bit_off.push_back(bo);
        bits = cn_align_up(bo + len[c], 64);
        // windows j = 0..J, J = first j with 200 j + 1000 >= len
        int64_t J = len[c] > 1000 ? (len[c] - 1000 + 199) / 200 : 0;
        for (int64_t j0 = 0; j0 <= J; j0 += 256) L.tiles.push_back(make_int2(c, (int)j0));
    }
    L.n_words = bits / 64 + 4;
    return L;
}

// the windows over a bitmap of marks; d_boff / d_tiles: the layout on the device (n_tiles window tiles)
int run_tw_scan_dev(cornetto_accel_t *h, const unsigned long long *d_bitmap, const int64_t *d_boff, const int2 *d_tiles, size_t n_tiles, const int32_t *d_len,
                    double thr, cornetto_win_t **wins, int64_t *n_wins)
{
    *wins = nullptr;
    *n_wins = 0;
    std::vector<int4> host;
    if (n_tiles) {
        unsigned long long *d_cnt = (unsigned long long *)cn_ws(h, WS_TW_CNT, 16);
        unsigned long long *p_cnt = (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
        if (!d_cnt || !p_cnt) return cn_fail(h, CORNETTO_E_NOMEM, "telowin: workspace allocation failed");
        size_t cap = std::max<size_t>(1u << 16, h->dev[WS_TW_OUT].bytes / sizeof(int4));
        CN_TRACE("tw_scan: small workspaces");
        for (int attempt = 0; attempt < 2; ++attempt) {
            int4 *d_out = (int4 *)cn_ws(h, WS_TW_OUT, cap * sizeof(int4));
            if (!d_out) return cn_fail(h, CORNETTO_E_NOMEM, "telowin: workspace allocation failed");
            CN_TRACE("tw_scan: out workspace");
            CN_HIP(h, hipMemsetAsync(d_cnt, 0, 8, h->stream));
            CN_TRACE("tw_scan: memset queued");
            TwArgs A{d_bitmap, d_boff, d_len, d_tiles, thr, d_out, d_cnt, (uint32_t)std::min<size_t>(cap, 0x7fffffff)};
            CN_LAUNCH(h, "tw_scan", tw_scan<<<dim3((unsigned)n_tiles), dim3(256), 0, h->stream>>>(A));
            CN_TRACE("tw_scan: kernel queued");
            // the count and, with it, the first 16 K windows (an assembly has ~10 K): one round trip instead of two
            const size_t spec = std::min<size_t>(cap, 16384);
            int4 *p_spec = (int4 *)cn_pin(h, PIN_TW, spec * sizeof(int4));
            if (!p_spec) return cn_fail(h, CORNETTO_E_NOMEM, "telowin: pinned allocation failed");
            CN_TRACE("tw_scan: pinned block");
            CN_HIP(h, hipMemcpyAsync(p_cnt, d_cnt, 8, hipMemcpyDeviceToHost, h->stream));
            CN_HIP(h, hipMemcpyAsync(p_spec, d_out, spec * sizeof(int4), hipMemcpyDeviceToHost, h->stream));
            CN_TRACE("tw_scan: queued");
            CN_HIP(h, hipStreamSynchronize(h->stream));
            CN_TRACE("tw_scan: synchronised");
            const unsigned long long cnt = p_cnt[0];
            if (cnt > cap) {   // exact retry with the true size; never a truncated answer
                if (attempt == 1) return cn_fail(h, CORNETTO_E_HIP, "telowin: %llu windows after resizing", cnt);
                cap = (size_t)cnt;
                continue;
            }
            host.resize(cnt);
            if (cnt) memcpy(host.data(), p_spec, std::min<size_t>((size_t)cnt, spec) * sizeof(int4));
            if (cnt > spec) CN_HIP(h, hipMemcpy(host.data() + spec, d_out + spec, ((size_t)cnt - spec) * sizeof(int4), hipMemcpyDeviceToHost));
            break;
        }
    }
    std::sort(host.begin(), host.end(), [](const int4 &a, const int4 &b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
    cornetto_win_t *w = (cornetto_win_t *)malloc((host.size() ? host.size() : 1) * sizeof(cornetto_win_t));
    if (!w) return cn_fail(h, CORNETTO_E_NOMEM, "telowin: host allocation failed");
    for (size_t i = 0; i < host.size(); ++i) {
        w[i].ctg = host[i].x; w[i].start = host[i].y; w[i].end = host[i].z; w[i].car = host[i].w;
    }
    *wins = w;
    *n_wins = (int64_t)host.size();
    return CORNETTO_OK;
}

// the same with the layout still on the host (uploaded into the handle's workspaces)
int run_tw_scan(cornetto_accel_t *h, const unsigned long long *d_bitmap, const WinLayout &L, const int32_t *d_len,
                double thr, cornetto_win_t **wins, int64_t *n_wins)
{
    int64_t *wb = nullptr;
    int2 *wt = nullptr;
    if (!L.tiles.empty()) {
        wb = (int64_t *)cn_ws(h, WS_TW_BOFF, L.bit_off.size() * 8);
        wt = (int2 *)cn_ws(h, WS_TW_TILES, L.tiles.size() * sizeof(int2));
        if (!wb || !wt) return cn_fail(h, CORNETTO_E_NOMEM, "telowin: workspace allocation failed");
        CN_HIP(h, hipMemcpyAsync(wb, L.bit_off.data(), L.bit_off.size() * 8, hipMemcpyHostToDevice, h->stream));
        CN_HIP(h, hipMemcpyAsync(wt, L.tiles.data(), L.tiles.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
    }
    return run_tw_scan_dev(h, d_bitmap, wb, wt, L.tiles.size(), d_len, thr, wins, n_wins);   // (synchronises before it returns: `L` outlives the uploads)
}

int telowin_from_hits(cornetto_accel_t *h, const cornetto_hit_t *hits, int64_t n_hits, const int32_t *ctg_len,
                      int32_t n_ctg, double thr_adj, cornetto_win_t **wins, int64_t *n_wins)
{
    WinLayout L = win_layout_from_lengths(ctg_len, n_ctg, nullptr);
    std::vector<int4> hv;
    hv.reserve((size_t)n_hits);
    for (int64_t i = 0; i < n_hits; ++i) {
        const cornetto_hit_t &x = hits[i];
        if (x.ctg < 0 || x.ctg >= n_ctg) return cn_fail(h, CORNETTO_E_ARG, "telowin: hit %lld names contig %d of %d", (long long)i, x.ctg, n_ctg);
        if (x.start < 0 || x.end > ctg_len[x.ctg])
            return cn_fail(h, CORNETTO_E_ARG, "telowin: hit %lld [%d,%d) lies outside contig %d of length %d", (long long)i, x.start, x.end, x.ctg, ctg_len[x.ctg]);
        if (x.start >= x.end) continue;   // the reference's marking loop does nothing
        const int64_t bo = L.bit_off[x.ctg];
        hv.push_back(make_int4((int)(uint32_t)(bo & 0xFFFFFFFFll), (int)(uint32_t)(bo >> 32), x.start, x.end));
    }
    unsigned long long *d_bm = (unsigned long long *)cn_ws(h, WS_TW_BITMAP, (size_t)L.n_words * 8);
    int4 *d_hits = (int4 *)cn_ws(h, WS_TW_HITS, hv.size() * sizeof(int4));
    int32_t *d_len = (int32_t *)cn_ws(h, WS_TW_LEN, (size_t)(n_ctg > 0 ? n_ctg : 1) * 4);
    if (!d_bm || !d_hits || !d_len) return cn_fail(h, CORNETTO_E_NOMEM, "telowin: workspace allocation failed");
    CN_HIP(h, hipMemsetAsync(d_bm, 0, (size_t)L.n_words * 8, h->stream));
    if (n_ctg) CN_HIP(h, hipMemcpyAsync(d_len, ctg_len, (size_t)n_ctg * 4, hipMemcpyHostToDevice, h->stream));
    if (!hv.empty()) {
        CN_HIP(h, hipMemcpyAsync(d_hits, hv.data(), hv.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
        const unsigned nb = (unsigned)((hv.size() + 255) / 256);
        CN_LAUNCH(h, "tw_fill", tw_fill<<<dim3(nb), dim3(256), 0, h->stream>>>(d_hits, (int64_t)hv.size(), d_bm));
    }
    int rc = run_tw_scan(h, d_bm, L, d_len, thr_adj, wins, n_wins);
    if (hipStreamSynchronize(h->stream) != hipSuccess && rc == CORNETTO_OK) rc = CORNETTO_E_HIP;   // hv / ctg_len are caller memory
    return rc;
}

// the whole telofind pass; optionally leaves the mark bitmap on the device (unbordered motifs)
// the window layout of a resident assembly (it depends on the contig table only): built once, kept on the device with the object.  The marks of
// the contigs lie back to back in the bitmap (a wrapped subset of a larger buffer — one rank's share — costs its own bases, not the span)
int ensure_tw_layout(cornetto_accel_t *h, cornetto_asm_t *am)
{
    if (am->tw_n_words >= 0) return CORNETTO_OK;
    // as win_layout_from_lengths(): the marks of the contigs back to back, every contig from a multiple of 64 bits; window tile j0 of contig c =
    // windows j0 .. j0 + 255 of its J + 1 (J = the first j with 200 j + 1000 >= len).  The tiles are written on the device (tiles.hpp).
    am->tw_boff.resize((size_t)am->n);
    int64_t bits = 0;
    for (int32_t c = 0; c < am->n; ++c) {
        am->tw_boff[c] = bits;
        bits = cn_align_up(bits + am->len[c], 64);
    }
    const int64_t ntl = cntiles::prefix(h, am->tw_pref, am->n, [&](int32_t c) {
        const int64_t J = am->len[c] > 1000 ? ((int64_t)am->len[c] - 1000 + 199) / 200 : 0;
        return J / 256 + 1;
    });
    if (ntl < 0) return cn_fail(h, CORNETTO_E_NOMEM, "telo_scan: device allocation failed");
    if (am->n > 0) {
        const bool ok = cn_obj_malloc(h, (void **)&am->d_tw_boff, (size_t)am->n * 8) == hipSuccess && cn_obj_malloc(h, (void **)&am->d_tw_tiles, ((size_t)ntl + 1) * sizeof(int2)) == hipSuccess &&
                        hipMemcpyAsync(am->d_tw_boff, am->tw_boff.data(), (size_t)am->n * 8, hipMemcpyHostToDevice, h->stream) == hipSuccess;
        if (!ok) return cn_fail(h, CORNETTO_E_NOMEM, "telo_scan: device allocation failed");
        if (ntl > 0) {
            cntiles::fill<<<dim3((unsigned)((ntl + 255) / 256)), dim3(256), 0, h->stream>>>(am->tw_pref.dev, am->n, ntl, TileFill{am->d_tw_tiles, 256});
            CN_HIP(h, hipGetLastError());
        }
    }
    am->tw_n_tiles = ntl;
    am->tw_n_words = bits / 64 + 4;
    return CORNETTO_OK;
}

int telofind_impl(cornetto_accel_t *h, const cornetto_asm_t *a_in, const char *motif_c, cornetto_hit_t **hits,
                  int64_t *n_hits, bool want_bitmap_req, unsigned long long **bitmap_out, bool *bitmap_valid)
{
    if (!h || !a_in || !motif_c) return cn_fail(h, CORNETTO_E_ARG, "telofind: bad argument");
    cornetto_asm_t *a = const_cast<cornetto_asm_t *>(a_in);   // only the cached tile table is touched
    const std::string motif(motif_c);
    const int k = (int)motif.size();
    if (k < 1) return cn_fail(h, CORNETTO_E_ARG, "telofind: empty motif (the reference never terminates on it)");
    if (k > (1 << 20)) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "telofind: motif of %d bytes; at most %d are supported", k, 1 << 20);
    const bool long_motif = k > MAX_MOTIF;   // beyond the 32-bit automaton: compared byte by byte (tf_scan<-1>), runs by the sequential rule
    CN_HIP(h, hipSetDevice(h->device));
    if (hits) { *hits = nullptr; *n_hits = 0; }
    if (bitmap_valid) *bitmap_valid = false;
    a->tf_est_cnt[0] = a->tf_est_cnt[1] = a->tf_est_cnt[2] = a->tf_est_cnt[3] = -1;

    const std::string rc = revcomp(motif);
    const bool bordered = long_motif || has_border(motif) || has_border(rc);
    const int H = long_motif ? 0 : (k <= 8 ? 7 : (k <= 16 ? 15 : 31));
    // shift-and tables: motif position j -> bit (31 - H + j); positions k..H are wildcards
    std::vector<uint2> lut(256);
    for (int c = 0; c < 256; ++c) {
        uint32_t f = 0, r = 0;
        const uint8_t u = c_toupper((uint8_t)c);      // src/find_telomere.c:76-81: sequence upper-cased, motif not
        for (int j = 0; j <= H && !long_motif; ++j) {
            const uint32_t bit = 1u << (31 - H + j);
            if (j >= k) { f |= bit; r |= bit; continue; }
            if (u == (uint8_t)motif[j]) f |= bit;
            if (u == (uint8_t)rc[j]) r |= bit;
        }
        if (H == 7 && !long_motif) {   // packed form (tf_scan<7>): forward automaton in bits 16-23, reverse in 0-7, the delay lines all ones
            f = ((f >> 24) << 16) | ((r >> 24) & 0xFFu) | 0xFF00FF00u;
            r = 0;
        }
        lut[c] = make_uint2(f, r);
    }
    // tiles in contig order (cached with the assembly)
    if (a->tf_n_tiles < 0) {
        // on the device (tiles.hpp): tile j of contig c = (c, j TF_TILE); the first tile of every contig stays on the host as well (32-bit: the
        // kernels' tile indices)
        const int64_t ntl = cntiles::prefix(h, a->tf_pref, a->n, [&](int32_t c) { return ((int64_t)a->len[c] + TF_TILE - 1) / TF_TILE; });
        if (ntl < 0) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: device allocation failed");
        if (ntl > 0x7fffffffll) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "telofind: too many tiles");
        a->tf_ctg_tile0.resize((size_t)a->n + 1);
        for (int32_t c = 0; c <= a->n; ++c) a->tf_ctg_tile0[c] = (int32_t)a->tf_pref.host[c];
        if (ntl > 0) {
            if (cn_obj_malloc(h, (void **)&a->d_tf_tiles, (size_t)ntl * sizeof(int2)) != hipSuccess || cn_obj_malloc(h, (void **)&a->d_tf_ct0, ((size_t)a->n + 1) * 4) != hipSuccess)
                return cn_fail(h, CORNETTO_E_NOMEM, "telofind: device allocation failed");
            cntiles::fill<<<dim3((unsigned)((ntl + 255) / 256)), dim3(256), 0, h->stream>>>(a->tf_pref.dev, a->n, ntl, TileFill{a->d_tf_tiles, (int32_t)TF_TILE});
            CN_HIP(h, hipGetLastError());
            CN_HIP(h, hipMemcpyAsync(a->d_tf_ct0, a->tf_ctg_tile0.data(), ((size_t)a->n + 1) * 4, hipMemcpyHostToDevice, h->stream));   // (a member: alive)
        }
        a->tf_n_tiles = ntl;
    }
    const std::vector<int32_t> &ctg_tile0 = a->tf_ctg_tile0;
    const size_t nt = (size_t)a->tf_n_tiles;
    CN_TRACE("telofind: tile table");

    cornetto_hit_t *out = nullptr;
    int64_t n_out = 0;
    if (nt > 0) {
        const size_t np = 4 * ((nt + 4095) / 4096) + 4;   // (scan partials of the four counters)
        uint2 *d_lut = (uint2 *)cn_ws(h, WS_TF_LUT, 256 * sizeof(uint2) + 2 * (size_t)k + 16);
        uint8_t *d_mot = reinterpret_cast<uint8_t *>(d_lut + 256);
        // small device block: totals[4] u64, ovf u32, err u32
        unsigned long long *d_cnt = (unsigned long long *)cn_ws(h, WS_TF_CNT, 64);
        uint4 *d_tc = (uint4 *)cn_ws(h, WS_TF_TC, nt * sizeof(uint4));
        // 4 scanned offset arrays + scan partials
        uint32_t *d_off = (uint32_t *)cn_ws(h, WS_TF_TB, (4 * nt + np) * sizeof(uint32_t));
        unsigned long long *p_cnt = (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
        if (!d_lut || !d_cnt || !d_tc || !d_off || !p_cnt) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: workspace allocation failed");
        uint32_t *d_offq[4] = {d_off, d_off + nt, d_off + 2 * nt, d_off + 3 * nt}, *d_part = d_off + 4 * nt;
        uint32_t *d_ovf = reinterpret_cast<uint32_t *>(d_cnt + 4), *d_err = d_ovf + 1;
        const std::string both = motif + rc;          // (pageable host memory: the copy has left it when the call returns)
        if (h->tf_lut_key != both || h->tf_lut_ptr != d_lut) {   // (the tables of the last call's motif are still there otherwise)
            h->tf_lut_key.clear();
            CN_HIP(h, hipMemcpyAsync(d_lut, lut.data(), 256 * sizeof(uint2), hipMemcpyHostToDevice, h->stream));
            if (long_motif) CN_HIP(h, hipMemcpyAsync(d_mot, both.data(), 2 * (size_t)k, hipMemcpyHostToDevice, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));   // `lut` and `both` are locals
            h->tf_lut_key = both;
            h->tf_lut_ptr = d_lut;
        }
        CN_HIP(h, hipMemsetAsync(d_cnt, 0, 64, h->stream));
        const bool want_bitmap = want_bitmap_req && !bordered;
        unsigned long long *d_bitmap = nullptr;
        if (want_bitmap) {
            CN_TRY(ensure_tw_layout(h, a));
            const size_t words = (size_t)a->tw_n_words;
            d_bitmap = (unsigned long long *)cn_ws(h, WS_TF_BITMAP, words * 8);
            if (!d_bitmap) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: bitmap allocation failed");
            // every word of a contig is written by the kernel; the memset only covers the padding behind the last contig, and that stays zero in
            // the block from one call for THIS assembly's layout to the next: filled once per (assembly, block).  (Round 4, five bench runs each
            // way: 7.84-7.90 ms per 3.16 Gbp step without the 395 MB fill, 7.87-7.94 with it, one slow run of 8.2 in either; the 1/8 share of the
            // assembly 0.31 instead of 0.39 ms per call.  Round 3 had measured the opposite with the telomere scan first in the step.)
            static const int refill = CN_DEV_INT("CORNETTO_TF_BITMAP_REFILL", 0);
            if (refill || !(h->tf_bm_uid == a->uid && h->tf_bm_ptr == d_bitmap && h->tf_bm_words == words)) {
                h->tf_bm_uid = 0;
                CN_HIP(h, hipMemsetAsync(d_bitmap, 0, words * 8, h->stream));
                h->tf_bm_uid = a->uid; h->tf_bm_ptr = d_bitmap; h->tf_bm_words = words;
            }
            if (bitmap_out) *bitmap_out = d_bitmap;
        }
        auto launch = [&](const TfArgs &A) -> int {
            if (long_motif) CN_LAUNCH(h, "tf_scan", tf_scan<-1><<<dim3((unsigned)((nt + TF_NT - 1) / TF_NT)), dim3(TF_THREADS), 0, h->stream>>>(A));
            else if (H == 7) CN_LAUNCH(h, "tf_scan", tf_scan<7><<<dim3((unsigned)((nt + TF_NT - 1) / TF_NT)), dim3(TF_THREADS), 0, h->stream>>>(A));
            else if (H == 15) CN_LAUNCH(h, "tf_scan", tf_scan<15><<<dim3((unsigned)((nt + TF_NT - 1) / TF_NT)), dim3(TF_THREADS), 0, h->stream>>>(A));
            else CN_LAUNCH(h, "tf_scan", tf_scan<31><<<dim3((unsigned)((nt + TF_NT - 1) / TF_NT)), dim3(TF_THREADS), 0, h->stream>>>(A));
            return CORNETTO_OK;
        };
        TfArgs A{};
        A.bases = a->d_bases; A.ctg_off = a->d_off; A.ctg_len = a->d_len; A.tiles = a->d_tf_tiles; A.lut = d_lut; A.k = k; A.mot = d_mot;
        A.bordered = bordered ? 1 : 0; A.bitmap = want_bitmap ? d_bitmap : nullptr; A.bm_off = a->d_tw_boff; A.tile_cnt = d_tc; A.ovf = d_ovf; A.n_tiles = (int64_t)nt;
        // pass 1: single pass into fixed rows (unbordered motif, hits wanted), or counts only
        const bool rows_mode = hits && !bordered;
        int32_t *d_rows[4] = {nullptr, nullptr, nullptr, nullptr};
        if (rows_mode) {
            for (int q = 0; q < 4; ++q) {
                d_rows[q] = (int32_t *)cn_ws(h, WS_TF_L0 + q, nt * TF_ROW * sizeof(int32_t));
                if (!d_rows[q]) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: workspace allocation failed");
            }
            A.mode = 2;
            A.list0 = d_rows[0]; A.list1 = d_rows[1]; A.list2 = d_rows[2]; A.list3 = d_rows[3];
        } else {
            A.mode = 0;
        }
        CN_TRACE("telofind: workspaces, tables uploaded");
        CN_TRY(launch(A));
        CN_TRACE("telofind: tf_scan queued");
        if (bitmap_valid) *bitmap_valid = want_bitmap;
        if (hits) {
            // place of every tile in the dense, contig-ordered lists + list totals
            CN_TRY(cnscan::exclusive_u32_multi(h, "tf_order", reinterpret_cast<const uint32_t *>(d_tc), (int64_t)nt, 4, 4, d_offq, d_part, d_cnt));
            CN_HIP(h, hipMemcpyAsync(p_cnt, d_cnt, 64, hipMemcpyDeviceToHost, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));   // also covers the `lut` upload
            CN_TRACE("telofind: counts on the host");
            unsigned long long cnt[4] = {p_cnt[0], p_cnt[1], p_cnt[2], p_cnt[3]};
            const uint32_t ovf = (uint32_t)(p_cnt[4] & 0xFFFFFFFFull);
            for (int q = 0; q < 4; ++q)
                if (cnt[q] > 0x7fffffffull) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "telofind: %llu list entries", cnt[q]);
            // dense lists: one workspace, 4 segments
            const size_t seg[4] = {(size_t)cnt[0], (size_t)cnt[1], (size_t)cnt[2], (size_t)cnt[3]};
            int32_t *d_dense = (int32_t *)cn_ws(h, WS_TF_RUNS, (seg[0] + seg[1] + seg[2] + seg[3] + 4) * sizeof(int32_t));
            if (!d_dense) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: workspace allocation failed");
            int32_t *d_list[4] = {d_dense, d_dense + seg[0], d_dense + seg[0] + seg[1], d_dense + seg[0] + seg[1] + seg[2]};
            if (rows_mode && ovf <= TF_ROW) {
                hipEvent_t ea = cn_event(h), eb = cn_event(h);
                (void)hipEventRecord(ea, h->stream);
                tf_gather<<<dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, h->stream>>>(d_rows[0], d_rows[1], d_rows[2], d_rows[3], d_tc, d_offq[0], d_offq[1],
                                                                                       d_offq[2], d_offq[3], d_list[0], d_list[1], d_list[2], d_list[3],
                                                                                       make_uint4(~0u, ~0u, ~0u, ~0u), (int64_t)nt);
                (void)hipEventRecord(eb, h->stream);
                h->recs.push_back(cornetto_accel::Rec{"tf_gather", ea, eb});
                CN_HIP(h, hipGetLastError());
            } else {
                // second pass writes straight to the scanned offsets (bordered motif, or a tile overflowed its row)
                A.mode = 1;
                A.bitmap = nullptr;
                A.list0 = d_list[0]; A.list1 = d_list[1]; A.list2 = d_list[2]; A.list3 = d_list[3];
                A.off0 = d_offq[0]; A.off1 = d_offq[1]; A.off2 = d_offq[2]; A.off3 = d_offq[3];
                CN_TRY(launch(A));
            }
            int32_t *d_ct0 = (int32_t *)cn_ws(h, WS_TF_NRUNS, ((size_t)a->n + 1) * 4 + 2 * (size_t)a->n * 4 + 16);
            if (!d_ct0) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: workspace allocation failed");
            if (bordered) CN_HIP(h, hipMemcpyAsync(d_ct0, ctg_tile0.data(), ((size_t)a->n + 1) * 4, hipMemcpyHostToDevice, h->stream));   // (the greedy rule writes behind it)
            if (!bordered) {
                if (cnt[0] != cnt[1] || cnt[2] != cnt[3])
                    return cn_fail(h, CORNETTO_E_HIP, "telofind: head/tail count mismatch (%llu/%llu, %llu/%llu)", cnt[0], cnt[1], cnt[2], cnt[3]);
                const size_t tot = (size_t)(cnt[0] + cnt[2]);
                uint32_t *d_coff = (uint32_t *)cn_ws(h, WS_TF_ROFF, 4 * ((size_t)a->n + 1) * 4);
                cornetto_hit_t *d_hits = (cornetto_hit_t *)cn_ws(h, WS_TF_HITS, tot * sizeof(cornetto_hit_t));
                if (!d_coff || !d_hits) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: workspace allocation failed");
                out = (cornetto_hit_t *)cn_result_alloc((tot ? tot : 1) * sizeof(cornetto_hit_t));
                if (!out) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: host allocation failed");
                hipEvent_t ea = cn_event(h), eb = cn_event(h);
                (void)hipEventRecord(ea, h->stream);
                tf_ctgoff<<<dim3((unsigned)((a->n + 256) / 256)), dim3(256), 0, h->stream>>>(a->d_tf_ct0, a->n, (int64_t)nt, d_offq[0], d_offq[1], d_offq[2],
                                                                                           d_offq[3], d_cnt, d_coff, d_err);
                if (tot)
                    tf_pair<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream>>>(d_list[0], d_list[1], d_list[2], d_list[3], d_coff, a->n,
                                                                                             (uint32_t)cnt[0], (uint32_t)cnt[2], k, d_hits);
                (void)hipEventRecord(eb, h->stream);
                h->recs.push_back(cornetto_accel::Rec{"tf_pair", ea, eb});
                bool ok = hipGetLastError() == hipSuccess;
                ok = ok && hipMemcpyAsync(p_cnt, d_cnt, 64, hipMemcpyDeviceToHost, h->stream) == hipSuccess;
                if (tot) ok = ok && cn_result_d2h(h, out, d_hits, tot * sizeof(cornetto_hit_t)) == hipSuccess;
                ok = ok && hipStreamSynchronize(h->stream) == hipSuccess;
                if (!ok || (p_cnt[4] >> 32) != 0) {
                    cn_result_quiesce(h);
                    cornetto_free(out);
                    return cn_fail(h, CORNETTO_E_HIP, "telofind: pairing run heads with tails failed%s", ok ? " (a contig has unequal heads and tails)" : "");
                }
                n_out = (int64_t)tot;
                for (int q = 0; q < 4; ++q) a->tf_est_cnt[q] = (int64_t)cnt[q];
            } else {
                // sequential greedy rule on the device over the dense match lists
                std::vector<uint4> tc(nt);
                CN_HIP(h, hipMemcpyAsync(tc.data(), d_tc, nt * sizeof(uint4), hipMemcpyDeviceToHost, h->stream));
                CN_HIP(h, hipStreamSynchronize(h->stream));
                std::vector<int64_t> run_off(2 * (size_t)a->n + 1, 0);
                for (int32_t c = 0; c < a->n; ++c) {
                    int64_t mf = 0, mr = 0;
                    for (int32_t t = ctg_tile0[c]; t < ctg_tile0[c + 1]; ++t) { mf += tc[t].x; mr += tc[t].z; }
                    run_off[2 * c + 1] = run_off[2 * c] + mf;
                    run_off[2 * c + 2] = run_off[2 * c + 1] + mr;
                }
                const int64_t tot = run_off[2 * (size_t)a->n];
                int64_t *d_roff = (int64_t *)cn_ws(h, WS_TF_ROFF, run_off.size() * 8);
                int2 *d_runs = (int2 *)cn_ws(h, WS_TF_L1, (size_t)tot * sizeof(int2));
                int32_t *d_nruns = d_ct0 + (a->n + 1);
                if (!d_roff || !d_runs) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: device allocation failed");
                CN_HIP(h, hipMemcpyAsync(d_roff, run_off.data(), run_off.size() * 8, hipMemcpyHostToDevice, h->stream));
                GreedyArgs G{d_list[0], d_list[2], d_offq[0], d_offq[2], d_tc, d_ct0, d_roff, a->n, k, d_runs, d_nruns};
                const unsigned nb = (unsigned)((2 * a->n + 63) / 64);
                CN_LAUNCH(h, "tf_greedy", tf_greedy<<<dim3(nb), dim3(64), 0, h->stream>>>(G));
                std::vector<int2> runs((size_t)tot);
                std::vector<int32_t> nr(2 * (size_t)a->n);
                CN_HIP(h, hipStreamSynchronize(h->stream));
                if (tot) CN_HIP(h, hipMemcpy(runs.data(), d_runs, (size_t)tot * sizeof(int2), hipMemcpyDeviceToHost));
                CN_HIP(h, hipMemcpy(nr.data(), d_nruns, nr.size() * 4, hipMemcpyDeviceToHost));
                int64_t total_runs = 0;
                for (int32_t v : nr) total_runs += v;
                out = (cornetto_hit_t *)malloc((size_t)(total_runs + 1) * sizeof(cornetto_hit_t));
                if (!out) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: host allocation failed");
                for (int32_t c = 0; c < a->n; ++c)
                    for (int strand = 0; strand < 2; ++strand)
                        for (int32_t i = 0; i < nr[2 * c + strand]; ++i) {
                            const int2 r = runs[run_off[2 * c + strand] + i];
                            out[n_out++] = cornetto_hit_t{c, strand, r.x, r.y};
                        }
            }
        } else {
            CN_HIP(h, hipStreamSynchronize(h->stream));   // `lut` is a local
        }
    }
    if (hits) {
        if (!out) {
            out = (cornetto_hit_t *)malloc(sizeof(cornetto_hit_t));
            if (!out) return cn_fail(h, CORNETTO_E_NOMEM, "telofind: host allocation failed");
        }
        *hits = out;
        *n_hits = n_out;
    }
    return CORNETTO_OK;
}

}  // namespace

extern "C" {

double cornetto_telowin_threshold(double threshold, double identity_percent)
{
    const double identity = identity_percent / 100;   // src/telomere_windows.c:53
    return threshold * pow(identity, 6);              // :54
}

int cornetto_telofind(cornetto_accel_t *h, const cornetto_asm_t *a, const char *motif, cornetto_hit_t **hits,
                      int64_t *n_hits)
{
    if (!h || !hits || !n_hits) return cn_fail(h, CORNETTO_E_ARG, "telofind: bad argument");
    cn_timing_begin(h);
    int rc = telofind_impl(h, a, motif, hits, n_hits, false, nullptr, nullptr);
    cn_timing_end(h);
    return rc;
}

int cornetto_telowin(cornetto_accel_t *h, const cornetto_hit_t *hits, int64_t n_hits, const int32_t *ctg_len,
                     int32_t n_ctg, double thr_adj, cornetto_win_t **wins, int64_t *n_wins)
{
    if (!h || !wins || !n_wins || n_hits < 0 || n_ctg < 0 || (n_hits > 0 && !hits) || (n_ctg > 0 && !ctg_len))
        return cn_fail(h, CORNETTO_E_ARG, "telowin: bad argument");
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    int rc = telowin_from_hits(h, hits, n_hits, ctg_len, n_ctg, thr_adj, wins, n_wins);
    cn_timing_end(h);
    return rc;
}

int cornetto_telo_scan(cornetto_accel_t *h, const cornetto_asm_t *a, const char *motif, double thr_adj,
                       cornetto_hit_t **hits, int64_t *n_hits, cornetto_win_t **wins, int64_t *n_wins)
{
    if (!h || !a || !wins || !n_wins || (hits && !n_hits)) return cn_fail(h, CORNETTO_E_ARG, "telo_scan: bad argument");
    cn_timing_begin(h);
    const int rc = cn_telo_scan_impl(h, a, motif, thr_adj, hits, n_hits, wins, n_wins);
    cn_timing_end(h);
    return rc;
}

}  // extern "C"

namespace {
std::string tf_est_key_of(const char *motif, double thr_adj)
{
    char b[40];
    unsigned long long bits;
    memcpy(&bits, &thr_adj, 8);
    snprintf(b, sizeof(b), "|%016llx", bits);
    return std::string(motif ? motif : "") + b;
}
}  // namespace

int cn_telo_scan_impl(cornetto_accel_t *h, const cornetto_asm_t *a, const char *motif, double thr_adj, cornetto_hit_t **hits, int64_t *n_hits, cornetto_win_t **wins,
                      int64_t *n_wins)
{
    // (Round 5 measured the one-go form of cn_telo_spec_queue here as well — one synchronisation instead of three for a repeated scan: the scan
    // itself no shorter inside the bench step, 2.53 against 2.54 ms, and its 26 MB of telomere runs copied at the very end instead of beside the
    // window kernel and the last two waits: 7.24-7.30 against 6.82-6.89 ms per step.  Not kept; cornetto_panel_step uses that form.)
    unsigned long long *d_bitmap = nullptr;
    bool valid = false;
    cornetto_hit_t *hh = nullptr;
    int64_t nh = 0;
    // a bordered motif needs the runs themselves to build the marks, so the hits are always fetched then
    const std::string m(motif ? motif : "");
    const bool need_hits = hits || has_border(m) || has_border(revcomp(m));
    CN_TRACE("telo_scan: enter");
    int rc = telofind_impl(h, a, motif, need_hits ? &hh : nullptr, need_hits ? &nh : nullptr, true, &d_bitmap, &valid);
    CN_TRACE("telo_scan: telofind done");
    if (CN_DEV_INT("CORNETTO_TRACE_SYNC", 0)) { (void)hipStreamSynchronize(h->stream); CN_TRACE("telo_scan: (trace) stream drained"); }
    if (rc == CORNETTO_OK) {
        if (valid) {
            cornetto_asm_t *am = const_cast<cornetto_asm_t *>(a);   // (its window layout was built with the bitmap: ensure_tw_layout)
            if (rc == CORNETTO_OK) rc = run_tw_scan_dev(h, d_bitmap, am->d_tw_boff, am->d_tw_tiles, (size_t)am->tw_n_tiles, a->d_len, thr_adj, wins, n_wins);
        } else {
            rc = telowin_from_hits(h, hh, nh, a->len.data(), a->n, thr_adj, wins, n_wins);
        }
    }
    CN_TRACE("telo_scan: windows done");
    {   // what the next scan of this assembly with this motif and threshold may size itself by (cn_telo_spec_queue); the list totals were
        // noted by telofind_impl
        cornetto_asm_t *am = const_cast<cornetto_asm_t *>(a);
        if (rc == CORNETTO_OK && valid && hits && am->tf_est_cnt[0] >= 0) {
            am->tf_est_key = tf_est_key_of(motif, thr_adj);
            am->tf_est_wins = *n_wins;
        } else {
            am->tf_est_key.clear();
        }
    }
    if (rc != CORNETTO_OK || !hits) {
        // (a lazy handle may still be copying into hh on its copy stream; hh is library memory — the pinned pool for large results)
        cn_result_quiesce(h);
        cornetto_free(hh);
    } else {
        *hits = hh;
        *n_hits = nh;
    }
    return rc;
}

// ---- the fused scan queued without its synchronisations (internal.hpp) ----------------------------------------------------------------
int cn_telo_spec_queue(cornetto_accel_t *h, cornetto_asm_t *a, const char *motif_c, double thr_adj, unsigned long long *p_cnt, CnTeloSpec *S)
{
    S->queued = false;
    S->p_cnt = p_cnt;
    if (!motif_c || a->tf_est_key.empty() || a->tf_est_key != tf_est_key_of(motif_c, thr_adj) || a->tf_est_cnt[0] < 0 || a->tf_est_wins < 0) return CORNETTO_OK;
    const std::string motif(motif_c);
    const int k = (int)motif.size();
    if (k < 1 || k > MAX_MOTIF || a->tf_n_tiles <= 0 || a->tw_n_words < 0 || a->tw_n_tiles <= 0) return CORNETTO_OK;
    const std::string rc = revcomp(motif), both = motif + rc;
    if (has_border(motif) || has_border(rc)) return CORNETTO_OK;
    const size_t nt = (size_t)a->tf_n_tiles;
    const size_t np = 4 * ((nt + 4095) / 4096) + 4;
    uint2 *d_lut = (uint2 *)cn_ws(h, WS_TF_LUT, 256 * sizeof(uint2) + 2 * (size_t)k + 16);
    if (!d_lut || h->tf_lut_key != both || h->tf_lut_ptr != d_lut) return CORNETTO_OK;      // (the tables of this motif are not on the device: the exact call uploads them)
    const int H = k <= 8 ? 7 : (k <= 16 ? 15 : 31);
    unsigned long long *d_cnt = (unsigned long long *)cn_ws(h, WS_TF_CNT, 64);
    uint4 *d_tc = (uint4 *)cn_ws(h, WS_TF_TC, nt * sizeof(uint4));
    uint32_t *d_off = (uint32_t *)cn_ws(h, WS_TF_TB, (4 * nt + np) * sizeof(uint32_t));
    const size_t words = (size_t)a->tw_n_words;
    unsigned long long *d_bitmap = (unsigned long long *)cn_ws(h, WS_TF_BITMAP, words * 8);
    if (!d_cnt || !d_tc || !d_off || !d_bitmap) return cn_fail(h, CORNETTO_E_NOMEM, "telo_scan: workspace allocation failed");
    uint32_t *d_offq[4] = {d_off, d_off + nt, d_off + 2 * nt, d_off + 3 * nt}, *d_part = d_off + 4 * nt;
    uint32_t *d_ovf = reinterpret_cast<uint32_t *>(d_cnt + 4), *d_err = d_ovf + 1;
    int32_t *d_rows[4];
    for (int q = 0; q < 4; ++q) {
        d_rows[q] = (int32_t *)cn_ws(h, WS_TF_L0 + q, nt * TF_ROW * sizeof(int32_t));
        if (!d_rows[q]) return cn_fail(h, CORNETTO_E_NOMEM, "telo_scan: workspace allocation failed");
    }
    // lists, pairing and copies by the counts of last time plus head room
    size_t seg[4], tot_cap = 0;
    const int force = CN_DEV_INT("CORNETTO_STEP_EST_FORCE", 0);                          // (tests: estimates that do not hold)
    for (int q = 0; q < 4; ++q) {
        seg[q] = (size_t)a->tf_est_cnt[q] + (size_t)a->tf_est_cnt[q] / 16 + 1024;
        if (force) seg[q] = (size_t)std::max(1, force);
        if (seg[q] > 0x7fffffffull) return CORNETTO_OK;
        S->seg_cap[q] = seg[q];
    }
    tot_cap = seg[0] + seg[2];
    int32_t *d_dense = (int32_t *)cn_ws(h, WS_TF_RUNS, (seg[0] + seg[1] + seg[2] + seg[3] + 4) * sizeof(int32_t));
    uint32_t *d_coff = (uint32_t *)cn_ws(h, WS_TF_ROFF, 4 * ((size_t)a->n + 1) * 4);
    cornetto_hit_t *d_hits = (cornetto_hit_t *)cn_ws(h, WS_TF_HITS, tot_cap * sizeof(cornetto_hit_t));
    unsigned long long *d_twcnt = (unsigned long long *)cn_ws(h, WS_TW_CNT, 16);
    const size_t win_ws_cap = std::max<size_t>(1u << 16, h->dev[WS_TW_OUT].bytes / sizeof(int4));
    int4 *d_wout = (int4 *)cn_ws(h, WS_TW_OUT, win_ws_cap * sizeof(int4));
    const size_t win_cap = force ? std::min<size_t>(win_ws_cap, (size_t)std::max(1, force / 64))
                                 : std::min<size_t>(win_ws_cap, (size_t)a->tf_est_wins + (size_t)a->tf_est_wins / 8 + 1024);
    int4 *p_wins = (int4 *)cn_pin(h, PIN_TW, win_cap * sizeof(int4));
    if (!d_dense || !d_coff || !d_hits || !d_twcnt || !d_wout || !p_wins) return cn_fail(h, CORNETTO_E_NOMEM, "telo_scan: workspace allocation failed");
    int32_t *d_list[4] = {d_dense, d_dense + seg[0], d_dense + seg[0] + seg[1], d_dense + seg[0] + seg[1] + seg[2]};
    cornetto_hit_t *out = (cornetto_hit_t *)cn_result_alloc(tot_cap * sizeof(cornetto_hit_t));
    if (!out) return cn_fail(h, CORNETTO_E_NOMEM, "telo_scan: host allocation failed");
    struct Guard {                                   // (an error below gives the buffer back)
        cornetto_accel_t *h;
        cornetto_hit_t **o;
        ~Guard() { if (*o) { cn_result_quiesce(h); cornetto_free(*o); } }
    } guard{h, &out};

    CN_HIP(h, hipMemsetAsync(d_cnt, 0, 64, h->stream));
    static const int refill = CN_DEV_INT("CORNETTO_TF_BITMAP_REFILL", 0);
    if (refill || !(h->tf_bm_uid == a->uid && h->tf_bm_ptr == d_bitmap && h->tf_bm_words == words)) {
        h->tf_bm_uid = 0;
        CN_HIP(h, hipMemsetAsync(d_bitmap, 0, words * 8, h->stream));
        h->tf_bm_uid = a->uid; h->tf_bm_ptr = d_bitmap; h->tf_bm_words = words;
    }
    TfArgs A{};
    A.bases = a->d_bases; A.ctg_off = a->d_off; A.ctg_len = a->d_len; A.tiles = a->d_tf_tiles; A.lut = d_lut; A.k = k; A.mot = reinterpret_cast<uint8_t *>(d_lut + 256);
    A.bordered = 0; A.bitmap = d_bitmap; A.bm_off = a->d_tw_boff; A.tile_cnt = d_tc; A.ovf = d_ovf; A.n_tiles = (int64_t)nt;
    A.mode = 2;
    A.list0 = d_rows[0]; A.list1 = d_rows[1]; A.list2 = d_rows[2]; A.list3 = d_rows[3];
    if (H == 7) CN_LAUNCH(h, "tf_scan", tf_scan<7><<<dim3((unsigned)((nt + TF_NT - 1) / TF_NT)), dim3(TF_THREADS), 0, h->stream>>>(A));
    else if (H == 15) CN_LAUNCH(h, "tf_scan", tf_scan<15><<<dim3((unsigned)((nt + TF_NT - 1) / TF_NT)), dim3(TF_THREADS), 0, h->stream>>>(A));
    else CN_LAUNCH(h, "tf_scan", tf_scan<31><<<dim3((unsigned)((nt + TF_NT - 1) / TF_NT)), dim3(TF_THREADS), 0, h->stream>>>(A));
    // the windows only need the marks: queued right behind the scan
    CN_HIP(h, hipMemsetAsync(d_twcnt, 0, 8, h->stream));
    TwArgs W{d_bitmap, a->d_tw_boff, a->d_len, a->d_tw_tiles, thr_adj, d_wout, d_twcnt, (uint32_t)std::min<size_t>(win_ws_cap, 0x7fffffff)};
    CN_LAUNCH(h, "tw_scan", tw_scan<<<dim3((unsigned)a->tw_n_tiles), dim3(256), 0, h->stream>>>(W));
    CN_TRY(cnscan::exclusive_u32_multi(h, "tf_order", reinterpret_cast<const uint32_t *>(d_tc), (int64_t)nt, 4, 4, d_offq, d_part, d_cnt));
    const uint4 caps = make_uint4((uint32_t)seg[0], (uint32_t)seg[1], (uint32_t)seg[2], (uint32_t)seg[3]);
    CN_LAUNCH(h, "tf_gather", tf_gather<<<dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, h->stream>>>(d_rows[0], d_rows[1], d_rows[2], d_rows[3], d_tc, d_offq[0], d_offq[1],
                                                                                                    d_offq[2], d_offq[3], d_list[0], d_list[1], d_list[2], d_list[3], caps,
                                                                                                    (int64_t)nt));
    {
        hipEvent_t ea = cn_event(h), eb = cn_event(h);
        (void)hipEventRecord(ea, h->stream);
        tf_ctgoff<<<dim3((unsigned)((a->n + 256) / 256)), dim3(256), 0, h->stream>>>(a->d_tf_ct0, a->n, (int64_t)nt, d_offq[0], d_offq[1], d_offq[2], d_offq[3], d_cnt, d_coff,
                                                                                   d_err);
        tf_pair_dev<<<dim3((unsigned)((tot_cap + 255) / 256)), dim3(256), 0, h->stream>>>(d_list[0], d_list[1], d_list[2], d_list[3], d_coff, a->n, d_cnt, caps, k, d_hits);
        (void)hipEventRecord(eb, h->stream);
        h->recs.push_back(cornetto_accel::Rec{"tf_pair", ea, eb});
        CN_HIP(h, hipGetLastError());
    }
    CN_HIP(h, hipMemcpyAsync(p_cnt, d_cnt, 64, hipMemcpyDeviceToHost, h->stream));
    CN_HIP(h, hipMemcpyAsync(p_cnt + 8, d_twcnt, 8, hipMemcpyDeviceToHost, h->stream));
    CN_HIP(h, hipMemcpyAsync(p_wins, d_wout, win_cap * sizeof(int4), hipMemcpyDeviceToHost, h->stream));
    CN_HIP(h, cn_result_d2h(h, out, d_hits, tot_cap * sizeof(cornetto_hit_t)));
    S->hits = out;
    out = nullptr;
    S->hit_cap = tot_cap;
    S->p_wins = p_wins;
    S->win_cap = win_cap;
    S->queued = true;
    return CORNETTO_OK;
}

int cn_telo_spec_finish(cornetto_accel_t *h, cornetto_asm_t *a, const char *motif, double thr_adj, CnTeloSpec *S, cornetto_hit_t **hits, int64_t *n_hits,
                        cornetto_win_t **wins, int64_t *n_wins)
{
    if (!S->queued) return 1;
    S->queued = false;
    const unsigned long long *c = S->p_cnt;
    const uint32_t ovf = (uint32_t)(c[4] & 0xFFFFFFFFull), err = (uint32_t)(c[4] >> 32);
    const unsigned long long n_win = c[8];
    bool ok = ovf <= TF_ROW && err == 0 && c[0] == c[1] && c[2] == c[3] && n_win <= S->win_cap;
    for (int q = 0; q < 4; ++q) ok = ok && c[q] <= S->seg_cap[q];
    cornetto_win_t *w = ok ? (cornetto_win_t *)malloc((n_win ? (size_t)n_win : 1) * sizeof(cornetto_win_t)) : nullptr;
    if (!w) {                                         // an estimate did not hold (or anything the exact call has its own answer for): nothing of this attempt is returned
        cn_result_quiesce(h);
        cornetto_free(S->hits);
        S->hits = nullptr;
        a->tf_est_key.clear();
        return 1;
    }
    std::vector<int4> host(S->p_wins, S->p_wins + n_win);
    std::sort(host.begin(), host.end(), [](const int4 &x, const int4 &y) { return x.x != y.x ? x.x < y.x : x.y < y.y; });
    for (size_t i = 0; i < host.size(); ++i) {
        w[i].ctg = host[i].x; w[i].start = host[i].y; w[i].end = host[i].z; w[i].car = host[i].w;
    }
    *wins = w;
    *n_wins = (int64_t)n_win;
    *hits = S->hits;
    *n_hits = (int64_t)(c[0] + c[2]);
    S->hits = nullptr;
    for (int q = 0; q < 4; ++q) a->tf_est_cnt[q] = (int64_t)c[q];
    a->tf_est_wins = (int64_t)n_win;
    (void)motif; (void)thr_adj;
    return CORNETTO_OK;
}

// sdust.hip — symmetric DUST (lh3) low-complexity masking for gfx950; replaces sdust_core() and helpers,
// src/sdust/sdust.c:66-160 of the reference, bit for bit (including the stale-window quirk after N runs).
//
// The reference is one sequential recurrence per contig.  Here every LANE runs that recurrence over its
// own chunk of a contig (thread-per-chunk speculation, exact by construction):
//
//  * State.  Everything except the perfect-interval list P is a pure function of the last W-2 emitted
//    3-mers (window w, counts cw/cv, scores rw/rv, suffix length L) plus min(l, W).  A lane therefore
//    starts 2W bases + (W-2) word emissions before its chunk with an empty state; once W-2 words have been
//    pushed the window state equals the true one, and every P entry created before that point has been
//    evicted (its start is < window start) at least 2W-4 bases later — before the chunk begins.  From the
//    chunk start on, the lane's state IS the sequential state.  The warm-up start is found by scanning
//    backwards for W-2 word emissions (not bases), so N-dense sequence is handled exactly.
//  * Output.  The reference's result list is the canonical union (overlapping or touching intervals
//    merged, :94-98) of the intervals it saves, in increasing start order.  Each lane records only the
//    intervals saved at times inside its chunk, merges them locally, and the chunk lists are stitched in
//    chunk order with the same rule.
//  * P without a list.  P is sorted by descending start; all that is ever read from it is, per start
//    value, the newest entry (largest finish, and — because an insert requires a ratio >= every entry with
//    start >= its own — also the best ratio), the minimum start, and emptiness.  finish = start + l + 3
//    for every entry (:123 with :111).  So P is a ring of one 32-bit slot (r,l) per start value, and
//    find_perfect() (:104-128) becomes one backward pass with a running maximum: O(W) instead of O(W*|P|).
//
// Per-lane state lives in LDS, laid out lane-minor ([index][lane], 4-byte columns) so that any per-lane
// index pattern is bank-conflict free: ring of 3-mers (bytes), cw/cv (bytes), P slots (dwords).
// One wavefront per workgroup; no barriers.  Integer/LDS-latency bound by nature, not HBM bound.
#include <algorithm>

#include "common.hpp"
#include "scan.hpp"

namespace {

__device__ __forceinline__ int nt4_code(uint32_t c);

struct SdChunk {
    int32_t ctg, start, end;
};

struct SdArgs {
    const uint8_t *bases;
    const int64_t *ctg_off;
    const int32_t *ctg_len;
    const SdChunk *chunks;
    int32_t n_chunks;
    int32_t T, W;
    uint2 *out;        // [n_chunks][cap] (start, finish)
    uint32_t *out_n;   // [n_chunks] number of intervals the chunk produced (may exceed cap: overflow)
    uint32_t cap;
    unsigned long long *stats;   // optional [4]: wave steps, cooperative find_perfect calls, cooperative trims, save/evicts
    uint32_t *ovf;               // max over chunks of (intervals produced) when that exceeds cap, else untouched
    uint32_t *slots;             // sdust_w64: [n_chunks][64] P slots (start & 63 -> r | l << 16), global memory
    int32_t map_stride;          // chunk -> lane mapping (0: 64-wave groups, 1: strided over the grid)
    // bounded warm-up search (sdust_w64): the local backward scan gives up after SD_SCAN_CAP bases; then
    const uint32_t *wtab;        //   exclusive prefix of per-256-base-block word-emission counts, or NULL
    const int64_t *wtab_base;    //   first table entry of each contig
    uint32_t *need_wtab;         //   set when a lane gave up and no table was supplied (host builds it and reruns)
};

constexpr int SD_SCAN_CAP = 1024;   // bases a lane scans backwards by itself before using the table
constexpr int SD_WBLK = 256;        // bases per table block

// word-emission counts per 256-base block of every contig: a word "ends" at q when bases q-2..q are A/C/G/T
// (this is what the recurrence pushes, src/sdust/sdust.c:144-145, independent of its state)
__global__ void sd_wordcount(const uint8_t *bases, const int64_t *ctg_off, const int32_t *ctg_len, const int64_t *wtab_base,
                             int32_t n_ctg, int64_t n_blocks, uint32_t *cnt)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_blocks) return;
    int lo = 0, hi = n_ctg - 1;                 // contig of this block: largest c with wtab_base[c] <= g
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (wtab_base[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const int c = lo;
    const int len = ctg_len[c];
    const uint8_t *seq = bases + ctg_off[c];
    const int q0 = (int)(g - wtab_base[c]) * SD_WBLK;
    int run = 0, n = 0;
    for (int q = q0 - 2 < 0 ? 0 : q0 - 2; q < q0 + SD_WBLK && q < len; ++q) {
        run = nt4_code(seq[q]) < 4 ? run + 1 : 0;
        n += (q >= q0) & (run >= 3);
    }
    cnt[g] = (uint32_t)n;
}


template <int RC>  // ring / slot capacity, power of two >= W - 2
struct SdLds {
    uint8_t lut[256];
    uint8_t ring[RC / 4][64][4];
    uint8_t cw[16][64][4];
    uint8_t cv[16][64][4];
    uint32_t slot[RC][64];
};

template <int RC>
__global__ __launch_bounds__(64) void sdust_kernel(SdArgs A)
{
    __shared__ SdLds<RC> S;
    const int lane = threadIdx.x;
    constexpr int MASK = RC - 1;

    // seq_nt4_table (src/sdust/sdust.c:23-40): A/a C/c G/g T/t -> 0..3, bytes 0..3 -> themselves, else 4
    for (int c = lane; c < 256; c += 64) {
        uint8_t v = 4;
        if (c < 4) v = (uint8_t)c;
        else if (c == 'A' || c == 'a') v = 0;
        else if (c == 'C' || c == 'c') v = 1;
        else if (c == 'G' || c == 'g') v = 2;
        else if (c == 'T' || c == 't') v = 3;
        S.lut[c] = v;
    }
    for (int i = 0; i < 16; ++i) {
        *reinterpret_cast<uint32_t *>(S.cw[i][lane]) = 0;
        *reinterpret_cast<uint32_t *>(S.cv[i][lane]) = 0;
    }
    for (int i = 0; i < RC; ++i) S.slot[i][lane] = 0;
    __syncthreads();

    const int cid = blockIdx.x * 64 + lane;
    if (cid >= A.n_chunks) return;
    const SdChunk ch = A.chunks[cid];
    const int len = A.ctg_len[ch.ctg];
    const uint8_t *seq = A.bases + A.ctg_off[ch.ctg];
    const int T = A.T, W = A.W, CAPW = W - 2;

#define RING(i) S.ring[((i) & MASK) >> 2][lane][(i) & 3]
#define CW(t) S.cw[(t) >> 2][lane][(t) & 3]
#define CV(t) S.cv[(t) >> 2][lane][(t) & 3]
#define SLOT(s) S.slot[(s) & MASK][lane]

    // ---- where to start: W-2 word emissions before (chunk start - 2W) ------------------------------
    int u = 0;
    if (ch.start > 0) {
        int y = ch.start - 2 * W;
        if (y > 2) {
            int need = CAPW, run = 0, p = y - 1;
            // walking down, `run` = number of consecutive ACGT bases at [p, p+run); a word ends at q
            // (bases q-2..q) for every q with a run of >= 3 ending there
            for (; p >= 0; --p) {
                if (S.lut[seq[p]] < 4) {
                    if (++run >= 3 && --need == 0) break;
                } else {
                    run = 0;
                }
            }
            u = p > 0 ? p : 0;
        }
    }

    // ---- sequential state ------------------------------------------------------------------------
    int l = 0, front = 0, size = 0, L = 0, rw = 0, rv = 0;
    unsigned t = 0;
    int nP = 0, minstart = 0;
    bool have_last = false;
    uint32_t last_s = 0, last_f = 0, n_out = 0;
    uint2 *out = A.out + (size_t)cid * A.cap;
    const int rec_from = ch.start;

    auto emit = [&](int ps, int pf) {      // :93-99 on the lane-local list
        if (have_last && ps <= (int)last_f) {
            if (pf > (int)last_f) last_f = (uint32_t)pf;
        } else {
            if (have_last) {
                if (n_out < A.cap) out[n_out] = make_uint2(last_s, last_f);
                ++n_out;
            }
            have_last = true;
            last_s = (uint32_t)ps;
            last_f = (uint32_t)pf;
        }
    };
    // save_masked_regions(start) (:88-102) when it is not a no-op: nP > 0 && minstart < start
    auto save_evict = [&](int start, int now) {
        const uint32_t sl = SLOT(minstart);
        if (now >= rec_from) emit(minstart, minstart + (int)(sl >> 16) + 3);
        int q = minstart;
        const int qend = start - minstart > RC ? minstart + RC : start;
        for (; q < qend && nP > 0; ++q)
            if (SLOT(q)) {
                SLOT(q) = 0;
                --nP;
            }
        if (nP > 0) {
            q = start;
            for (int g = 0; g < RC && SLOT(q) == 0; ++g) ++q;   // bounded: a live entry lies within RC of start
            minstart = q;
        }
    };

    const int stop = (ch.end == len) ? len + 1 : ch.end;   // the last chunk also runs the sentinel step i == len
    uint32_t word = 0;
    if (u < len) word = *reinterpret_cast<const uint32_t *>(seq + (u & ~3)) >> (8 * (u & 3));
    for (int i = u; i < stop; ++i) {
        if ((i & 3) == 0 && i != u && i < len) word = *reinterpret_cast<const uint32_t *>(seq + i);
        const int b = i < len ? S.lut[word & 0xFFu] : 4;
        word >>= 8;
        if (b < 4) {
            ++l;
            t = (t << 2 | (unsigned)b) & 63u;                           // :144
            if (l >= 3) {
                const int start = (l - W > 0 ? l - W : 0) + (i + 1 - l);   // :146
                if (nP > 0 && minstart < start) save_evict(start, i);  // :147
                // shift_window (:66-86)
                if (size >= CAPW) {
                    const int s = RING(front);
                    front = (front + 1) & MASK;
                    --size;
                    const int c = CW(s) - 1;
                    CW(s) = (uint8_t)c;
                    rw -= c;
                    if (L > size) {
                        --L;
                        const int d = CV(s) - 1;
                        CV(s) = (uint8_t)d;
                        rv -= d;
                    }
                }
                RING(front + size) = (uint8_t)t;
                ++size;
                ++L;
                {
                    const int c = CW(t);
                    CW(t) = (uint8_t)(c + 1);
                    rw += c;
                    const int d = CV(t);
                    CV(t) = (uint8_t)(d + 1);
                    rv += d;
                    if ((d + 1) * 10 > T << 1) {                        // :79
                        int s;
                        do {
                            s = RING(front + size - L);
                            const int e = CV(s) - 1;
                            CV(s) = (uint8_t)e;
                            rv -= e;
                            --L;
                        } while (s != (int)t && L > 0);
                    }
                }
                if (rw * 10 > L * T) {                                  // :149 -> find_perfect (:104-128)
                    int r = rv, max_r = 0, max_l = 0, fold = size;
                    const int i0 = size - L - 1;
                    for (int k = i0; k >= 0; --k) {
                        const int tt = RING(front + k);
                        const int c = CV(tt);
                        CV(tt) = (uint8_t)(c + 1);
                        r += c;
                        const int new_l = size - k - 1;
                        if (r * 10 > T * new_l) {                       // :112
                            while (fold > k) {                          // :113-117 as a running maximum
                                --fold;
                                const uint32_t sl = SLOT(start + fold);
                                if (sl) {
                                    const int pr = (int)(sl & 0xFFFFu), pl = (int)(sl >> 16);
                                    if (max_r == 0 || pr * max_l > max_r * pl) {
                                        max_r = pr;
                                        max_l = pl;
                                    }
                                }
                            }
                            if (max_r == 0 || r * max_l >= max_r * new_l) {   // :118
                                max_r = r;
                                max_l = new_l;
                                const int ps = start + k;
                                if (SLOT(ps) == 0) {
                                    if (nP == 0 || ps < minstart) minstart = ps;
                                    ++nP;
                                }
                                SLOT(ps) = (uint32_t)r | ((uint32_t)new_l << 16);
                            }
                        }
                    }
                    for (int k = 0; k <= i0; ++k) {                     // undo the in-place use of cv as c[]
                        const int tt = RING(front + k);
                        CV(tt) = (uint8_t)(CV(tt) - 1);
                    }
                }
            }
        } else {
            int start = (l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l);  // :152
            while (nP > 0) {                                            // :153
                if (minstart >= start) start = minstart + 1;
                save_evict(start, i);
                ++start;
            }
            l = 0;
            t = 0;                                                      // :154 — window and counters kept
        }
    }
    if (have_last) {
        if (n_out < A.cap) out[n_out] = make_uint2(last_s, last_f);
        ++n_out;
    }
    A.out_n[cid] = n_out;
    if (n_out > A.cap) atomicMax(A.ovf, n_out);
#undef RING
#undef CW
#undef CV
#undef SLOT
}


// ---------------------------------------------------------------------------------------------------
// sdust_w64: the production kernel for W - 2 <= 64 (default W = 64).  Same recurrence, but every
// data-dependent LOOP of the reference is replaced by wave-cooperative, loop-free code, because in a
// 64-lane wave "rare per lane" is "every step per wave":
//   * cv / rv are not maintained at all.  L (the longest suffix of the window in which no 3-mer occurs
//     more than m = 2T/10 times, :79-85) is kept as the absolute index `vs` of the first word of that
//     suffix; pushing word t can only move vs just past the (m+1)-th most recent occurrence of t, which is
//     looked up — only when cw[t] > m — by ONE ballot over the owner's ring (lane j reads ring slot j).
//   * find_perfect (:104-128): lane j takes window position j of the owning lane; suffix scores r_j
//     come from 6 ballots (equal-word mask), a popcount and a suffix-sum scan; the running maximum over
//     P entries / earlier candidates is a suffix-max scan with exact cross-multiplied ratio compares.
//   * P occupancy is a 64-bit mask per lane (bit = start & 63), so save_masked_regions (:88-102) and the
//     N flush (:153) are rotates / ctz instead of list walks.
//     Per-lane LDS state is just the ring and the 64 byte counters (8 KB per wave): 20 waves per CU.
// Chunks are dealt to lanes strided over the whole grid, so that a long low-complexity array (telomere,
// satellite) is spread over many waves instead of serialising inside one.
// ---------------------------------------------------------------------------------------------------
struct SdLds64 {
    uint8_t lut[256];          // seq_nt4_table
    uint8_t ring[16][64][4];   // [slot >> 2][lane][slot & 3], slot = absolute word index & 63
    uint8_t cw[16][64][4];     // [3-mer >> 2][lane][3-mer & 3] = copies of the 3-mer in the window
};

// seq_nt4_table (src/sdust/sdust.c:23-40) without a table: A/a C/c G/g T/t -> 0..3, bytes 0..3 -> themselves, else 4
__device__ __forceinline__ int nt4_code(uint32_t c)
{
    const uint32_t cl = c | 0x20u, idx = cl - 0x61u;                 // a=0 c=2 g=6 t=19
    const bool acgt = idx < 20u && ((0x80045u >> idx) & 1u);
    uint32_t code = (cl >> 1) & 3u;                                  // a0 c1 g3 t2
    code ^= code >> 1;                                               // a0 c1 g2 t3
    return acgt ? (int)code : (c < 4u ? (int)c : 4);
}

__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ unsigned long long rdlane64(unsigned long long v, int l)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ unsigned long long rotr64(unsigned long long x, int r)
{
    r &= 63;
    return r ? (x >> r) | (x << (64 - r)) : x;
}
__device__ __forceinline__ unsigned long long rotl64(unsigned long long x, int r) { return rotr64(x, 64 - (r & 63)); }


// ---- wave64 forward inclusive scans on DPP (no LDS traffic): row_shr 1,2,4,8 inside each row of 16 lanes,
// then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3.
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
__device__ __forceinline__ int wave_scan_add(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, DPP_ROW_SHR(1), 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, DPP_ROW_SHR(2), 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, DPP_ROW_SHR(4), 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, DPP_ROW_SHR(8), 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, DPP_ROW_BCAST15, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, DPP_ROW_BCAST31, 0xC, 0xF, false);
    return x;
}
// running maximum of the ratio r/l (r == 0: no entry); exact, products < 2^24
__device__ __forceinline__ void ratio_max_step(int &xr, int &xl, int pr, int pl)
{
    if (pr != 0 && (xr == 0 || __mul24(pr, xl) > __mul24(xr, pl))) { xr = pr; xl = pl; }
}
#define RATIO_SCAN_STEP(ctrl, rmask, bound)                                                   \
    {                                                                                         \
        const int pr = __builtin_amdgcn_update_dpp(0, xr, (ctrl), (rmask), 0xF, (bound));       \
        const int pl = __builtin_amdgcn_update_dpp(0, xl, (ctrl), (rmask), 0xF, (bound));       \
        ratio_max_step(xr, xl, pr, pl);                                                       \
    }
__device__ __forceinline__ void wave_scan_ratio_max(int &xr, int &xl)
{
    RATIO_SCAN_STEP(DPP_ROW_SHR(1), 0xF, true)
    RATIO_SCAN_STEP(DPP_ROW_SHR(2), 0xF, true)
    RATIO_SCAN_STEP(DPP_ROW_SHR(4), 0xF, true)
    RATIO_SCAN_STEP(DPP_ROW_SHR(8), 0xF, true)
    RATIO_SCAN_STEP(DPP_ROW_BCAST15, 0xA, false)
    RATIO_SCAN_STEP(DPP_ROW_BCAST31, 0xC, false)
}

__global__ __launch_bounds__(64) void sdust_w64(SdArgs A)
{
    __shared__ SdLds64 S;
    const int lane = threadIdx.x;
    for (int i = 0; i < 16; ++i) *reinterpret_cast<uint32_t *>(S.cw[i][lane]) = 0;
    for (int c = lane; c < 256; c += 64) S.lut[c] = (uint8_t)nt4_code((uint32_t)c);
    __syncthreads();

    // chunk of this lane: strided over the whole grid (lane l of wave w owns chunk l * waves + w).  A wave's 64
    // lanes then sample 64 far-apart places of the input, so the share of low-complexity lanes in a wave is the
    // global average instead of the local one (a telomere array or a satellite-rich small contig no longer
    // serialises inside one wave: 50 -> 28 ms on the 3.16 Gbp assembly), while the waves resident at any time
    // still advance through 64 compact regions (no measurable memory penalty).  map_stride = 0 keeps the
    // older 64-wave group interleave for comparison.
    int cid = lane * (int)gridDim.x + (int)blockIdx.x;
    if (!A.map_stride) cid = (((int)blockIdx.x >> 6) << 12) + lane * 64 + ((int)blockIdx.x & 63);
    bool active = cid < A.n_chunks;

    const int T = A.T, W = A.W, CAPW = W - 2;
    const int m = (T << 1) / 10;                     // cv[t]*10 > T<<1  <=>  cv[t] > m   (:79)
    SdChunk ch{0, 0, 0};
    int len = 0;
    const uint8_t *seq = A.bases;
    if (active) {
        ch = A.chunks[cid];
        len = A.ctg_len[ch.ctg];
        seq = A.bases + A.ctg_off[ch.ctg];
    }
    // P slots of this lane: one 256-byte row in global memory (L2 resident; touched only around
    // find_perfect / save_masked_regions), which keeps the per-wave LDS at 20 KB = 8 waves per CU
    uint32_t *myslots = A.slots + (size_t)(active ? cid : 0) * 64;
#define RINGL(sl) S.ring[((sl) & 63) >> 2][lane][(sl) & 3]
#define CWL(t) S.cw[(t) >> 2][lane][(t) & 3]

    // ---- warm-up start: W-2 word emissions before (chunk start - 2W) --------------------------------
    int u = 0;
    if (active && ch.start > 0) {
        const int y = ch.start - 2 * W;
        if (y > 2) {
            int need = CAPW, run = 0, p = y - 1;
            const int floor_p = y - SD_SCAN_CAP > 0 ? y - SD_SCAN_CAP : 0;
            for (; p >= floor_p; --p) {
                if (nt4_code(seq[p]) < 4) {
                    if (++run >= 3 && --need == 0) break;
                } else {
                    run = 0;
                }
            }
            if (need == 0 || floor_p == 0) {
                u = p > 0 ? p : 0;
            } else if (A.wtab == nullptr) {
                // N-dense stretch longer than the local scan: ask the host for the word-count table and a rerun
                atomicOr(A.need_wtab, 1u);
                active = false;
            } else {
                // rank of the wanted word among the word emissions of the contig, from the block table
                const uint32_t *tab = A.wtab + A.wtab_base[ch.ctg];
                const uint32_t t0 = tab[0];
                const int yb = y / SD_WBLK;
                int wy = (int)(tab[yb] - t0);                     // emissions ending before block yb
                {
                    int r2 = 0;
                    for (int q = yb * SD_WBLK - 2 < 0 ? 0 : yb * SD_WBLK - 2; q < y; ++q) {
                        r2 = nt4_code(seq[q]) < 4 ? r2 + 1 : 0;
                        wy += (q >= yb * SD_WBLK) & (r2 >= 3);
                    }
                }
                if (wy < CAPW) {
                    u = 0;
                } else {
                    const int rank = wy - CAPW + 1;               // 1-based rank of the oldest word that must be replayed
                    int lo = 0, hi = yb;                          // largest block b with (tab[b] - t0) < rank
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if ((int)(tab[mid] - t0) < rank) lo = mid; else hi = mid - 1;
                    }
                    int seen = (int)(tab[lo] - t0), r2 = 0, q = lo * SD_WBLK - 2 < 0 ? 0 : lo * SD_WBLK - 2;
                    for (; q < y; ++q) {
                        r2 = nt4_code(seq[q]) < 4 ? r2 + 1 : 0;
                        if (q >= lo * SD_WBLK && r2 >= 3 && ++seen == rank) break;
                    }
                    u = q - 2 > 0 ? q - 2 : 0;                     // first base of that word
                }
            }
        }
    }
    u &= ~63;  // starting a little earlier is still exact, and keeps every lane on the same 64-byte phase

    // ---- per-lane sequential state --------------------------------------------------------------------
    int l = 0, size = 0, rw = 0;
    int p = -1;                 // absolute index of the newest word in the window
    int vs = 0;                 // absolute index of the first word of v (the suffix with all counts <= m)
    unsigned t = 0, s_pref = 0;
    unsigned long long occ = 0; // occupied P slots, bit = start & 63
    int minstart = 0;
    bool have_last = false;
    uint32_t last_s = 0, last_f = 0, n_out = 0;
    uint2 *out = A.out + (size_t)(active ? cid : 0) * A.cap;
    const int rec_from = ch.start;
    const int stop = (ch.end == len) ? len + 1 : ch.end;

    auto emit = [&](int ps, int pf) {               // :93-99 on the lane-local list
        if (have_last && ps <= (int)last_f) {
            if (pf > (int)last_f) last_f = (uint32_t)pf;
        } else {
            if (have_last) {
                if (n_out < A.cap) out[n_out] = make_uint2(last_s, last_f);
                ++n_out;
            }
            have_last = true;
            last_s = (uint32_t)ps;
            last_f = (uint32_t)pf;
        }
    };
    // save_masked_regions(start) when it does something: occ != 0 && minstart < start   (:88-102)
    auto save_evict = [&](int start, int now) {
        const uint32_t sl = __hip_atomic_load(&myslots[minstart & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (now >= rec_from) emit(minstart, minstart + (int)(sl >> 16) + 3);
        const int gone = start - minstart;           // starts minstart .. start-1 leave the window
        if (gone >= 64) {
            occ = 0;
        } else {
            unsigned long long r = rotr64(occ, minstart & 63);   // bit 0 <-> minstart
            r &= ~0ull << gone;
            occ = rotl64(r, minstart & 63);
        }
        if (occ) minstart = start + __builtin_ctzll(rotr64(occ, start & 63));
    };

    // Bases are consumed one 64-byte block per lane per 64 steps (4 x dwordx4, fetched a whole block ahead), so
    // that every 64-byte sector is requested from L2 / the fabric once: with one dword per 4 steps the ~40 k
    // concurrent per-lane streams of an XCD overflowed its 4 MB L2 and each line was re-fetched ~6 times
    // (rocprofv3 FETCH_SIZE, profiles/).  A block never leaves its contig: contigs start 64-byte aligned.
    uint4 nb0 = make_uint4(0, 0, 0, 0), nb1 = nb0, nb2 = nb0, nb3 = nb0;
    if (active && u < len) {
        const uint4 *q = reinterpret_cast<const uint4 *>(seq + u);
        nb0 = q[0]; nb1 = q[1]; nb2 = q[2]; nb3 = q[3];
    }
    active = active && u < stop;
    const bool small_t = T <= 100000;   // L*T < 2^24: 24-bit multiplies are exact

    unsigned st_steps = 0, st_fp = 0, st_trim = 0;
    // Everything in the loop body is predicated arithmetic except three regions: the two rare P-maintenance
    // paths (entered on a wave-uniform test) and the word step itself.  Lanes that are done (or never had a
    // chunk) see b = 4 with an empty P and change nothing.
    for (int k64 = 0; __any(active); k64 += 64) {
      // current block -> 16 dwords that are rotated down by one per group of 4 steps; next block in flight
      uint32_t c0 = nb0.x, c1 = nb0.y, c2 = nb0.z, c3 = nb0.w, c4 = nb1.x, c5 = nb1.y, c6 = nb1.z, c7 = nb1.w;
      uint32_t c8 = nb2.x, c9 = nb2.y, c10 = nb2.z, c11 = nb2.w, c12 = nb3.x, c13 = nb3.y, c14 = nb3.z, c15 = nb3.w;
      if (active && u + k64 + 64 < len) {
          const uint4 *q = reinterpret_cast<const uint4 *>(seq + u + k64 + 64);
          nb0 = q[0]; nb1 = q[1]; nb2 = q[2]; nb3 = q[3];
      }
      for (int k4 = k64; k4 < k64 + 64; k4 += 4) {
        const uint32_t word = c0;
        c0 = c1; c1 = c2; c2 = c3; c3 = c4; c4 = c5; c5 = c6; c6 = c7; c7 = c8;
        c8 = c9; c9 = c10; c10 = c11; c11 = c12; c12 = c13; c13 = c14; c14 = c15;
        // seq_nt4_table (:23-40) for the 4 bytes of the group at once: four independent LDS reads
        const uint32_t codes4 = (uint32_t)S.lut[word & 0xFFu] | ((uint32_t)S.lut[(word >> 8) & 0xFFu] << 8) |
                                ((uint32_t)S.lut[(word >> 16) & 0xFFu] << 16) | ((uint32_t)S.lut[word >> 24] << 24);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        ++st_steps;
        const int i = u + k4 + kk;                   // position of this lane; all lanes share i & 63
        const int b = (active & (i < len)) ? (int)((codes4 >> (8 * kk)) & 7u) : 4;
        const bool isbase = b < 4;
        const int l_old = l;
        l = isbase ? l + 1 : 0;
        t = isbase ? ((t << 2 | (unsigned)b) & 63u) : 0u;                     // :144 / :154
        const bool isword = isbase & (l >= 3);
        bool need_trim = false, need_fp = false;

        // ---- P maintenance, rare (one wave-uniform test): the flush at an N or at the end of the sequence
        // (:152-153) and save_masked_regions (:147); both only matter while P is non-empty
        if (__any(active & (occ != 0) & (!isbase | isword))) {
            if (active && !isbase) {
                int st = (l_old - W + 1 > 0 ? l_old - W + 1 : 0) + (i + 1 - l_old);
                while (occ) {
                    if (minstart >= st) st = minstart + 1;
                    save_evict(st, i);
                    ++st;
                }
            }
            if (isword && occ != 0) {
                const int start = (l - W > 0 ? l - W : 0) + (i + 1 - l);     // :146
                if (minstart < start) save_evict(start, i);
            }
        }
        if (isword) {
            // shift_window (:66-86) without cv / rv, straight-line: both table entries are read at once
            // (the oldest word was prefetched at the end of the previous word step)
            const int pop = size >= CAPW ? 1 : 0;
            const unsigned s = s_pref;
            int cs = CWL(s);
            int ct = CWL(t);
            cs -= pop;                               // --cw[s]   (:71)
            CWL(s) = (uint8_t)cs;
            ct = s == t ? cs : ct;
            rw -= cs * pop;
            size += 1 - pop;
            ++p;
            RINGL(p) = (uint8_t)t;                   // :75
            rw += ct;                                // rw += cw[t]++   (:77)
            CWL(t) = (uint8_t)(ct + 1);
            // v must not hold more than m copies of t.  Only when the window now holds more than m can v,
            // a suffix of it, do so: those lanes get their v start moved by the cooperative pass below.
            need_trim = ct + 1 > m;
            if (m == 0) vs = p + 1;
            s_pref = RINGL(p - size + 1);            // the word the next pop removes
        }
        // ---- cooperative trim: vs moves just past the (m+1)-th most recent occurrence of t inside v --------
        // (one ballot over the owner's ring: lane j reads ring slot j; about one lane per wave-step needs it
        // in non-repetitive sequence)
        unsigned long long todo = 0;
        if (m > 0) {
            todo = __ballot(need_trim);
            st_trim += (unsigned)__popcll(todo);
            while (todo) {
                const int o = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                todo &= todo - 1;
                const int o_p = rdlane(p, o), o_size = rdlane(size, o), o_vs = rdlane(vs, o);
                const unsigned o_t = (unsigned)rdlane((int)t, o);
                const unsigned mine = S.ring[lane >> 2][o][lane & 3];             // ring slot `lane` of the owner
                const unsigned long long eq = __ballot(mine == o_t);
                // chronological order: bit k <-> absolute word index o_p - 63 + k
                const unsigned long long chron = rotr64(eq, (o_p + 1) & 63);
                const int ws = o_p - o_size + 1;
                const int first = o_vs > ws ? o_vs : ws;                           // first word of v before the trim
                const int Lc = o_p - first + 1;                                    // 1..64
                const unsigned long long inv = chron & (Lc >= 64 ? ~0ull : ~0ull << (64 - Lc));
                if (__popcll(inv) > m) {
                    const int oldest = __builtin_ctzll(inv);                       // oldest occurrence of t inside v
                    vs = lane == o ? o_p - 63 + oldest + 1 : vs;
                }
            }
        }
        {
            const int ws = p - size + 1;
            const int first = vs > ws ? vs : ws;
            const int L = p - first + 1;
            need_fp = isword & (small_t ? (rw * 10 > __mul24(L, T)) : (rw * 10 > L * T));     // :149
        }
        // ---- cooperative find_perfect (:104-128) ------------------------------------------------------
        // lane <-> window position j = 63 - lane, so that "suffix of the window" = "prefix of the wave" and
        // both scans are forward DPP scans (no LDS round trips).
        todo = __ballot(need_fp);
        st_fp += (unsigned)__popcll(todo);
        while (todo) {
            const int o = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
            todo &= todo - 1;
            const int o_p = rdlane(p, o), o_size = rdlane(size, o), o_vs = rdlane(vs, o);
            const int ws = o_p - o_size + 1;
            const int first = o_vs > ws ? o_vs : ws;
            const int i0 = first - ws - 1;                                     // = size - L - 1
            const int j = 63 - lane;                                           // window position (0 = oldest)
            const bool inwin = j < o_size;
            const int rslot = (ws + j) & 63;
            const unsigned wj = inwin ? S.ring[rslot >> 2][o][rslot & 3] : 0u;
            unsigned long long eq = __ballot(inwin);
#pragma unroll
            for (int bb = 0; bb < 6; ++bb) {
                const bool bit = (wj >> bb) & 1u;
                const unsigned long long bal = __ballot(bit);
                eq &= bit ? bal : ~bal;
            }
            // equal words at later window positions = lower lanes; suffix score r_j = inclusive prefix sum
            const int r = wave_scan_add(inwin ? __popcll(eq & ((1ull << lane) - 1ull)) : 0);
            const int new_l = o_size - j - 1;                                  // :111
            const bool cand = inwin && j <= i0 && r * 10 > __mul24(T, new_l);  // :112 (new_l < 64, T < 2^21)
            const unsigned long long candmask = __ballot(cand);
            if (candmask == 0) continue;                                       // nothing can be inserted
            const int o_l = rdlane(l, o), o_i = rdlane(i, o);
            const int o_start = (o_l - W > 0 ? o_l - W : 0) + (o_i + 1 - o_l);          // :146
            const unsigned long long o_occ = rdlane64(occ, o);
            const int o_cid = rdlane(cid, o);
            uint32_t *orow = A.slots + (size_t)o_cid * 64;
            const int sidx = (o_start + j) & 63;
            const bool has_e = inwin && ((o_occ >> sidx) & 1ull);
            const uint32_t e = has_e ? __hip_atomic_load(&orow[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const int er = (int)(e & 0xFFFFu), el = (int)(e >> 16);
            // X_j = better of (existing entry with this start, candidate j); inclusive maximum over positions >= j
            int xr = er, xl = el;
            if (cand && (er == 0 || __mul24(r, el) >= __mul24(er, new_l))) { xr = r; xl = new_l; }
            wave_scan_ratio_max(xr, xl);
            // maximum over positions > j = the scan value one lane down (wave_shr:1; lane 0 gets 0)
            const int sr = __builtin_amdgcn_update_dpp(0, xr, 0x138, 0xF, 0xF, true);
            const int sl2 = __builtin_amdgcn_update_dpp(0, xl, 0x138, 0xF, 0xF, true);
            int mr = sr, ml = sl2;                                             // :113-117: entries with start >= i + start
            if (er != 0 && (sr == 0 || __mul24(er, sl2) > __mul24(sr, el))) { mr = er; ml = el; }
            const bool ins = cand && (mr == 0 || __mul24(r, ml) >= __mul24(mr, new_l));   // :118
            if (ins) orow[sidx] = (uint32_t)r | ((uint32_t)new_l << 16);       // start = i + start, finish = start + l + 3
            const unsigned long long insj = __brevll(__ballot(ins));           // bit j <-> window position j
            if (insj && lane == o) {
                const int lowest = o_start + __builtin_ctzll(insj);
                if (occ == 0 || lowest < minstart) minstart = lowest;
                occ |= rotl64(insj, o_start & 63);
            }
        }
        active = active & (i + 1 < stop);
      }
      }
    }
    if (cid < A.n_chunks) {                          // every lane that owned a chunk publishes its list
        if (have_last) {
            if (n_out < A.cap) out[n_out] = make_uint2(last_s, last_f);
            ++n_out;
        }
        A.out_n[cid] = n_out;
        if (n_out > A.cap) atomicMax(A.ovf, n_out);
    }
    if (A.stats && lane == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)st_steps);
        atomicAdd(&A.stats[1], (unsigned long long)st_fp);
        atomicAdd(&A.stats[2], (unsigned long long)st_trim);
    }
#undef RINGL
#undef CWL
}

// chunk rows (fixed capacity) -> one dense list in chunk order, tagged with the contig: one wavefront per chunk
__global__ void sdust_gather(const uint2 *in, const uint32_t *cnt, const uint32_t *dst_off, uint32_t cap, const SdChunk *chunks,
                             int32_t n_chunks, cornetto_ivl_t *dst)
{
    const int cid = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (cid >= n_chunks) return;
    const uint32_t n = cnt[cid];
    const int32_t ctg = chunks[cid].ctg;
    const uint2 *src = in + (size_t)cid * cap;
    cornetto_ivl_t *d = dst + dst_off[cid];
    for (uint32_t i = threadIdx.x & 63; i < n; i += 64) d[i] = cornetto_ivl_t{ctg, (int32_t)src[i].x, (int32_t)src[i].y};
}

int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

}  // namespace

extern "C" {

int cornetto_sdust_asm(cornetto_accel_t *h, const cornetto_asm_t *a_in, int32_t T, int32_t W, cornetto_ivl_t **ivls,
                       int64_t *n_ivls)
{
    if (!h || !a_in || !ivls || !n_ivls) return cn_fail(h, CORNETTO_E_ARG, "sdust: bad argument");
    cornetto_asm_t *a = const_cast<cornetto_asm_t *>(a_in);   // only the cached chunk table is touched
    *ivls = nullptr;
    *n_ivls = 0;
    if (W < 3 || W > 258) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: -w %d outside 3..258 (the reference crashes below 3)", W);
    if (T < 0 || T > (1 << 20)) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: -t %d outside 0..2^20", T);
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);

    // chunk = bases per lane.  Small enough that a long low-complexity array is shared by many waves, large
    // enough that the ~3W-base speculative warm-up stays a few percent.  CORNETTO_SDUST_CHUNK overrides
    // (tests use tiny chunks to stress the speculative start).
    int64_t chunk = env_int("CORNETTO_SDUST_CHUNK", 0);
    if (chunk <= 0) chunk = 1536;
    chunk = std::max<int64_t>(16, chunk);
    if (a->sd_chunk != chunk) {
        std::vector<SdChunk> chunks;
        for (int32_t c = 0; c < a->n; ++c)
            for (int64_t s = 0; s < a->len[c]; s += chunk)
                chunks.push_back(SdChunk{c, (int32_t)s, (int32_t)std::min<int64_t>(a->len[c], s + chunk)});
        if (chunks.size() > 0x7fffffffull) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: too many chunks");
        if (a->d_sd_chunks) { (void)hipFree(a->d_sd_chunks); a->d_sd_chunks = nullptr; }
        if (!chunks.empty()) {
            if (hipMalloc(&a->d_sd_chunks, chunks.size() * sizeof(SdChunk)) != hipSuccess)
                return cn_fail(h, CORNETTO_E_NOMEM, "sdust: device allocation failed");
            CN_HIP(h, hipMemcpyAsync(a->d_sd_chunks, chunks.data(), chunks.size() * sizeof(SdChunk), hipMemcpyHostToDevice, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));
        }
        a->sd_chunk = chunk;
        a->sd_n_chunks = (int64_t)chunks.size();
    }
    const size_t nc = (size_t)a->sd_n_chunks;
    cornetto_ivl_t *o = nullptr;
    int64_t n_out = 0;
    if (nc > 0) {
        const SdChunk *d_chunks = reinterpret_cast<const SdChunk *>(a->d_sd_chunks);
        // per chunk: count (4 B) + ordered offset (4 B) + scan partials; then {total u64, ovf u32}
        uint32_t *d_cnt = (uint32_t *)cn_ws(h, WS_SD_CNT, nc * 8 + ((nc + 4095) / 4096 + 1) * 4);
        unsigned long long *d_tot = (unsigned long long *)cn_ws(h, WS_SD_STATS, 64);
        unsigned long long *p_tot = (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
        if (!d_cnt || !d_tot || !p_tot) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation failed");
        uint32_t *d_off = d_cnt + nc, *d_part = d_off + nc;
        const bool want_stats = env_int("CORNETTO_SDUST_STATS", 0) != 0;
        size_t cap = (size_t)std::max<int64_t>(16, chunk / 32);
        cap = (size_t)env_int("CORNETTO_SDUST_CAP", (int)cap);
        if (h->dev[WS_SD_OUT].bytes / (nc * sizeof(uint2)) > cap) cap = h->dev[WS_SD_OUT].bytes / (nc * sizeof(uint2));
        unsigned long long tot = 0;
        uint2 *d_out = nullptr;
        for (int attempt = 0; attempt < 4; ++attempt) {
            d_out = (uint2 *)cn_ws(h, WS_SD_OUT, nc * cap * sizeof(uint2));
            if (!d_out) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation of %zu bytes failed", nc * cap * sizeof(uint2));
            CN_HIP(h, hipMemsetAsync(d_tot, 0, 64, h->stream));
            uint32_t *d_slots = (uint32_t *)cn_ws(h, WS_SD_OFF, nc * 64 * sizeof(uint32_t));
            if (!d_slots) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation failed");
            SdArgs A{a->d_bases, a->d_off, a->d_len, d_chunks, (int32_t)nc, T, W, d_out, d_cnt, (uint32_t)cap,
                     want_stats ? d_tot + 2 : nullptr, reinterpret_cast<uint32_t *>(d_tot + 1), d_slots, env_int("CORNETTO_SDUST_MAP", 1),
                     a->d_wtab, a->d_wtab_base, reinterpret_cast<uint32_t *>(d_tot + 1) + 1};
            unsigned nb = (unsigned)((nc + 63) / 64);
            const int variant = env_int("CORNETTO_SDUST_VARIANT", 0);   // 1 = force the per-lane reference-shaped kernel
            if (W - 2 <= 64 && variant == 0) {
                if (env_int("CORNETTO_SDUST_MAP", 1) == 0) nb = (unsigned)((nc + 4095) / 4096) * 64;   // whole groups of 64 waves
                CN_LAUNCH(h, "sdust_kernel", sdust_w64<<<dim3(nb), dim3(64), 0, h->stream>>>(A));
            } else if (W - 2 <= 64) {
                CN_LAUNCH(h, "sdust_kernel", sdust_kernel<64><<<dim3(nb), dim3(64), 0, h->stream>>>(A));
            } else {
                CN_LAUNCH(h, "sdust_kernel", sdust_kernel<256><<<dim3(nb), dim3(64), 0, h->stream>>>(A));
            }
            // ordered position of every chunk's intervals (chunks are in contig order) + grand total
            CN_TRY(cnscan::exclusive_u32(h, "sdust_scan", d_cnt, (int64_t)nc, 1, d_off, d_part, d_tot));
            CN_HIP(h, hipMemcpyAsync(p_tot, d_tot, 64, hipMemcpyDeviceToHost, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));
            if (want_stats)
                fprintf(stderr, "[sdust stats] chunks %zu waves %u wave-steps %llu find_perfect calls %llu trims %llu\n", nc, nb, p_tot[2], p_tot[3], p_tot[4]);
            const uint32_t ovf = (uint32_t)(p_tot[1] & 0xFFFFFFFFull);
            const bool need_wtab = (p_tot[1] >> 32) != 0;
            if (need_wtab) {
                // some lane met more than SD_SCAN_CAP bases without W-2 word emissions (N-dense input): build the
                // word-count table once for this assembly and run again — bounded work whatever the input
                if (a->d_wtab) return cn_fail(h, CORNETTO_E_HIP, "sdust: word-count table requested twice");
                std::vector<int64_t> base(a->n + 1, 0);
                for (int32_t c = 0; c < a->n; ++c) base[c + 1] = base[c] + a->len[c] / SD_WBLK + 2;
                a->n_wblocks = base[a->n];
                const size_t npart = ((size_t)a->n_wblocks + 4095) / 4096 + 1;
                uint32_t *d_wcnt = nullptr;
                if (hipMalloc((void **)&a->d_wtab, ((size_t)a->n_wblocks + 1) * 4) != hipSuccess ||
                    hipMalloc((void **)&a->d_wtab_base, ((size_t)a->n + 1) * 8) != hipSuccess ||
                    hipMalloc((void **)&d_wcnt, ((size_t)a->n_wblocks + npart + 1) * 4) != hipSuccess)
                    return cn_fail(h, CORNETTO_E_NOMEM, "sdust: word-count table allocation failed");
                CN_HIP(h, hipMemcpyAsync(a->d_wtab_base, base.data(), ((size_t)a->n + 1) * 8, hipMemcpyHostToDevice, h->stream));
                const unsigned nbw = (unsigned)((a->n_wblocks + 255) / 256);
                CN_LAUNCH(h, "sd_wordcount", sd_wordcount<<<dim3(nbw), dim3(256), 0, h->stream>>>(a->d_bases, a->d_off, a->d_len, a->d_wtab_base, a->n,
                                                                                         a->n_wblocks, d_wcnt));
                int rc = cnscan::exclusive_u32(h, "sd_wordscan", d_wcnt, a->n_wblocks, 1, a->d_wtab, d_wcnt + a->n_wblocks, nullptr);
                hipError_t e = hipStreamSynchronize(h->stream);   // `base` is a local
                (void)hipFree(d_wcnt);
                if (rc != CORNETTO_OK) return rc;
                if (e != hipSuccess) return cn_fail(h, CORNETTO_E_HIP, "sdust: word-count table build failed");
                continue;
            }
            if (ovf <= cap) {
                tot = p_tot[0];
                break;
            }
            if (attempt >= 2) return cn_fail(h, CORNETTO_E_HIP, "sdust: a chunk produced %u intervals after resizing to %zu", ovf, cap);
            cap = ovf;   // rerun with room for the densest chunk: results are never truncated
        }
        if (tot > 0x7fffffffull) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: %llu intervals", tot);
        o = (cornetto_ivl_t *)cn_result_alloc((tot ? tot : 1) * sizeof(cornetto_ivl_t));
        if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: host allocation failed");
        if (tot > 0) {
            cornetto_ivl_t *d_dst = (cornetto_ivl_t *)cn_ws(h, WS_SD_DST, (size_t)tot * sizeof(cornetto_ivl_t));
            cornetto_ivl_t *p_dst = (cornetto_ivl_t *)cn_pin(h, PIN_A, (size_t)tot * sizeof(cornetto_ivl_t));
            if (!d_dst || !p_dst) { cornetto_free(o); return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation failed"); }
            const unsigned nbg = (unsigned)((nc + 3) / 4);
            hipEvent_t ea = cn_event(h), eb = cn_event(h);
            (void)hipEventRecord(ea, h->stream);
            sdust_gather<<<dim3(nbg), dim3(256), 0, h->stream>>>(d_out, d_cnt, d_off, (uint32_t)cap, d_chunks, (int32_t)nc, d_dst);
            (void)hipEventRecord(eb, h->stream);
            h->recs.push_back(cornetto_accel::Rec{"sdust_gather", ea, eb});
            if (hipGetLastError() != hipSuccess ||
                hipMemcpyAsync(p_dst, d_dst, (size_t)tot * sizeof(cornetto_ivl_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) {
                cornetto_free(o);
                return cn_fail(h, CORNETTO_E_HIP, "sdust: gather / copy back failed");
            }
            // stitch chunk lists in order with the reference's merge rule (src/sdust/sdust.c:94-98)
            for (unsigned long long j = 0; j < tot; ++j) {
                const cornetto_ivl_t v = p_dst[j];
                if (n_out > 0 && o[n_out - 1].ctg == v.ctg && v.start <= o[n_out - 1].finish) {
                    if (v.finish > o[n_out - 1].finish) o[n_out - 1].finish = v.finish;
                } else {
                    o[n_out++] = v;
                }
            }
        }
    }
    cn_timing_end(h);
    if (!o) {
        o = (cornetto_ivl_t *)malloc(sizeof(cornetto_ivl_t));
        if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: host allocation failed");
    }
    *ivls = o;
    *n_ivls = n_out;
    return CORNETTO_OK;
}

// process-wide handle for the sdust() drop-in
static cornetto_accel_t *g_handle = nullptr;

uint64_t *cornetto_sdust(void *km, const uint8_t *seq, int l_seq, int T, int W, int *n)
{
    if (n) *n = -1;
    if (km || !seq || !n) return nullptr;
    if (!g_handle) {
        const char *d = getenv("CORNETTO_DEVICE");
        if (cornetto_accel_open(&g_handle, d ? atoi(d) : 0, nullptr) != CORNETTO_OK) return nullptr;
    }
    int64_t len = l_seq < 0 ? (int64_t)strlen((const char *)seq) : l_seq;   // :139
    cornetto_asm_t *a = nullptr;
    const uint8_t *seqs[1] = {seq};
    if (cornetto_asm_upload(g_handle, seqs, &len, 1, &a) != CORNETTO_OK) return nullptr;
    cornetto_ivl_t *iv = nullptr;
    int64_t ni = 0;
    int rc = cornetto_sdust_asm(g_handle, a, T, W, &iv, &ni);
    cornetto_asm_free(g_handle, a);
    if (rc != CORNETTO_OK) return nullptr;
    uint64_t *r = (uint64_t *)malloc((ni ? ni : 1) * sizeof(uint64_t));
    if (r) {
        for (int64_t i = 0; i < ni; ++i) r[i] = (uint64_t)(uint32_t)iv[i].start << 32 | (uint32_t)iv[i].finish;
        *n = (int)ni;
    }
    free(iv);
    return r;
}

}  // extern "C"

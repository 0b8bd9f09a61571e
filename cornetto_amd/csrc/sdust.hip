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
//    chunk start on, the lane's state IS the sequential state.  The warm-up start is found (sd_prep) by
//    scanning backwards for W-2 word emissions (not bases), so N-dense sequence is handled exactly.
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
// Two kernels: sdust_w64 (W - 2 <= 64, T in 5..100000: the production path, described above it) and the older
// sdust_kernel<RC> for every other parameter pair, which keeps the reference's per-lane loops and all of its state in
// LDS (lane-minor layout, [index][lane] in 4-byte columns: bank-conflict free for any per-lane index).
// One wavefront per workgroup; no barriers.  VALU-issue bound by nature, not HBM bound.
#include <algorithm>
#include <type_traits>
#include <ctime>
#include <sched.h>

#include "common.hpp"
#include "scan.hpp"
#include "tiles.hpp"
#include "ivlmerge.hpp"

namespace {

__device__ __forceinline__ int nt4_code(uint32_t c);
struct SdArgs;
struct SdChunk;
template <bool LEAN = false>
__device__ int sd_find_start(const SdArgs &A, const SdChunk ch, const uint8_t *seq);

struct SdChunk {
    int32_t ctg, start, end;
};
// tiles.hpp: chunk j of contig c
struct SdChunkFill {
    SdChunk *out;
    const int32_t *len;
    int32_t chunk;
    __device__ void operator()(int64_t t, int c, int64_t j) const
    {
        const int64_t s = j * chunk, e = s + chunk;
        out[t] = SdChunk{c, (int32_t)s, (int32_t)(e < len[c] ? e : len[c])};
    }
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
    uint32_t *slots;             // sdust_w64: [waves * 64 lanes][64] P slots (start & 63 -> ratio key | l << 24), global memory
    const uint32_t *perm;        // sdust_w64: queue position -> chunk (low-complexity chunks first), or NULL
    uint32_t *queue;             // sdust_w64: next queue position
    uint32_t *claim;             // sdust_w64: [n_chunks] 0 = free; set by the lane that takes the chunk (from the queue, or by running on into it)
    int32_t q_len;               // sdust_w64: queue positions (perm entries, 0xFFFFFFFF = hole)
    int32_t run_on;              // sdust_w64: lanes run on into the next chunk when it is free (CORNETTO_SDUST_RUNON, default 1)
    int32_t chunk;               // sdust_w64: bases per chunk of the main part (every such chunk of a contig but its last)
    // bounded warm-up search (sdust_w64): the local backward scan gives up after SD_SCAN_CAP bases; then
    const uint32_t *wtab;        //   exclusive prefix of per-256-base-block word-emission counts, or NULL
    const int64_t *wtab_base;    //   first table entry of each contig
    uint32_t *need_wtab;         //   set when a lane gave up and no table was supplied (host builds it and reruns)
    const uint32_t *q_len_dev;   // sdust_w64: when set, the number of queue positions is read from here (the list sd_slowlist made)
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

// The same kernel for windows of 256 to 1024 words (258 <= W <= 1026; the reference takes any -w): the state of a lane — ring,
// 32-bit counters, P slots — lives in global memory, laid out [index][lane of the grid] so that the 64 lanes of a wave touch
// consecutive words.  Correct, not fast: nobody masks with such windows routinely.  Slots hold r (< 2^20) | l << 20.
__global__ __launch_bounds__(64) void sdust_kernel_g(SdArgs A, uint8_t *g_ring, uint32_t *g_cw, uint32_t *g_cv, uint32_t *g_slot, int RC)
{
    __shared__ uint8_t s_lut[256];
    const int lane = threadIdx.x;
    const int MASK = RC - 1;
    const size_t NL = (size_t)gridDim.x * 64, gl = (size_t)blockIdx.x * 64 + lane;

    // seq_nt4_table (src/sdust/sdust.c:23-40): A/a C/c G/g T/t -> 0..3, bytes 0..3 -> themselves, else 4
    for (int c = lane; c < 256; c += 64) {
        uint8_t v = 4;
        if (c < 4) v = (uint8_t)c;
        else if (c == 'A' || c == 'a') v = 0;
        else if (c == 'C' || c == 'c') v = 1;
        else if (c == 'G' || c == 'g') v = 2;
        else if (c == 'T' || c == 't') v = 3;
        s_lut[c] = v;
    }
    for (int i = 0; i < 64; ++i) {
        g_cw[(size_t)i * NL + gl] = 0;
        g_cv[(size_t)i * NL + gl] = 0;
    }
    for (int i = 0; i < RC; ++i) g_slot[(size_t)i * NL + gl] = 0;
    __syncthreads();

    const int cid = blockIdx.x * 64 + lane;
    if (cid >= A.n_chunks) return;
    const SdChunk ch = A.chunks[cid];
    const int len = A.ctg_len[ch.ctg];
    const uint8_t *seq = A.bases + A.ctg_off[ch.ctg];
    const int T = A.T, W = A.W, CAPW = W - 2;

#define RING(i) g_ring[(size_t)((i) & MASK) * NL + gl]
#define CW(t) g_cw[(size_t)(t) * NL + gl]
#define CV(t) g_cv[(size_t)(t) * NL + gl]
#define SLOT(s) g_slot[(size_t)((s) & MASK) * NL + gl]

    // ---- where to start: W-2 word emissions before (chunk start - 2W) ------------------------------
    int u = 0;
    if (ch.start > 0) {
        int y = ch.start - 2 * W;
        if (y > 2) {
            int need = CAPW, run = 0, p = y - 1;
            // walking down, `run` = number of consecutive ACGT bases at [p, p+run); a word ends at q
            // (bases q-2..q) for every q with a run of >= 3 ending there
            for (; p >= 0; --p) {
                if (s_lut[seq[p]] < 4) {
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
        if (now >= rec_from) emit(minstart, minstart + (int)(sl >> 20) + 3);
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
        const int b = i < len ? s_lut[word & 0xFFu] : 4;
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
                    CW(s) = (uint32_t)c;
                    rw -= c;
                    if (L > size) {
                        --L;
                        const int d = CV(s) - 1;
                        CV(s) = (uint32_t)d;
                        rv -= d;
                    }
                }
                RING(front + size) = (uint8_t)t;
                ++size;
                ++L;
                {
                    const int c = CW(t);
                    CW(t) = (uint32_t)(c + 1);
                    rw += c;
                    const int d = CV(t);
                    CV(t) = (uint32_t)(d + 1);
                    rv += d;
                    if ((d + 1) * 10 > T << 1) {                        // :79
                        int s;
                        do {
                            s = RING(front + size - L);
                            const int e = CV(s) - 1;
                            CV(s) = (uint32_t)e;
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
                        CV(tt) = (uint32_t)(c + 1);
                        r += c;
                        const int new_l = size - k - 1;
                        if (r * 10 > T * new_l) {                       // :112
                            while (fold > k) {                          // :113-117 as a running maximum
                                --fold;
                                const uint32_t sl = SLOT(start + fold);
                                if (sl) {
                                    const int pr = (int)(sl & 0xFFFFFu), pl = (int)(sl >> 20);
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
                                SLOT(ps) = (uint32_t)r | ((uint32_t)new_l << 20);
                            }
                        }
                    }
                    for (int k = 0; k <= i0; ++k) {                     // undo the in-place use of cv as c[]
                        const int tt = RING(front + k);
                        CV(tt) = CV(tt) - 1;
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
//   * cv / rv / rw / L are not maintained at all: find_perfect over EVERY suffix of the window, without the gate of :149,
//     is the same function (a suffix inside v is never a candidate of :112, and a candidate implies the gate: DESIGN.md
//     section 4.2b), so v only ever decided WHEN to call.  That decision is a per-lane lower bound M of the smallest margin
//     T (q - 1) - 10 r over the suffixes that hold m + 1 copies of some word (see its declaration): no pass while M >= 0.
//   * find_perfect (:104-128): lane j takes window position j of the owning lane; suffix scores r_j
//     come from 6 ballots (equal-word mask), mbcnt and a suffix-sum scan; the running maximum over P entries /
//     earlier candidates is a max-scan of exact integer ratio keys (sd_ratio_key).  Every pass leaves the exact minimum
//     margin behind as the owner's new M.
//   * P occupancy is a 64-bit mask per lane (bit = start & 63), so save_masked_regions (:88-102) and the
//     N flush (:153) are rotates / ctz instead of list walks.
//     Per-lane LDS state is just the ring and the 64 byte counters (8.4 KB per wave): 19 waves per CU.
// The waves stay resident and every lane takes chunks from one queue (low-complexity chunks first, one per wave):
// see "jobs" in the kernel.
// ---------------------------------------------------------------------------------------------------
#ifndef SD_EQT
#define SD_EQT 1
#endif
// Lanes of ONE wave hand data to each other through LDS (the equal-word tables) and through global memory (P slots): the
// hardware executes a wave's memory instructions in order, and this keeps the COMPILER from reordering them — release / acquire
// fences at wavefront scope around a wave barrier.  No instruction comes out of it (checked in the ISA: the s_waitcnt the
// surrounding loads and stores need anyway are all there is).
#define SD_LDS_ORDER()                                                \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        \
        __builtin_amdgcn_wave_barrier();                              \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        \
    } while (0)
struct SdLds64 {
    uint32_t cw[16][64];       // [3-mer >> 2][lane]: four byte counters (3-mer & 3) = copies of the 3-mer in the window
    uint8_t ring[64][64];      // [lane][(absolute word index & 63) ^ swizzle(lane)]: lanes in phase fall on distinct banks, counted in 32 banks per half wave or in 64
#if SD_EQT
    uint32_t eqt[64];          // find_perfect: [3-mer] -> lanes of one half of the wave that hold it (zero between passes)
#endif
};
// byte offset into SdLds64::ring of word index i of the lane whose key is X = 64 lane | 4 swizzle(lane): one 3-input logic op.
// swizzle = ((lane >> 2) & 15) ^ 8 ((lane >> 1) & 1): the 16 lanes that share lane & 1 within a half wave (row bases 64 bytes apart =
// 16 dwords: banks 0 or 16 of 32) get 16 different dwords, and so do the 16 lanes that share lane & 3 (64 banks); with the plain
// (lane >> 2) & 15 of the first version rocprofv3 counted 200 M more LDS bank-conflict cycles per launch
#ifndef SD_WPB
#define SD_WPB 1                // waves per workgroup of sdust_w64 (independent waves; 2, 5 and 10 measured slower: a workgroup's waves crowd onto the same SIMDs)
#endif
#define SD_RING_KEY(l) ((uint32_t)(l) << 6 | ((((uint32_t)(l) >> 2) & 15u) ^ (((uint32_t)(l) >> 1) & 1u) << 3) << 2)
#define SD_RING_OFF(X, i) (((X) & ~63u) | (((uint32_t)(i) ^ (X)) & 63u))
__device__ __forceinline__ uint32_t sd_ring_off(uint32_t X, uint32_t i)       // SD_RING_OFF as the single instruction it is: 63 ? i ^ X : X
{
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, 63 bitop3:0x6c" : "=v"(r) : "v"(i), "v"(X));
    return r;
}

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

__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x)
{
#define SD_MAX_STEP(ctrl, rmask, bound) { const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, (ctrl), (rmask), 0xF, (bound)); x = y > x ? y : x; }
    SD_MAX_STEP(DPP_ROW_SHR(1), 0xF, true)
    SD_MAX_STEP(DPP_ROW_SHR(2), 0xF, true)
    SD_MAX_STEP(DPP_ROW_SHR(4), 0xF, true)
    SD_MAX_STEP(DPP_ROW_SHR(8), 0xF, true)
    SD_MAX_STEP(DPP_ROW_BCAST15, 0xA, false)
    SD_MAX_STEP(DPP_ROW_BCAST31, 0xC, false)
#undef SD_MAX_STEP
    return x;
}
// Order-preserving integer key of the ratio r / l, 0 <= r < 2^11, 1 <= l <= 64: floor(r * 2^13 / l) < 2^24.  Two different
// ratios with denominators <= 64 differ by at least 1 / 4096, i.e. by at least 2 after scaling: their keys differ and
// are ordered like the ratios; equal ratios give equal keys.  (The reference cross-multiplies, src/sdust/sdust.c:115,118.)
__device__ __forceinline__ uint32_t sd_ratio_key(uint32_t r, uint32_t l) { return ((r & 0x7FFu) << 13) / (l & 0x7Fu); }
// vdst[lane] = val (one VALU op instead of compare + select); lane and val are wave-uniform
__device__ __forceinline__ int sd_writelane(int vdst, int val, int lane)
{
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(vdst) : "s"(val), "s"(lane) : "m0");
    return vdst;
}
__device__ __forceinline__ int wave_min_all(int x)      // minimum over the wave, in lane 63
{
    // one v_min_i32 with a DPP source per step: lanes without a source in their row keep their value (bound_ctrl off)
    asm volatile("s_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(x));
    return x;
}
typedef uint32_t sd_v16u __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned long long sd_ballot(bool x) { return __builtin_amdgcn_ballot_w64(x); }
__device__ __forceinline__ bool sd_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0; }
constexpr int SD_NEVER = 0x7fffffff;

// ---------------------------------------------------------------------------------------------------
// sdust_dense: the kernel for chunks that lie INSIDE a repeat array (sd_prep's sample of the chunk: few distinct 3-mers).
// There find_perfect runs, and inserts, at every base; done wave-cooperatively for one lane at a time (sdust_w64) that
// costs a wave ~150 instructions per base, and a 5 Mb satellite array is 3 000 consecutive such chunks.  Here the 64
// lanes of a wave each take one such chunk and every lane walks its own window (the reference's loop, :104-128): all
// lanes need it at every step, so nothing diverges, ~15 instructions per base and lane.
//
// What makes the walk simple (tools/sim/sdust_trigger_sim.c checks it on millions of steps for many (T, W)):
//   With m = floor(T / 5), v (:79-85) is the longest suffix of the window in which no 3-mer occurs more than m times.  A
//   suffix of q words inside v has score r <= q (m - 1) / 2 (and <= q (q - 1) / 2), hence 10 r <= T (q - 1): it is never a
//   candidate of :112.  And a candidate of q > L words gives 10 rw >= 10 r > T (q - 1) >= T L: the gate of :149 holds by
//   itself.  So find_perfect over EVERY suffix of the window, at every word, without the gate, is the same function, and
//   v, L, rv, cv, rw, cw need not exist: the walk starts at the newest word with r = 0.
// The walk needs, for the word at each window position, the number of equal words behind it (newer): r(suffix) is the sum
// of those over the suffix.  Every 3-mer has a push counter G (never decremented, mod 256); a ring slot keeps the word and
// the value of its counter right after its own push: equal newer words = G[word] now - that value.  No counts to build
// up and take down again per walk, no read-modify-write chains: the walk only READS the ring, G and the P slots, eight
// positions at a time (independent LDS reads, one wait per batch).
// P is the slot ring + occupancy mask of sdust_w64 (slot = r | l << 16, compared by cross-multiplication as in :115,118).
// All state in LDS, lane-minor ([index][lane]: bank-conflict free for any per-lane index).
// ---------------------------------------------------------------------------------------------------
struct SdDenseLds {
    uint16_t ring[64][64];     // [word index & 63][lane]: word | (G[word] after its push) << 8
    uint8_t G[16][64][4];      // [word >> 2][lane][word & 3]: pushes of the word so far (mod 256)
    uint32_t slot[64][64];     // [start & 63][lane]
};

__global__ __launch_bounds__(64) void sdust_dense(SdArgs A, const uint32_t *list, int n_list, uint32_t *started)
{
    __shared__ SdDenseLds S;
    const int lane = threadIdx.x;
    // the host launches the main kernel once this one is on the chip (its blocks need 28 KB of LDS each: see the launch)
    if (started && blockIdx.x == 0 && lane == 0) __hip_atomic_store(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int T = A.T, W = A.W, CAPW = W - 2;
    const int H = n_list;
    for (int job = blockIdx.x; job * 64 < H; job += gridDim.x) {
        for (int i = 0; i < 16; ++i) *reinterpret_cast<uint32_t *>(S.G[i][lane]) = 0;
        for (int i = 0; i < 64; ++i) S.slot[i][lane] = 0;
        const bool have = job * 64 + lane < H;
        const int cid = have ? (int)list[job * 64 + lane] : 0;
        const SdChunk ch = have ? A.chunks[cid] : SdChunk{0, 0, 0};
        const int len = have ? A.ctg_len[ch.ctg] : 0;
        const uint8_t *seq = A.bases + A.ctg_off[ch.ctg];
        int u = have ? sd_find_start(A, ch, seq) : 0;
        const bool live = have && u >= 0;             // u < 0: the word-count table is needed, the host reruns with it
        if (u < 0) u = 0;
        const int stop = !live ? u : (ch.end == len ? len + 1 : ch.end);   // the last chunk also runs the sentinel step i == len
        const int rec_from = ch.start;

        int l = 0, p = -1;                            // p: index of the newest word (count of pushes - 1); the window is words max(0, p - (W-3)) .. p
        unsigned t = 0;
        unsigned long long occ = 0;
        int minstart = 0;
        bool have_last = false;
        uint32_t last_s = 0, last_f = 0, n_out = 0;
        uint2 *out = A.out + (size_t)cid * A.cap;

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
        // save_masked_regions(start) when it does something: occ != 0 && minstart < start   (:88-102).  Slots that hold no
        // entry are kept at zero (r = 0: "nothing here" for the walk below), so the slots of the entries that leave are cleared.
        auto save_evict = [&](int start, int now) {
            const uint32_t sl = S.slot[minstart & 63][lane];
            if (now >= rec_from) emit(minstart, minstart + (int)(sl >> 16) + 3);
            const int gone = start - minstart;           // starts minstart .. start-1 leave the window
            unsigned long long r = rotr64(occ, minstart & 63);   // bit 0 <-> minstart
            unsigned long long out_m = gone >= 64 ? r : r & ~(~0ull << gone);
            while (out_m) {
                S.slot[(minstart + __builtin_ctzll(out_m)) & 63][lane] = 0;
                out_m &= out_m - 1;
            }
            if (gone >= 64) {
                occ = 0;
            } else {
                r &= ~0ull << gone;
                occ = rotl64(r, minstart & 63);
            }
            if (occ) minstart = start + __builtin_ctzll(rotr64(occ, start & 63));
        };

        uint32_t word = 0;
        if (live && u < len) word = *reinterpret_cast<const uint32_t *>(seq + (u & ~3)) >> (8 * (u & 3));
        int i = u;
        while (sd_any(live && i < stop)) {
            const bool on = live && i < stop;
            bool isword = false;
            int start = 0;
            if (on) {
                if ((i & 3) == 0 && i != u && i < len) word = *reinterpret_cast<const uint32_t *>(seq + i);
                const int b = i < len ? nt4_code(word & 0xFFu) : 4;
                word >>= 8;
                if (b < 4) {
                    ++l;
                    t = (t << 2 | (unsigned)b) & 63u;                           // :144
                    if (l >= 3) {
                        isword = true;
                        start = (l - W > 0 ? l - W : 0) + (i + 1 - l);           // :146
                        if (occ != 0 && minstart < start) save_evict(start, i);  // :147
                        // shift_window (:66-86): the oldest word just falls out of the index range; push t
                        ++p;
                        const unsigned g = (unsigned)(S.G[t >> 2][lane][t & 3] + 1) & 0xFFu;
                        S.G[t >> 2][lane][t & 3] = (uint8_t)g;
                        S.ring[p & 63][lane] = (uint16_t)(t | (g << 8));
                    }
                } else {
                    int st = (l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l);      // :152
                    while (occ) {                                                // :153
                        if (minstart >= st) st = minstart + 1;
                        save_evict(st, i);
                        ++st;
                    }
                    l = 0;
                    t = 0;                                                       // :154 — window and counters kept
                }
            }
            // ---- find_perfect (:104-128) over every suffix of the window, newest word first; branch-free per position
            // (selects instead of jumps: all lanes do the same thing).  The running maximum (max_r, max_l) starts at (0, 1):
            // "r max_l >= max_r l" and "pr max_l > max_r pl" are then what :115,:118 test with max_r == 0, and an empty slot
            // (r = 0, l = 0) never wins.
            if (sd_any(isword)) {
                const int size = p + 1 < CAPW ? p + 1 : CAPW;
                const int limit = isword ? size : 0;                             // positions new_l < limit exist
                const int base = start + size - 1;                               // start value of the newest window position
                int r = 0, max_r = 0, max_l = 1, lowest = -1;                    // lowest: largest new_l inserted
                const uint32_t p63 = (uint32_t)p & 63u, b63 = (uint32_t)base & 63u;
                unsigned long long newbits = 0;                                  // bit new_l: inserted there
                for (int kb = 0; kb < CAPW; kb += 8) {
                    uint32_t e[8], gv[8], sl[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) e[j] = S.ring[(p63 - (uint32_t)(kb + j)) & 63u][lane];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned w = e[j] & 63u;
                        gv[j] = S.G[w >> 2][lane][w & 3];
                        sl[j] = S.slot[(b63 - (uint32_t)(kb + j)) & 63u][lane];
                    }
                    uint32_t nb8 = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int new_l = kb + j;                                // words in the suffix - 1 (:111)
                        if (new_l >= CAPW) continue;                             // (compile-time: the last batch is shorter)
                        const bool livej = new_l < limit;
                        const int c = (int)((gv[j] - (e[j] >> 8)) & 0xFFu);       // equal words behind this one
                        r += livej ? c : 0;
                        const int pr = (int)(sl[j] & 0xFFFFu), pl = (int)(sl[j] >> 16);
                        const bool fold = __mul24(pr, max_l) > __mul24(max_r, pl);               // :113-117: entries with start >= i + start
                        max_r = fold ? pr : max_r;
                        max_l = fold ? pl : max_l;
                        const bool ins = (int)livej & (int)(__mul24(r, 10) > T * new_l) & (int)(__mul24(r, max_l) >= __mul24(max_r, new_l));   // :112, :118
                        max_r = ins ? r : max_r;
                        max_l = ins ? new_l : max_l;
                        nb8 |= ins ? 1u << j : 0u;
                        if (ins) S.slot[(b63 - (uint32_t)new_l) & 63u][lane] = (uint32_t)r | ((uint32_t)new_l << 16);
                    }
                    newbits |= (unsigned long long)nb8 << kb;
                }
                if (newbits) {
                    // position new_l <-> start value base - new_l <-> occupancy bit (base - new_l) & 63
                    const unsigned long long rev = __brevll(newbits);            // bit 63 - new_l
                    const bool was_empty = occ == 0;
                    occ |= rotr64(rev, (63 - base) & 63);
                    lowest = base - (63 - __builtin_clzll(newbits));
                    if (was_empty || lowest < minstart) minstart = lowest;
                }
            }
            ++i;
        }
        if (have) {
            if (have_last) {
                if (n_out < A.cap) out[n_out] = make_uint2(last_s, last_f);
                ++n_out;
            }
            A.out_n[cid] = n_out;
            if (n_out > A.cap) atomicMax(A.ovf, n_out);
        }
    }
}


// Requires 1 <= m = 2T/10 and T <= 100000 (24-bit products exact); other thresholds take the legacy kernel.
template <bool STATS>
__global__ __launch_bounds__(64 * SD_WPB, STATS ? 4 : 5) void sdust_w64(SdArgs A)      // (5 waves per SIMD: at most 96 VGPRs)
{
    // SD_WPB independent waves per workgroup (no barrier, nothing shared): a CU holds 16 workgroups at most, whatever their size
#if SD_WPB == 1
    __shared__ SdLds64 S;
#else
    __shared__ SdLds64 SS[SD_WPB];
    SdLds64 &S = SS[threadIdx.x >> 6];
#endif
    const int lane = threadIdx.x & 63;
    const size_t wave_id = (size_t)blockIdx.x * SD_WPB + (threadIdx.x >> 6);
    for (int i = 0; i < 16; ++i) S.cw[i][lane] = 0;
    for (int i = 0; i < 16; ++i) reinterpret_cast<uint32_t *>(&S.ring[lane][0])[i] = 0;     // (ring bytes index eqt: always < 64)
#if SD_EQT
    S.eqt[lane] = 0;
#endif

    const int T = A.T, W = A.W, CAPW = W - 2;
    const int q_len = A.q_len_dev ? (int)__builtin_amdgcn_readfirstlane((int)*A.q_len_dev) : A.q_len;
    const int m = (T << 1) / 10;                     // cv[t]*10 > T<<1  <=>  cv[t] > m   (:79)
    // find_perfect, lane <-> window position: with a full window the suffix of this lane has l_full words after its first
    const int l_full = CAPW - 64 + lane;
    const uint32_t m_full = l_full >= 2 ? (uint32_t)((0x100000000ull + (unsigned)l_full - 1ull) / (unsigned)l_full) : 0u;
    // smallest margin of a suffix with exactly m + 1 copies of one word and every other word at most m times:
    // all of its words equal (q = m + 1): m (T - 5m - 5); with o >= 1 other words: T (m + o) - 5 m (m + 1) - 10 g(o),
    // g(o) = most pairs o words make with at most m copies each
    // (bound_eq < bound_ne: the difference at o is T o - 10 g(o) >= o (T - 5m + 5) > 0).  T < 5 (m = 0): the pass always runs.
    const int bound_eq = m ? m * (T - 5 * m - 5) : -1;
    int bound_ne = m ? 1 << 29 : -1;
    for (int oo = 1; m && oo <= CAPW; ++oo) {
        const int g = (oo / m) * (m * (m - 1) / 2) + (oo % m) * (oo % m - 1) / 2;
        const int v = T * (m + oo) - 5 * m * (m + 1) - 10 * g;
        bound_ne = v < bound_ne ? v : bound_ne;
    }

    // ---- jobs.  The grid is as many waves as fit on the chip at once; every LANE takes chunks from one global
    // queue (A.perm order: chunks sampled as low-complexity first) until it is empty, so nothing waits for a
    // "round" of workgroups to retire and a wave slowed down by a lane inside a satellite or telomere array
    // (find_perfect runs, and inserts, at every base there) just takes fewer chunks.  K counts the steps of the
    // wave; a lane that starts a chunk at step K0 from position u keeps ubase = u - K0, so that step K is
    // position ubase + K for it.  Chunks start on 64-step boundaries (u is 64-byte aligned: all lanes stay on
    // the same phase of the 64-byte blocks).
    bool done = false, hasjob = false;
    int cid = 0, cur = 0, ubase = 0;                 // cid .. cur: the chunks of the lane's current run (contiguous, same contig)
    uint32_t blk64 = 0;                              // (contig offset + ubase) / 64: the lane's stream in 64-byte blocks
    // steps >= endk see a non-base: the end of the sequence (:141, flushes P and records it) or the end of the
    // chunk (flushes P without recording: those intervals belong to the next chunk).  Only steps
    // recfrom_k <= k < nrun record.
    int endk = 0;
    bool islast = false;                             // the chunk ends its contig: nrun = endk + 1 steps, else endk
#define SD_NRUN (endk + (islast ? 1 : 0))

    // ---- per-lane sequential state; word indices are counts of pushed words since the chunk's warm-up start
    int p = -1;                 // index of the newest word in the window
    int o = 0;                  // index of the oldest word in the window (size = p - o + 1)
    // No v, L, rv, rw (section 4.2b of DESIGN.md: find_perfect over every suffix of the window, without the gate of :149, is the
    // same function).  What decides WHEN the cooperative pass runs is a lower bound M of the smallest margin T (q - 1) - 10 r
    // over the suffixes that hold m + 1 copies of some 3-mer (only those can be candidates of :112): no pass while M >= 0.
    //   push of a word with ct copies in the window:  every such suffix gets one word longer and at most ct pairs richer,
    //       M += T - 10 ct;  new ones appear only when ct >= m:
    //   ct >= m: a suffix that was not hot before holds exactly m + 1 copies of t (the m + 1 newest; one more and it held
    //       m + 1 before the push) and every other word at most m times: margin >= bound_eq if its m + 1 newest words are
    //       all t (then the previous word is t), >= bound_ne otherwise (both computed above);
    //   the pass leaves the exact minimum over all suffixes of >= m + 1 words behind.
    // tools/sim/sdust_trigger_sim.c ("hot-suffix tracker") checks the bound on 10^7 steps for a dozen (T, W): never above the
    // true minimum; in random sequence it asks for 0.39 passes per wave-step — the gate with its exact L asked for 0.15, but
    // needed 1.06 cooperative trims per wave-step to keep L.
    int bound_eq_v = bound_eq, bound_ne_v = bound_ne;           // both in vector registers: the select needs no move per word
    asm volatile("" : "+v"(bound_eq_v), "+v"(bound_ne_v));
    int M = -1;                 // (no bound yet: the first word runs the pass)
    unsigned tprev = 0xFFu;     // the word pushed before this one
    unsigned s_pref = 0;        // ring[o]: the word the next pop removes
    unsigned long long occ = 0; // occupied P slots, bit = start & 63
    int minstart = 0;
    int evict_k = SD_NEVER;     // first step at which P needs attention (eviction, or a flush at a non-base)
    int LN = -1;                // step of the last non-base before the current group of 4 steps
    bool have_last = false;
    uint32_t last_s = 0, last_f = 0, n_out = 0;

    auto emit = [&](int ps, int pf) {               // :93-99 on the lane-local list
        if (have_last && ps <= (int)last_f) {
            if (pf > (int)last_f) last_f = (uint32_t)pf;
        } else {
            if (have_last) {
                if (n_out < (uint32_t)(cur - cid + 1) * A.cap) A.out[(size_t)cid * A.cap + n_out] = make_uint2(last_s, last_f);   // the rows of a run are contiguous
                ++n_out;
            }
            have_last = true;
            last_s = (uint32_t)ps;
            last_f = (uint32_t)pf;
        }
    };
    // save_masked_regions(start) when it does something: occ != 0 && minstart < start   (:88-102)
    auto save_evict = [&](int start, int nowk) {
        // P slots of the lane: one 256-byte row in global memory (only slots whose occupancy bit is set are ever read).
        // A slot may have been written by ANOTHER lane of this wave (find_perfect below): every store of the wave has to be
        // acknowledged by L2, where the sc1 loads look, before a slot is read — waited for here, in front of the load, so
        // that the store itself does not hold the wave up.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SD_LDS_ORDER();
        const uint32_t sl = __hip_atomic_load(&A.slots[(wave_id * 64 + lane) * 64 + (minstart & 63)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nowk >= A.chunks[cid].start - ubase && nowk < SD_NRUN) emit(minstart, minstart + (int)(sl >> 24) + 3);
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

    // Bases are consumed one 64-byte block per lane per 64 steps, in two 32-byte requests (2 x dwordx4 each): with one
    // dword per 4 steps the ~40 k concurrent per-lane streams of an XCD overflowed its 4 MB L2 and each line was
    // re-fetched ~6 times (rocprofv3 FETCH_SIZE, profiles/).  A block never leaves its contig: contigs start 64-byte aligned.
    sd_v16u blk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // the lane's current 64-byte block; each half is refilled half a block ahead
    int lenk = 0;                                  // contig length as a step (len - ubase): blocks are fetched while the contig goes on
    uint32_t pcn = 0x04040404u;                      // codes of the previous 4 positions (before u: non-bases, l = 0)
    uint8_t *const ring0 = &S.ring[0][0];
    const uint32_t ringX = SD_RING_KEY(lane);
    const uint32_t bit_lo = lane < 32 ? 1u << lane : 0u, bit_hi = lane < 32 ? 0u : 1u << (lane - 32);

    unsigned st_plain = 0, st_steps = 0, st_fp = 0, st_full = 0, st_jobs = 0, st_iter = 0, st_qt = 0;
    const unsigned long long st_t0 = STATS ? wall_clock64() : 0ull;
    for (int k64 = 0;; k64 += 64) {
      // ---- a lane whose chunk ends inside the coming block runs on into the next chunk of the contig if nobody has
      // taken it yet: no warm-up, no reset, the records go on in the same (contiguous) rows.  Only when that fails does it
      // publish and take another chunk from the queue, which hands out every 8th chunk first so that runs have room.
      // (The next chunk of the contig is [end, min(end + chunk, contig length)): no loads.  The round trip of the claim
      // stalls this wave only; the SIMD's other waves keep the VALU busy — asking a block ahead changed nothing.)
      auto run_on = [&]() {
          if (A.run_on && sd_any(hasjob && !islast && endk < k64 + 64)) {
              while (hasjob && !islast && endk < k64 + 64) {
                  if (atomicCAS(&A.claim[cur + 1], 0u, 1u) != 0u) break;
                  ++cur;
                  endk = A.chunks[cur].end - ubase;          // (chunks differ in size: the last part of the queue is made of short ones)
                  islast = endk == lenk;
              }
          }
      };
      run_on();
      // ---- lanes whose run is finished publish it and take the next free chunk from the queue
      const bool need = !done && k64 >= SD_NRUN;
      if (sd_any(need)) {
          if (need && hasjob) {
              if (have_last) {
                  if (n_out < (uint32_t)(cur - cid + 1) * A.cap) A.out[(size_t)cid * A.cap + n_out] = make_uint2(last_s, last_f);
                  ++n_out;
              }
              A.out_n[cid] = n_out;                    // (the other chunks of the run keep the 0 the host put there)
              if (n_out > (uint32_t)(cur - cid + 1) * A.cap) atomicMax(A.ovf, n_out);
          }
          bool want = need;
          int got = -1;
          const unsigned long long st_q0 = STATS ? wall_clock64() : 0ull;
          while (sd_any(want)) {
              if (STATS) ++st_iter;
              const unsigned long long wmask = sd_ballot(want);
              const int first = __builtin_ctzll(wmask);
              int base = 0;
              if (lane == first) base = (int)atomicAdd(A.queue, (uint32_t)__popcll(wmask));
              base = rdlane(base, first);
              if (want) {
                  const int idx = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u));
                  if (idx < 0 || idx >= q_len) {
                      want = false;                    // the queue is empty
                  } else {
                      const uint32_t c = A.perm ? A.perm[idx] : (uint32_t)idx;
                      if (c < (uint32_t)A.n_chunks && atomicCAS(&A.claim[c], 0u, 1u) == 0u) {
                          got = (int)c;
                          want = false;
                      }
                  }
              }
          }
          if (STATS) st_qt += (unsigned)(wall_clock64() - st_q0);
          if (need) {
              hasjob = got >= 0;
              done = !hasjob;
              endk = 0;
              islast = false;
              blk = sd_v16u{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
              lenk = 0;
              occ = 0;
              evict_k = SD_NEVER;
              if (hasjob) {
                  cid = cur = got;
                  const SdChunk ch = A.chunks[cid];
                  const int u0 = sd_find_start(A, ch, A.bases + A.ctg_off[ch.ctg]);
                  const int u = u0 < 0 ? -1 : (u0 & ~63);   // starting a little earlier is still exact, and keeps every lane on the same 64-byte phase; < 0: word-count table needed
                  have_last = false;
                  n_out = 0;
                  if (u >= 0) {
                      ubase = u - k64;
                      blk64 = (uint32_t)((A.ctg_off[ch.ctg] + u) >> 6) - (uint32_t)(k64 >> 6);
                      endk = (ch.end - u) + k64;
                      islast = (ch.end - u) + k64 == A.ctg_len[ch.ctg] - ubase;
                      p = -1; o = 0; s_pref = 0; M = -1; tprev = 0xFFu;
                      LN = k64 - 1;
                      pcn = 0x04040404u;
                      for (int i = 0; i < 16; ++i) S.cw[i][lane] = 0;
                      lenk = A.ctg_len[ch.ctg] - ubase;
                      {
                          const uint4 *q = reinterpret_cast<const uint4 *>(A.bases + A.ctg_off[ch.ctg] + u);      // u < ch.end <= contig length
                          const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                          blk.s0 = q0.x; blk.s1 = q0.y; blk.s2 = q0.z; blk.s3 = q0.w; blk.s4 = q1.x; blk.s5 = q1.y; blk.s6 = q1.z; blk.s7 = q1.w;
                          blk.s8 = q2.x; blk.s9 = q2.y; blk.sa = q2.z; blk.sb = q2.w; blk.sc = q3.x; blk.sd = q3.y; blk.se = q3.z; blk.sf = q3.w;
                      }
                  } else {
                      endk = k64;                        // empty job (the host reruns with the word-count table)
                      islast = true;                     // (and never runs on)
                  }
              }
          }
          if (STATS) ++st_jobs;
          if (sd_ballot(!done) == 0) break;
          run_on();          // a chunk just taken may itself end inside the coming block (contig starts, tiny chunks)
      }
      // Bases are consumed one 64-byte block per lane per 64 steps.  The block lives in 16 registers that are read with
      // a uniform index; each half is refilled once it is dead — words 0-7 at group 12 with the first half of the NEXT
      // block, words 8-15 after group 15 — i.e. 16 and 32 steps before it is needed, from the same contig
      // whether or not the lane's own chunk goes on (if it runs on into the next chunk the data is there; if it takes
      // another chunk, the block is loaded afresh).
      // The packed decode below knows letters only.  A byte <= 3 (seq_nt4_table maps 0..3 to themselves) clears the
      // top six bits of its byte lane in the AND of the words; letters never do (they all carry 0x40): such a half
      // (and, harmlessly, a few others: zero padding behind a contig, '-' or '*' next to letters) is decoded byte by byte.
#define SD_HAS_LOW_BYTE(a) ((((a) & 0xFCFCFCFCu) - 0x01010101u) & ~((a) & 0xFCFCFCFCu) & 0x80808080u)
      const bool more = k64 + 64 < lenk;     // (done lanes and lanes without a chunk: lenk = 0)
      bool slow_any = sd_any(SD_HAS_LOW_BYTE(blk.s0 & blk.s1 & blk.s2 & blk.s3 & blk.s4 & blk.s5 & blk.s6 & blk.s7) != 0);
#pragma clang loop unroll(disable)
      for (int g = 0; g < 16; ++g) {
        const int k4 = k64 + 4 * g;
        if (g == 8) slow_any = sd_any(SD_HAS_LOW_BYTE(blk.s8 & blk.s9 & blk.sa & blk.sb & blk.sc & blk.sd & blk.se & blk.sf) != 0);
        if (g == 12) {                               // (as late as the latency allows: the other half of the sector follows 16 steps later)
            if (more) {
                const uint4 *q = reinterpret_cast<const uint4 *>(A.bases + ((uint64_t)(blk64 + (uint32_t)(k64 >> 6) + 1u) << 6));
                const uint4 q0 = q[0], q1 = q[1];
                blk.s0 = q0.x; blk.s1 = q0.y; blk.s2 = q0.z; blk.s3 = q0.w; blk.s4 = q1.x; blk.s5 = q1.y; blk.s6 = q1.z; blk.s7 = q1.w;
            }
        }
        const uint32_t word = blk[g];
        // ---- 4 positions -> 4 codes (bits 0-1 base, bit 2 non-base) ------------------------------------
        uint32_t cn;
        if (!slow_any) {
            // fold case, look the low 3 bits up (A1 C3 T4 G7): the code, and the letter that must be there
            const uint32_t y = word & 0xDFDFDFDFu, idx = y & 0x07070707u;
            const uint32_t code = __builtin_amdgcn_perm(0x02000003u, 0x01000000u, idx);      // idx 7..4 | 3..0
            const uint32_t expd = __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, idx);
            const uint32_t d = y ^ expd;
            const uint32_t nz = ((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d;                        // bit 7 of a byte: byte != 0
            cn = code | ((nz >> 5) & 0x04040404u);
        } else {
            cn = (uint32_t)nt4_code(word & 0xFFu) | ((uint32_t)nt4_code((word >> 8) & 0xFFu) << 8) |
                 ((uint32_t)nt4_code((word >> 16) & 0xFFu) << 16) | ((uint32_t)nt4_code(word >> 24) << 24);
        }
        if (sd_any(k4 + 4 > endk)) {                  // end of the sequence / of the chunk inside this group
            asm volatile("; chunk end" ::);          // (keeps the compiler from doing this arithmetic in every group)
            const int rem = endk - k4 > 0 ? endk - k4 : 0;
            if (rem < 4) cn |= 0x04040404u << (8 * rem);
        }
        // ---- 4 words: t = three consecutive codes (:144), bit 6 = not a word (l < 3), bit 7 = non-base ----
        uint32_t tw;
        {
            const uint32_t y1 = __builtin_amdgcn_alignbyte(cn, pcn, 3);      // code of the previous position
            const uint32_t y2 = __builtin_amdgcn_alignbyte(cn, pcn, 2);      // and of the one before
            const uint32_t t = (cn & 0x03030303u) | ((y1 & 0x03030303u) << 2) | ((y2 & 0x03030303u) << 4);
            const uint32_t nany = (cn | y1 | y2) & 0x04040404u;
            tw = t | (nany << 4) | ((cn & 0x04040404u) << 5);
            pcn = cn;
        }
        const uint32_t nmask = tw & 0x80808080u;
        const uint32_t tw8 = tw >> 8;
        const bool grp_n = sd_any(nmask != 0);
        // next step at which P needs attention: the oldest start leaves the window (i >= minstart + W, :146-147), or
        // the next non-base of this group among the bytes selected by `above`
        auto next_evict = [&](uint32_t above) {
            const uint32_t mm = nmask & above;
            const int nn = mm ? k4 + (__builtin_ctz(mm) >> 3) : SD_NEVER;
            const int ev = minstart - ubase + W;
            return occ ? (ev < nn ? ev : nn) : SD_NEVER;
        };
        if (grp_n) evict_k = next_evict(0xFFFFFFFFu);
        // One step.  PLAIN: every lane has a word at each of the 4 steps, every lane's window is full and no lane's P needs
        // attention before the group ends (6 of 7 groups in ordinary sequence): no gate, no maintenance test, every step pops.  Returns whether some lane's P got an
        // entry (evict_k changed: the rest of the group takes the general steps).
        auto word_step = [&](auto plain_c, const int kk) __attribute__((always_inline)) -> bool {
          constexpr bool PLAIN = decltype(plain_c)::value;
          const int k = k4 + kk;
          if (STATS) ++st_steps;
          const unsigned f = (tw >> (8 * kk)) & 0xFFu;
          // ---- P maintenance, rare: save_masked_regions (:147) when the oldest start leaves the window, and the
          // flush at a non-base (:152-153)
          if (!PLAIN) {   // (a plain divergent branch: the compiler's skip-if-no-lane is the wave-wide test)
              if (k >= evict_k) {
                  const uint32_t nb4 = nmask & (kk == 0 ? 0u : 0xFFFFFFFFu >> (32 - 8 * kk));          // non-bases before this step
                  const int lastN = ubase + (nb4 ? k4 + ((31 - __builtin_clz(nb4)) >> 3) : LN);
                  const int i = ubase + k;
                  if (f & 0x80u) {
                      const int l_old = i - 1 - lastN;
                      int st = (l_old - W + 1 > 0 ? l_old - W + 1 : 0) + (i + 1 - l_old);
                      while (occ) {
                          if (minstart >= st) st = minstart + 1;
                          save_evict(st, k);
                          ++st;
                      }
                  } else if (f < 64u) {
                      const int start = i + 1 - W > lastN + 1 ? i + 1 - W : lastN + 1;                            // :146
                      if (occ != 0 && minstart < start) save_evict(start, k);
                  }
                  evict_k = next_evict(kk >= 3 ? 0u : 0xFFFFFFFFu << (8 * (kk + 1)));
              }
          }
          const bool isword = PLAIN || f < 64u;
          const unsigned long long wordmask = PLAIN ? ~0ull : sd_ballot(f < 64u);
          if (isword) {
              // shift_window (:66-86) without cv / rv / rw: the two counters are byte fields of LDS dwords, updated by
              // atomics (one LDS op each instead of a read and a write); only the push needs the old value back
              const bool pop = PLAIN || p >= CAPW - 1;         // size >= W - 2 (:68): o = max(0, p - (W - 3)) at all times; plain groups: every window is full
              const unsigned s = s_pref;
              // byte field of the counter inside its dword: 8 * (word & 3); shifts use the low 5 bits of the amount only
              const unsigned sh_s = (s << 3) & 31u, sh_t = (f << 3) & 31u;
              (void)__hip_atomic_fetch_sub(&S.cw[s >> 2][lane], pop ? 1u << sh_s : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              const uint32_t old_t = __hip_atomic_fetch_add(&S.cw[f >> 2][lane], 1u << sh_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              o += pop ? 1 : 0;
              ++p;
              // :75 (the byte straight out of the group's dword or of its copy shifted by one byte: bits 0-7 or 16-23, which a
              // byte store can take without a shift)
              ring0[sd_ring_off(ringX, (uint32_t)p)] = (uint8_t)(((kk & 1) ? tw8 : tw) >> (8 * (kk & 2)));
              s_pref = ring0[sd_ring_off(ringX, (uint32_t)o)];
              const int ct = (int)__builtin_amdgcn_ubfe(old_t, f << 3, 8u);                    // cw[t]++   (:77), after the pop
              // the bound (see the declaration of M)
              const int adv = __mul24(ct, -10) + M + T;
              const int bn = f == tprev ? bound_eq_v : bound_ne_v;
              const int at_m = adv < bn ? adv : bn;
              M = ct < m ? adv : at_m;
              tprev = f;                                       // (the previous WORD: the window lives on across non-bases)
          }
          // the lanes whose bound does not exclude a candidate
          unsigned long long fp_todo = sd_ballot(M < 0) & wordmask;
          bool inserted = false;                               // (uniform) P of some lane got an entry: evict_k changed
          // ---- cooperative find_perfect (:104-128) ------------------------------------------------------
          // lane <-> window position j = 63 - lane, so that "suffix of the window" = "prefix of the wave" and
          // both scans are forward DPP scans (no LDS round trips).
          if (fp_todo) {
              if (STATS) st_fp += (unsigned)__popcll(fp_todo);
              unsigned long long todo = fp_todo;
              while (todo) {
                  const int ol = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                  todo &= todo - 1;
                  const int o_p = rdlane(p, ol), ws = rdlane(o, ol);
                  const int o_size = o_p - ws + 1;
                  const int j = 63 - lane;                                           // window position (0 = oldest)
                  uint32_t woff;
                  asm("v_bitop3_b32 %0, %1, %2, 63 bitop3:0x6c" : "=v"(woff) : "v"(ws + j), "s"(rdlane((int)ringX, ol)));   // sd_ring_off, the owner's key in an SGPR
                  const unsigned wj = ring0[woff];
                  // lanes holding the same word at a later window position (= lower lanes): the lanes of each half of the wave
                  // in turn set their bit in the table entry of their word, everybody reads the entry of its own word and clears
                  // it (the LDS operations of a wave execute in order, each for all its lanes)
                  const bool inwin = j < o_size;
                  const unsigned long long inb = ~0ull << (64 - o_size);             // (1 <= o_size <= 64)
#if SD_EQT
                  uint32_t *const te = &S.eqt[wj];
                  (void)__hip_atomic_fetch_or(te, inwin ? bit_lo : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  SD_LDS_ORDER();
                  const uint32_t eq_lo = __hip_atomic_load(te, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  SD_LDS_ORDER();
                  __hip_atomic_store(te, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  SD_LDS_ORDER();
                  (void)__hip_atomic_fetch_or(te, inwin ? bit_hi : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  SD_LDS_ORDER();
                  const uint32_t eq_hi = __hip_atomic_load(te, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  SD_LDS_ORDER();
                  __hip_atomic_store(te, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
                  // one ballot per bit of the word, kept as two 32-bit halves (per bit: sign-extended bit, compare, two 3-input logic ops)
                  uint32_t eq_lo = (uint32_t)inb, eq_hi = (uint32_t)(inb >> 32);
#pragma unroll
                  for (int bb = 0; bb < 6; ++bb) {
                      const int ext = __builtin_amdgcn_sbfe((int)wj, bb, 1);                     // all ones / zero
                      unsigned long long bal;
                      asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(bal) : "v"(ext));                  // ballot of the bit
                      // eq &= ~(ext ^ bal): one 3-input logic op per half (table 0x90 = src0 & ~(src1 ^ src2))
                      asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(eq_lo) : "v"(ext), "s"((uint32_t)bal));
                      asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(eq_hi) : "v"(ext), "s"((uint32_t)(bal >> 32)));
                  }
#endif
                  // suffix score r_j = inclusive prefix sum
                  // (mbcnt: set bits of the mask below this lane)
                  int below;
                  asm("v_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %2, %0" : "=&v"(below) : "v"(eq_lo), "v"(eq_hi));   // (opaque: a select, not a branch)
                  const int r = wave_scan_add(inwin ? below : 0);
                  const int new_l = o_size - j - 1;                                  // :111
                  // :112 (new_l < 64, T < 2^17)
                  const int margin = __mul24(T, new_l) - __mul24(r, 10);
                  // every suffix of the window is examined (:107 starts behind v: the suffixes inside v are never candidates); the
                  // exact minimum over those of at least m + 1 words becomes the owner's bound
                  const int mn = rdlane(wave_min_all(lane >= 64 - o_size + m ? margin : (1 << 29)), 63);   // in the window, new_l >= m
                  M = sd_writelane(M, mn, ol);
                  // a suffix of fewer than m + 1 words is never a candidate (r <= q (q - 1) / 2, 5 q <= 5 m <= T): with a minimum >= 0
                  // over the others nothing can be inserted, and the ballot of the candidates is not even taken
                  if (mn >= 0) continue;
                  const unsigned long long candmask = sd_ballot(margin < 0) & inb;
                  if (STATS) ++st_full;
                  const bool cand = (candmask >> lane) & 1ull;
                  int startv;                                                        // :146 for every lane's own state
                  {
                      int ln = LN;
                      asm volatile("" : "+v"(ln));                                   // (keeps this arithmetic inside the passes that have candidates)
                      const uint32_t nb4 = nmask & (kk == 0 ? 0u : 0xFFFFFFFFu >> (32 - 8 * kk));
                      const int lastN = ubase + (nb4 ? k4 + ((31 - __builtin_clz(nb4)) >> 3) : ln);
                      startv = ubase + k + 1 - W > lastN + 1 ? ubase + k + 1 - W : lastN + 1;
                  }
                  const int o_start = rdlane(startv, ol);
                  const unsigned long long o_occ = rdlane64(occ, ol);
                  uint32_t *orow = A.slots + (wave_id * 64 + ol) * 64;
                  const int sidx = (o_start + j) & 63;
                  const bool has_e = inwin && ((o_occ >> sidx) & 1ull);
                  if (o_occ) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (see save_evict)
                  SD_LDS_ORDER();
                  const uint32_t e = has_e ? __hip_atomic_load(&orow[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                  // Ratios r / l are compared through key = floor(r * 2^13 / l) (sd_ratio_key): exact for l <= 64.  All that
                  // :113-118 need of P is, for every start, the best ratio among the entries with a start at or after it.
                  const uint32_t key_e = e & 0xFFFFFFu;                              // 0: no entry with this start
                  // (the window is full almost always: new_l = lane - 64 + (W - 2) is then a constant of the lane and the division
                  // of the key a multiplication by its reciprocal, exact for numerators < 2^24 and divisors <= 64)
                  uint32_t key_c = 0u;
                  if (o_size == CAPW) key_c = cand ? (l_full == 1 ? ((uint32_t)r & 0x7FFu) << 13 : __umulhi(((uint32_t)r & 0x7FFu) << 13, m_full)) : 0u;
                  else key_c = cand ? sd_ratio_key((uint32_t)r, (uint32_t)new_l) : 0u;
                  // X_j = better of (existing entry with this start, candidate j); inclusive maximum over positions >= j
                  const uint32_t xs = wave_scan_max(key_e > key_c ? key_e : key_c);
                  // maximum over positions > j = the scan value one lane down (wave_shr:1; lane 0 gets 0)
                  const uint32_t sk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xs, 0x138, 0xF, 0xF, true);
                  const uint32_t km = sk > key_e ? sk : key_e;                       // :113-117: entries with start >= i + start
                  const bool ins = cand && key_c >= km;                              // :118
                  if (ins) __hip_atomic_store(&orow[sidx], key_c | ((uint32_t)new_l << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // start = i + start, finish = start + l + 3
                  SD_LDS_ORDER();
                  // The entry is read back by OTHER lanes (the owner in save_masked_regions, lane j' of a later find_perfect):
                  // the store must have reached L2, where their sc1 loads look, before any of them reads — they wait (vmcnt(0)) in
                  // front of their loads.
                  const unsigned long long insj = __brevll(sd_ballot(ins));           // bit j <-> window position j
                  inserted |= insj != 0;
                  if (insj && lane == ol) {
                      const int lowest = o_start + __builtin_ctzll(insj);
                      if (occ == 0 || lowest < minstart) minstart = lowest;
                      occ |= rotl64(insj, o_start & 63);
                      evict_k = next_evict(kk >= 3 ? 0u : 0xFFFFFFFFu << (8 * (kk + 1)));
                  }
              }
          }
          return inserted;
        };
        int kk0 = 0;
        if (!sd_any((int)((tw & 0xC0C0C0C0u) != 0u) | (int)(evict_k < k4 + 4) | (int)(p < CAPW - 1))) {
            kk0 = 4;
            if (STATS) ++st_plain;
            if (word_step(std::true_type{}, 0)) kk0 = 1;
            else if (word_step(std::true_type{}, 1)) kk0 = 2;
            else if (word_step(std::true_type{}, 2)) kk0 = 3;
            else (void)word_step(std::true_type{}, 3);
        }
#pragma clang loop unroll(disable)
        for (int kk = kk0; kk < 4; ++kk) (void)word_step(std::false_type{}, kk);
        if (grp_n && nmask) LN = k4 + ((31 - __builtin_clz(nmask)) >> 3);
      }
      if (more) {
          const uint4 *q = reinterpret_cast<const uint4 *>(A.bases + ((uint64_t)(blk64 + (uint32_t)(k64 >> 6) + 1u) << 6));
          const uint4 q2 = q[2], q3 = q[3];
          blk.s8 = q2.x; blk.s9 = q2.y; blk.sa = q2.z; blk.sb = q2.w; blk.sc = q3.x; blk.sd = q3.y; blk.se = q3.z; blk.sf = q3.w;
      }
#undef SD_HAS_LOW_BYTE
    }
    if (STATS && A.stats && lane == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)st_steps);
        atomicAdd(&A.stats[1], (unsigned long long)st_fp);
        atomicAdd(&A.stats[2], (unsigned long long)st_plain);
        const unsigned long long dt = wall_clock64() - st_t0;      // 100 MHz ticks this wave spent in the loop
        atomicAdd(&A.stats[3], dt);
        atomicMax(&A.stats[4], dt);
        atomicAdd(&A.stats[8], (unsigned long long)st_full);
        atomicAdd(&A.stats[9], (unsigned long long)st_iter);
        atomicAdd(&A.stats[10], (unsigned long long)st_qt);
        // histogram over the wave's time in the loop (0.5 ms bins): waves, their find_perfect calls with candidates, their end times
        const unsigned bin = (unsigned)(dt / 50000ull) < 31u ? (unsigned)(dt / 50000ull) : 31u;
        atomicAdd(&A.stats[14 + 4 * bin], 1ull);
        atomicAdd(&A.stats[15 + 4 * bin], (unsigned long long)st_full);
        atomicAdd(&A.stats[16 + 4 * bin], (unsigned long long)st_jobs);
        atomicAdd(&A.stats[17 + 4 * bin], (unsigned long long)st_fp);
    }
}

// bit 7 of every byte of the result: that byte of `word` is not one of A C G T a c g t
__device__ __forceinline__ uint32_t sd_not_acgt(uint32_t word)
{
    const uint32_t y = word & 0xDFDFDFDFu, idx = y & 0x07070707u;          // fold case; low 3 bits: A1 C3 T4 G7
    const uint32_t d = y ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, idx);
    return ((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d;
}

// ---- where a lane starts for a chunk: W-2 word emissions before (chunk start - 2W), found by scanning backwards (see
// the header comment); the common case — the W bases there are plain letters — is checked with 16-byte loads, the scan
// gives up after SD_SCAN_CAP bases of N-dense sequence and uses the word-count table, or asks the host for it (-1).
// LEAN (the walk of sd_sift, where this is a rare fallback of one wave): without the five 16-byte loads in flight — inlined there
// they were the kernel's highest register pressure and the one place that spilled
template <bool LEAN>
__device__ int sd_find_start(const SdArgs &A, const SdChunk ch, const uint8_t *seq)
{
    const int W = A.W, CAPW = W - 2;
    int u = 0;
    if (ch.start > 0) {
        const int y = ch.start - 2 * W;
        if (y > 2) {
            int need = CAPW, run = 0, p = y - 1;
            const int floor_p = y - SD_SCAN_CAP > 0 ? y - SD_SCAN_CAP : 0;
            // common case, checked with dword loads: the W bases before y are all letters A/C/G/T, so the W-2 words end
            // exactly there and the scan would stop at y - W
            bool plain = y - W > 0;
            if (plain) {
                uint32_t bad = 0;
                const int q0 = (y - W) & ~15;                                  // 16-byte loads, all issued before use
                if (!LEAN && y - q0 <= 80) {
                    uint4 v[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) v[i] = q0 + 16 * i < y ? *reinterpret_cast<const uint4 *>(seq + q0 + 16 * i) : make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
#pragma unroll
                    for (int i = 0; i < 5; ++i) bad |= sd_not_acgt(v[i].x) | sd_not_acgt(v[i].y) | sd_not_acgt(v[i].z) | sd_not_acgt(v[i].w);
                } else {
                    for (int q = q0; q < y; q += 4) bad |= sd_not_acgt(*reinterpret_cast<const uint32_t *>(seq + q));
                }
                plain = (bad & 0x80808080u) == 0;
            }
            if (plain) {
                p = y - W;
                need = 0;
            } else {
                for (; p >= floor_p; --p) {
                    if (nt4_code(seq[p]) < 4) {
                        if (++run >= 3 && --need == 0) break;
                    } else {
                        run = 0;
                    }
                }
            }
            if (need == 0 || floor_p == 0) {
                u = p > 0 ? p : 0;
            } else if (A.wtab == nullptr) {
                // N-dense stretch longer than the local scan: ask the host for the word-count table and a rerun
                atomicOr(A.need_wtab, 1u);
                u = -1;
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
    return u;
}

#include "sdust_sift.hpp"

// ---- scheduling hint per chunk, before the main kernel: the 64 bytes in the middle of the chunk; a sample whose 62
// 3-mers take few distinct values (random sequence: ~40 of 64) lies in a repeat array.  The flag only orders the work;
// results do not depend on it.
// Also resets what the main kernel expects cleared per chunk (claim flag, interval count) and the queue order array
// (`perm`, n_chunks + 160 positions of "nothing here"): one launch instead of three memsets in front of the scan.
// the 64 bytes in the middle of a chunk: few distinct 3-mers = inside a repeat array
__device__ __forceinline__ uint32_t sd_sample_heavy(const SdChunk ch, const uint8_t *seq)
{
    const int at = (ch.start + ((ch.end - ch.start) >> 1)) & ~63;      // blocks never leave the (64-byte padded) contig
    const uint4 *src = reinterpret_cast<const uint4 *>(seq + at);
    const uint4 x0 = src[0], x1 = src[1], x2 = src[2], x3 = src[3];
    const uint32_t w16[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
    uint32_t seen_lo = 0, seen_hi = 0;
    unsigned t = 0;
#pragma unroll
    for (int d = 0; d < 16; ++d)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            // letters only, case folded: (c >> 1) & 3 separates A C G T; anything else just lands on one of them
            t = ((t << 2) | ((w16[d] >> (8 * b + 1)) & 3u)) & 63u;
            const uint32_t bit = 1u << (t & 31u);
            seen_lo |= t < 32u ? bit : 0u;
            seen_hi |= t < 32u ? 0u : bit;
        }
    return __popc(seen_lo) + __popc(seen_hi) <= 20 ? 1u : 0u;
}

__global__ void sd_prep(SdArgs A, uint32_t *flag, uint32_t *perm)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < A.n_chunks + 160) perm[c] = 0xFFFFFFFFu;
    if (c >= A.n_chunks) return;
    A.claim[c] = 0;
    A.out_n[c] = 0;
    const SdChunk ch = A.chunks[c];
    flag[c] = sd_sample_heavy(ch, A.bases + A.ctg_off[ch.ctg]);
}

// the flags alone, as bytes (first call for an assembly: the host cuts the flagged chunks into shorter ones)
__global__ void sd_flags(const SdChunk *chunks, int32_t n_chunks, const uint8_t *bases, const int64_t *ctg_off, uint8_t *flag)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const SdChunk ch = chunks[c];
    flag[c] = (uint8_t)sd_sample_heavy(ch, bases + ctg_off[ch.ctg]);
}

// Queue order: flagged chunks first, but only one in every S = min(64, chunks / flagged) positions, the rest filled
// with the other chunks: a wave fetches 64 consecutive positions at a time, and 64 lanes inside repeat arrays in ONE
// wave would serialise (every find_perfect of a wave runs on all of its 64 lanes).  The other chunks come in P
// passes over the input — every P-th chunk, then the ones halfway between, ... (phases in bit-reversed order, `turn`) —
// so that a lane has free chunks ahead of it to run on into.  perm has n_chunks + P + 16 positions, 0xFFFFFFFF where
// nothing lands.
struct SdPasses {
    uint32_t P;
    uint8_t turn[64];                 // turn[phase] = which pass hands the chunks of that phase out
};
__global__ void sd_order(const uint32_t *flag, const uint32_t *rank, const unsigned long long *n_flagged, int32_t n_chunks, uint32_t *perm, SdPasses ps,
                         uint32_t *dense_list, uint32_t *claim, int32_t tail0)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    uint32_t H = (uint32_t)*n_flagged;
    const uint32_t r = rank[c];
    // the others, in input order: main part [0, Lm), then the part made of short chunks [Lm, L); each is handed out in P
    // passes of its own (every P-th chunk first ...), the short ones after all of the main part
    const uint32_t L = (uint32_t)n_chunks - H;
    const uint32_t Lm = tail0 < n_chunks ? (uint32_t)tail0 - rank[tail0] : L;
    const uint32_t j0 = (uint32_t)c - r;                     // rank among the others
    // (the short chunks in ONE pass, in input order: a run must not grow over several of them again)
    const uint32_t stripe_m = (Lm + ps.P - 1u) / ps.P;
    const uint32_t j = j0 < Lm ? (uint32_t)ps.turn[j0 % ps.P] * stripe_m + j0 / ps.P                       // < P * stripe_m <= Lm + P - 1
                               : ps.P * stripe_m + (j0 - Lm);
    if (dense_list) {
        // the flagged chunks go to sdust_dense: they get no queue position and count as taken (nobody runs on into them)
        if (flag[c]) {
            dense_list[r] = (uint32_t)c;
            claim[c] = 1u;
            return;
        }
        perm[j] = (uint32_t)c;
        return;
    }
    uint32_t S = H ? (uint32_t)n_chunks / H : 64u;
    S = S > 64u ? 64u : (S < 1u ? 1u : S);
    uint32_t pos;
    if (flag[c]) {
        pos = r * S;
    } else {
        if (S > 1u && j < H * (S - 1u)) pos = (j / (S - 1u)) * S + 1u + j % (S - 1u);
        else pos = H * S + (j - H * (S - 1u));
    }
    perm[pos] = (uint32_t)c;
}

// chunk rows (fixed capacity) -> one dense list in chunk order, tagged with the contig
__global__ void sdust_gather(const uint2 *in, const uint32_t *cnt, const uint32_t *dst_off, uint32_t cap, const SdChunk *chunks,
                             int32_t n_chunks, cornetto_ivl_t *dst, uint32_t dst_cap = 0xFFFFFFFFu)
{
    const int cid = blockIdx.x * blockDim.x + threadIdx.x;
    if (cid >= n_chunks) return;
    uint32_t n = cnt[cid];
    if (n == 0) return;
    if (n > cap) n = cap;                              // (a chunk with more rows than a row holds: the caller sees the overflow and runs again)
    const int32_t ctg = chunks[cid].ctg;
    const uint2 *src = in + (size_t)cid * cap;
    const uint32_t o = dst_off[cid];
    for (uint32_t i = 0; i < n && o + i < dst_cap; ++i) dst[o + i] = cornetto_ivl_t{ctg, (int32_t)src[i].x, (int32_t)src[i].y};
}

// one sample per 2048 bases of every contig (blockIdx.y = contig): how many lie inside a repeat array (sd_sample_heavy)
__global__ void sd_sample_count(const uint8_t *bases, const int64_t *ctg_off, const int32_t *ctg_len, unsigned long long *count)
{
    const int c = blockIdx.y;
    const int len = ctg_len[c];
    const uint8_t *seq = bases + ctg_off[c];
    unsigned n = 0;
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s * 2048 + 64 <= len; s += (long long)gridDim.x * blockDim.x)
        n += sd_sample_heavy(SdChunk{c, (int32_t)(s * 2048), (int32_t)std::min<long long>(len, s * 2048 + 2048)}, seq);
    const unsigned long long m = sd_ballot(n != 0);
    (void)m;
    for (int d = 32; d > 0; d >>= 1) n += __shfl_down(n, d);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(count, (unsigned long long)n);
}

// order[] of sd_sift: the chunks of the walk list first, then the others as they come (rank = how many of the list lie in front)
__global__ void sd_make_order(const uint32_t *walk, const uint32_t *iswalk, const uint32_t *rank, int32_t n_chunks, uint32_t *order)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_chunks) return;
    const uint32_t nw = walk[0];
    if ((uint32_t)k < nw) order[k] = walk[2 + k];
    if (!iswalk[k]) order[nw + (uint32_t)k - rank[k]] = (uint32_t)k;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// what is left of a call whose work cornetto_sdust_asm_begin() queued in one go: wait, look at the counts, hand the result over.
// 1: the estimates did not hold (the caller runs the call the long way); 0: done; < 0: error
int sd_one_go_finish(cornetto_accel_t *h, cornetto_asm_t *a, cornetto_ivl_t *of, unsigned long long *p_tot, size_t cap, size_t n_cap, size_t m_cap, int64_t key,
                     bool walk_pending, cornetto_ivl_t **ivls, int64_t *n_ivls)
{
    if (hipStreamSynchronize(h->stream) != hipSuccess) {
        cornetto_free(of);
        return cn_fail(h, CORNETTO_E_HIP, "sdust: stitch / copy back failed");
    }
    const uint32_t ovf_f = (uint32_t)(p_tot[1] & 0xFFFFFFFFull);
    const bool wtab_f = (p_tot[1] >> 32) != 0;
    const unsigned long long rows_f = p_tot[0], out_f = rows_f ? p_tot[9] : 0ull;
    if (!wtab_f && ovf_f <= cap && rows_f <= n_cap && out_f != ~0ull && out_f <= m_cap && out_f <= rows_f && (rows_f == 0 || out_f > 0)) {
        if (walk_pending) a->sd_walk_key = key;
        a->sd_est_rows = (int64_t)rows_f;
        a->sd_est_out = (int64_t)out_f;
        cn_timing_end(h);
        *ivls = of;
        *n_ivls = (int64_t)out_f;
        return 0;
    }
    cornetto_free(of);                                         // (the estimate did not hold: the long way)
    a->sd_est_key = -1;
    return 1;
}

// phase 0: the whole call.  phase 1 (cornetto_sdust_asm_begin): queue it in one go if the last call for this table left its counts behind and
// return without waiting (h->sd_pend.state = 1); nothing queued if not (state 0); or, where the one-go form could not be set up after the main
// kernel was launched, the whole call (state 2, the result kept in h->sd_pend)
int sdust_asm_impl(cornetto_accel_t *h, const cornetto_asm_t *a_in, int32_t T, int32_t W, cornetto_ivl_t **ivls, int64_t *n_ivls, const int phase)
{
    // CORNETTO_SDUST_TRACE=1: host-side time stamps of the call's phases on stderr (development aid)
    static const bool trace = CN_DEV_INT("CORNETTO_SDUST_TRACE", 0) != 0;
    struct timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto stamp = [&](const char *what) {
        if (!trace) return;
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        fprintf(stderr, "[sdust trace] %-22s %8.3f ms\n", what, (t.tv_sec - ts0.tv_sec) * 1e3 + (t.tv_nsec - ts0.tv_nsec) * 1e-6);
    };
    if (!h || !a_in || !ivls || !n_ivls) return cn_fail(h, CORNETTO_E_ARG, "sdust: bad argument");
    cornetto_asm_t *a = const_cast<cornetto_asm_t *>(a_in);   // only the cached chunk table is touched
    *ivls = nullptr;
    *n_ivls = 0;
    // (W - 2 words in the window.  Up to 64: sdust_w64; up to 255: the older kernel with byte counters in LDS; up to 1024: the
    // same with its state in global memory.  Beyond, the reference's own 32-bit products r * l (:115,:118) overflow.)
    if (W < 3 || W > 1026) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: -w %d outside 3..1026 (the reference crashes below 3)", W);
    if (T < 0 || T > (1 << 20)) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: -t %d outside 0..2^20", T);
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);

    // waves the chip holds at once (LDS-bound: 19 per CU), once per handle
    const int variant0 = CN_DEV_INT("CORNETTO_SDUST_VARIANT", 0);
    const bool w64_path = W - 2 <= 64 && T >= 5 && T <= 100000 && variant0 == 0;
    if (w64_path && h->sd_slots == 0) {
        // The kernel must not spill: round-1 builds of it that kept registers in scratch memory gave results that changed from
        // run to run on MI355X (ROCm 7.2), builds without scratch never did (20 bench steps over 3.16 Gbp digest identically:
        // bench.py "determinism").  The cause could not be isolated since: hipcc 7.2 no longer produces a spilling build of this
        // kernel even when asked for 8 waves per SIMD or 48 registers ("failed to meet occupancy target": it keeps 89 VGPRs), so
        // there is nothing to test.  What was tightened meanwhile: the P-slot hand-off between lanes waits for the wave's
        // outstanding stores in front of EVERY slot load.  The guard covers both instantiations that can be launched;
        // CORNETTO_SDUST_ALLOW_SCRATCH=1 lifts it.
        for (int inst = 0; inst < 4; ++inst) {           // (sd_sift: the production builds; its counting build may keep a register in scratch)
            hipFuncAttributes fa;
            const void *fn = inst == 3 ? reinterpret_cast<const void *>(&sd_sift<false, SIFT_CAP_DEFAULT>) : inst == 2 ? reinterpret_cast<const void *>(&sd_sift<false>)
                             : inst ? reinterpret_cast<const void *>(&sdust_w64<true>) : reinterpret_cast<const void *>(&sdust_w64<false>);
            if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return cn_fail(h, CORNETTO_E_HIP, "sdust: hipFuncGetAttributes failed");
            if (fa.localSizeBytes != 0 && !CN_DEV_INT("CORNETTO_SDUST_ALLOW_SCRATCH", 0))
                return cn_fail(h, CORNETTO_E_HIP, "sdust: kernel%s built with %zu bytes of scratch per lane (register spills): refusing to run it",
                               inst == 1 ? " (statistics build)" : inst >= 2 ? " sd_sift" : "", (size_t)fa.localSizeBytes);
        }
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sdust_w64<false>, 64 * SD_WPB, 0) != hipSuccess || per_cu < 1) per_cu = 16;
        {
            // MI355X hands LDS out in 1280-byte granules (tools/ubench/lds_occupancy: 8192..8960 bytes -> 18 workgroups per CU,
            // 9216..10240 -> 16), which the occupancy query does not know: the waves beyond that would start when the others end
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&sdust_w64<false>)) == hipSuccess && fa.sharedSizeBytes > 0)
                per_cu = std::min<int>(per_cu, (int)(163840 / ((fa.sharedSizeBytes + 1279) / 1280 * 1280)));
        }
        per_cu *= SD_WPB;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus < 1) cus = 256;
        h->sd_slots = per_cu * cus;
        h->sd_cus = cus;
    }
    stamp("kernel attributes");
    const int64_t sd_waves = w64_path ? CN_DEV_INT("CORNETTO_SDUST_WAVES", std::max(1, h->sd_slots / h->sd_cus * h->share / 100) * h->sd_cus) : 0;

    // chunk = bases per lane.  Small enough that a long low-complexity array is shared by many waves, large
    // enough that the ~3W-base speculative warm-up stays a few percent.  CORNETTO_SDUST_CHUNK overrides
    // (tests use tiny chunks to stress the speculative start).
    int64_t chunk = CN_DEV_INT("CORNETTO_SDUST_CHUNK", 0);
    if (chunk <= 0) {
        // about 1536, adjusted so that the chunks come out as a whole number of rounds over the resident lanes: every
        // lane works through its chunks at the same pace, and 6.6 chunks per lane cost as much time as 7
        chunk = 1536;
        const int64_t lanes = sd_waves * 64;
        if (lanes > 0 && a->total / lanes >= 1536) {
            const int64_t rounds = a->total / (lanes * 1536);                      // >= 1
            chunk = std::min<int64_t>(3072, ((a->total + lanes * rounds - 1) / (lanes * rounds) + 63) / 64 * 64);
        }
    }
    chunk = std::max<int64_t>(16, chunk);
    if (W > 257 && CN_DEV_INT("CORNETTO_SDUST_CHUNK", 0) <= 0) chunk = std::max<int64_t>(chunk, 32 * (int64_t)W);   // (warm-up: 3 W bases per chunk)
    // The sift / resolve stages (sdust_sift.hpp) take every chunk of plain letters; they want chunks of whole 64-base tiles,
    // at least 256 bases (the look-back of a chunk stays inside the chunk before it), at most 62 tiles.  CORNETTO_SDUST_SIFT=0 keeps the
    // per-lane recurrence of sdust_w64 for everything (also what an explicit chunk size outside that range does: the tests
    // with tiny chunks stress exactly that kernel).
    // The sift / resolve stages are the default (3.16 Gbp uniform: 6.4 against 7.0-7.3 ms alone, 9.7 against 10.0 ms per two-stream bench
    // step; with 3 % of the bases in satellite arrays 16.3 against 29.5 ms; a 395 Mb share 0.96 against 4.7 ms).  CORNETTO_SDUST_SIFT=0
    // keeps the per-lane recurrence; -1 decides once per resident assembly from one 64-byte sample per 2048 bases (sift when at least 1 in
    // 256 lies inside a repeat array, for read-level sets and for assemblies below 2 Gbases) — how the choice was made while the
    // per-lane kernel was still ahead on uniform sequence.
    const int sift_env = CN_DEV_INT("CORNETTO_SDUST_SIFT", 1);      // 1 (default) sift / resolve, 0 the per-lane recurrence, -1 decide by the sample
    if (w64_path && sift_env < 0 && a->sd_auto < 0) {
        // (below ~2 Gbases the resident lanes of sdust_w64 have less than one chunk each and its time stops falling — one chunk is 4 ms
        // of sequential steps for a lane: 395 Mb take 4.7 ms against 1.0 ms in sd_sift, whose unit of work is a wave)
        if (a->n > 4096 || a->total < 2000000000ll) {
            a->sd_auto = 1;
        } else {
            unsigned long long *d_cnt8 = (unsigned long long *)cn_ws(h, WS_SD_STATS, 2048 + 64 * 64);
            unsigned long long *p_cnt8 = (unsigned long long *)cn_pin(h, PIN_SMALL, 2048);
            if (!d_cnt8 || !p_cnt8) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation failed");
            CN_HIP(h, hipMemsetAsync(d_cnt8, 0, 8, h->stream));
            int32_t maxlen = 0;
            for (int32_t c = 0; c < a->n; ++c) maxlen = std::max(maxlen, a->len[c]);
            const unsigned bx = (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, (maxlen / 2048 + 255) / 256));
            sd_sample_count<<<dim3(bx, (unsigned)a->n), dim3(256), 0, h->stream>>>(a->d_bases, a->d_off, a->d_len, d_cnt8);
            CN_HIP(h, hipGetLastError());
            CN_HIP(h, hipMemcpyAsync(p_cnt8 + 220, d_cnt8, 8, hipMemcpyDeviceToHost, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));
            const int64_t samples = a->total / 2048, heavy = (int64_t)p_cnt8[220];
            a->sd_auto = heavy * 256 >= std::max<int64_t>(samples, 1) && heavy >= 64 ? 1 : 0;
            if (trace) fprintf(stderr, "[sdust trace] %lld of %lld samples inside repeat arrays: %s\n", (long long)heavy, (long long)samples, a->sd_auto ? "sift / resolve" : "per-lane recurrence");
        }
    }
    bool sift_on = w64_path && (sift_env > 0 || (sift_env < 0 && a->sd_auto == 1));
    // (its own default, SIFT_CHUNK_DEFAULT in sdust_sift.hpp: 1792 bases since round 5.  Rounds 3-4: 1536 bases = 26 tiles with the two in front; 7.2 KB of LDS per wave = six 1280-byte granules, 21 waves per
    // CU; 1792 needs a seventh granule: measured 6.85 against 7.2 ms on the 3.16 Gbp assembly)
    if (sift_on && CN_DEV_INT("CORNETTO_SDUST_CHUNK", 0) <= 0) chunk = SIFT_CHUNK_DEFAULT;
    if (sift_on && (chunk % 64 != 0 || chunk < 256 || chunk > 3968)) sift_on = false;     // (64 tiles with the two in front)
    // Optionally the last part of the work is cut into shorter chunks, handed out last and in one pass: when the queue runs
    // dry every wave still has to finish the chunks its lanes hold, and a wave-step costs the same with 3 busy lanes as with
    // 64.  Measured on the 3.16 Gbp assembly (tools/perf_probe.py, 5 launches each): 0 % 8.5-9.1 ms, 10 % / 4x 8.5-8.8,
    // 20 % / 4x 8.7-9.1, 20 % / 2x 8.5-9.0, 30 % / 4x 8.9-9.2 — the extra warm-ups cost what the shorter drain saves, so it is
    // OFF by default.  CORNETTO_SDUST_TAIL = percent of the bases, CORNETTO_SDUST_TAILDIV = how many times shorter (4).
    // Results do not depend on the decomposition.
    const int tail_pct = sift_on ? 0 : std::min(90, std::max(0, CN_DEV_INT("CORNETTO_SDUST_TAIL", 0)));
    const int tail_div = std::min(16, std::max(1, CN_DEV_INT("CORNETTO_SDUST_TAILDIV", 4)));
    const int64_t small = std::max<int64_t>(64, (chunk / tail_div + 63) / 64 * 64);
    const int64_t tail_from = tail_pct > 0 && tail_div > 1 ? a->total - a->total * tail_pct / 100 : a->total + 1;   // in bases, assembly order
    const int64_t key = chunk + (int64_t)tail_pct * (1ll << 40) + (int64_t)tail_div * (1ll << 48) + (sift_on ? 1ll << 56 : 0);
    if (a->sd_chunk != key && tail_pct == 0) {
        // the table on the device (tiles.hpp): chunk j of contig c = [j chunk, min(len, (j + 1) chunk))
        const int64_t nch = cntiles::prefix(h, a->sd_pref, a->n, [&](int32_t c) { return ((int64_t)a->len[c] + chunk - 1) / chunk; });
        if (nch < 0) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: device allocation failed");
        if (nch > 0x7fffffffll) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: too many chunks");
        if (a->d_sd_chunks) { (void)hipFree(a->d_sd_chunks); a->d_sd_chunks = nullptr; }
        if (nch > 0) {
            if (cn_obj_malloc(h, &a->d_sd_chunks, (size_t)nch * sizeof(SdChunk)) != hipSuccess)
                return cn_fail(h, CORNETTO_E_NOMEM, "sdust: device allocation failed");
            cntiles::fill<<<dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, h->stream>>>(a->sd_pref.dev, a->n, nch,
                                                                                           SdChunkFill{reinterpret_cast<SdChunk *>(a->d_sd_chunks), a->d_len, (int32_t)chunk});
            CN_HIP(h, hipGetLastError());
        }
        a->sd_tail0 = nch;
        a->sd_chunk = key;
        a->sd_n_chunks = nch;
        a->sd_flagged = -1;
        a->sd_refined = false;
        a->sd_plan_key = -1;
    }
    if (a->sd_chunk != key) {
        std::vector<SdChunk> chunks;
        int64_t seen = 0;
        a->sd_tail0 = -1;
        for (int32_t c = 0; c < a->n; ++c) {
            for (int64_t s = 0; s < a->len[c];) {
                const bool tail = seen + s >= tail_from;
                if (tail && a->sd_tail0 < 0) a->sd_tail0 = (int64_t)chunks.size();
                const int64_t sz = tail ? small : chunk;
                chunks.push_back(SdChunk{c, (int32_t)s, (int32_t)std::min<int64_t>(a->len[c], s + sz)});
                s += sz;
            }
            seen += a->len[c];
        }
        if (a->sd_tail0 < 0) a->sd_tail0 = (int64_t)chunks.size();
        if (chunks.size() > 0x7fffffffull) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: too many chunks");
        if (a->d_sd_chunks) { (void)hipFree(a->d_sd_chunks); a->d_sd_chunks = nullptr; }
        if (!chunks.empty()) {
            if (hipMalloc(&a->d_sd_chunks, chunks.size() * sizeof(SdChunk)) != hipSuccess)
                return cn_fail(h, CORNETTO_E_NOMEM, "sdust: device allocation failed");
            CN_HIP(h, hipMemcpyAsync(a->d_sd_chunks, chunks.data(), chunks.size() * sizeof(SdChunk), hipMemcpyHostToDevice, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));
        }
        a->sd_chunk = key;
        a->sd_n_chunks = (int64_t)chunks.size();
        a->sd_flagged = -1;
        a->sd_refined = false;
        a->sd_plan_key = -1;
    }
    // Optionally (CORNETTO_SDUST_DENSE_SPLIT = n > 1; default 1 = off) the first call for a table cuts the chunks sampled as
    // low-complexity into n shorter ones when they will go to the dense kernel: one job of that kernel is one chunk for one
    // lane, and a wave that is alone on its SIMD issues one instruction every ~6 cycles — ~10 us per step, 20 ms per 1.7 kb job.
    // Measured on the satellite profile (3.16 Gbp, 1 056 dense jobs): n = 1 dense 20 ms beside a 29 ms main kernel, n = 2
    // 28 / 31 ms, n = 4 27-32 / 34 ms, n = 8 35-40 / 38 ms: at 28 KB of LDS per dense wave only five fit on a CU, so more jobs
    // are more rounds (plus their warm-ups), not more waves per SIMD.  The split pays only once the dense state is smaller
    // (16-bit P slots, 12-bit ring entries: ~18 KB) — DESIGN.md section 8.
    if (w64_path && !sift_on && !a->sd_refined && a->sd_n_chunks > 0 && CN_DEV_INT("CORNETTO_SDUST_ORDER", 1)) {
        a->sd_refined = true;
        const int dense_mode = CN_DEV_INT("CORNETTO_SDUST_DENSE", 1);
        const int split = std::min(16, std::max(1, CN_DEV_INT("CORNETTO_SDUST_DENSE_SPLIT", 1)));
        const size_t n0 = (size_t)a->sd_n_chunks;
        if (dense_mode && split > 1) {
            uint8_t *d_f = (uint8_t *)cn_ws(h, WS_SD_PERM, n0 + 64);
            std::vector<uint8_t> f(n0);
            if (!d_f) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation failed");
            sd_flags<<<dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, h->stream>>>(reinterpret_cast<const SdChunk *>(a->d_sd_chunks), (int32_t)n0, a->d_bases, a->d_off, d_f);
            CN_HIP(h, hipGetLastError());
            CN_HIP(h, hipMemcpyAsync(f.data(), d_f, n0, hipMemcpyDeviceToHost, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));
            size_t nf = 0;
            for (size_t i = 0; i < n0; ++i) nf += f[i];
            if (nf > 0 && (dense_mode == 2 || nf >= std::max<size_t>(1024, n0 / 256))) {
                std::vector<SdChunk> old(n0), chunks;
                CN_HIP(h, hipMemcpy(old.data(), a->d_sd_chunks, n0 * sizeof(SdChunk), hipMemcpyDeviceToHost));
                chunks.reserve(n0 + nf * (size_t)(split - 1));
                int64_t tail0 = -1;
                for (size_t i = 0; i < n0; ++i) {
                    if ((int64_t)i == a->sd_tail0) tail0 = (int64_t)chunks.size();
                    const int32_t len_i = old[i].end - old[i].start;
                    const int32_t piece = std::max(64, ((len_i + split - 1) / split + 63) / 64 * 64);
                    if (!f[i] || len_i <= piece) { chunks.push_back(old[i]); continue; }
                    for (int32_t st = old[i].start; st < old[i].end; st += piece) chunks.push_back(SdChunk{old[i].ctg, st, std::min(old[i].end, st + piece)});
                }
                if (chunks.size() > 0x7fffffffull) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: too many chunks");
                (void)hipFree(a->d_sd_chunks);
                a->d_sd_chunks = nullptr;
                if (hipMalloc(&a->d_sd_chunks, chunks.size() * sizeof(SdChunk)) != hipSuccess) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: device allocation failed");
                CN_HIP(h, hipMemcpy(a->d_sd_chunks, chunks.data(), chunks.size() * sizeof(SdChunk), hipMemcpyHostToDevice));
                a->sd_tail0 = tail0 < 0 ? (int64_t)chunks.size() : tail0;
                a->sd_n_chunks = (int64_t)chunks.size();
                a->sd_plan_key = -1;
            }
        }
    }
    const size_t nc = (size_t)a->sd_n_chunks;
    cornetto_ivl_t *o = nullptr;
    int64_t n_out = 0;
    stamp("chunk table");
    if (nc > 0) {
        const SdChunk *d_chunks = reinterpret_cast<const SdChunk *>(a->d_sd_chunks);
        // per chunk: count (4 B) + ordered offset (4 B) + scan partials; then {total u64, ovf u32}
        uint32_t *d_cnt = (uint32_t *)cn_ws(h, WS_SD_CNT, nc * 8 + ((nc + 4095) / 4096 + 1) * 4);
        unsigned long long *d_tot = (unsigned long long *)cn_ws(h, WS_SD_STATS, 2048 + 64 * 64);   // (+ the chunk counters of sd_sift)   // [0] total [1] overflow | table request [2..6] stats [7] flagged [8] queue
        unsigned long long *p_tot = (unsigned long long *)cn_pin(h, PIN_SMALL, 2048);
        if (!d_cnt || !d_tot || !p_tot) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation failed");
        uint32_t *d_off = d_cnt + nc, *d_part = d_off + nc;
        stamp("small workspaces");
        const bool env_stats = CN_DEV_INT("CORNETTO_SDUST_STATS", 0) != 0;
        const bool want_stats = env_stats || h->sd_stats != 0;
        size_t cap = (size_t)std::max<int64_t>(16, chunk / 32);
        cap = (size_t)CN_DEV_INT("CORNETTO_SDUST_CAP", (int)cap);
        if (h->dev[WS_SD_OUT].bytes / (nc * sizeof(uint2)) > cap) cap = h->dev[WS_SD_OUT].bytes / (nc * sizeof(uint2));
        unsigned long long tot = 0;
        uint2 *d_out = nullptr;
        // the rest of the call in one go (below) needs the counts the last call for this table left behind: ONE predicate for the early-out of
        // cornetto_sdust_asm_begin() here and for the one-go form itself
        const int64_t est_key = key * 131 + T * 1031 + W;
        const bool one_go = w64_path && sift_on && !want_stats && a->sd_est_key == est_key && a->sd_est_rows >= 0 && CN_DEV_INT("CORNETTO_SDUST_FUSED", 1);
        if (phase == 1 && !one_go)
            return CORNETTO_OK;                        // (nothing to size the rest of the call by: cornetto_sdust_asm_end runs it)
        for (int attempt = 0; attempt < 4; ++attempt) {
            d_out = (uint2 *)cn_ws(h, WS_SD_OUT, nc * cap * sizeof(uint2));
            if (!d_out) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation of %zu bytes failed", nc * cap * sizeof(uint2));
            CN_HIP(h, hipMemsetAsync(d_tot, 0, 2048 + 64 * 64, h->stream));
            // P slot rows: one per resident lane (sdust_w64) / unused by the older kernels
            uint32_t *d_slots = (uint32_t *)cn_ws(h, WS_SD_OFF, (size_t)(std::max<int64_t>(sd_waves, 1) + SD_WPB) * 64 * 64 * sizeof(uint32_t));
            if (!d_slots) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation failed");
            stamp("row + slot workspaces");
            const bool use_w64 = w64_path;                              // CORNETTO_SDUST_VARIANT=1 forces the per-lane reference-shaped kernel
            SdArgs A{a->d_bases, a->d_off, a->d_len, d_chunks, (int32_t)nc, T, W, d_out, d_cnt, (uint32_t)cap,
                     want_stats ? d_tot + 2 : nullptr, reinterpret_cast<uint32_t *>(d_tot + 1), d_slots, nullptr,
                     reinterpret_cast<uint32_t *>(d_tot + 8), nullptr, 0, CN_DEV_INT("CORNETTO_SDUST_RUNON", 1), (int32_t)chunk, a->d_wtab, a->d_wtab_base, reinterpret_cast<uint32_t *>(d_tot + 1) + 1};
            unsigned nb = (unsigned)((nc + 63) / 64);
            bool sift_walk_pending = false;
            bool dense_pending = false;
            // (whatever path leaves this attempt — a failed launch, a failed allocation — the dense kernel on the second stream is
            // through before the workspaces it writes can be handed to anybody else)
            struct DenseJoin {
                cornetto_accel_t *h;
                bool *pending;
                ~DenseJoin() { if (*pending && h->stream2) (void)hipStreamSynchronize(h->stream2); }
            } dense_join{h, &dense_pending};
            if (use_w64 && sift_on) {
                // ---- one launch: sift + resolve over the chunks of plain letters, the base-by-base walk over the others ----
                CN_HIP(h, hipMemsetAsync(d_cnt, 0, nc * 4, h->stream));       // (a wave that asks for the word-count table publishes nothing)
                const uint32_t reg_cap = (uint32_t)(chunk + 128);
                const uint32_t lds_wave = sift_lds_bytes(reg_cap);
                int lmin = 1;
                while (5 * (lmin + 1) <= T && lmin < 16) ++lmin;
                // the chunks that are walked base by base, kept with the assembly's chunk table (see SiftArgs):
                // [0] count [1] pad | list [nc] | flags [nc] | rank [nc] | order [nc] | scan partials
                const bool walk_known = a->d_sd_walk && a->sd_walk_key == key;
                const size_t walk_words = 2 + 4 * nc + ((nc + 4095) / 4096 + 2);
                if (!walk_known) {
                    if (a->d_sd_walk) { (void)hipFree(a->d_sd_walk); a->d_sd_walk = nullptr; }
                    a->sd_walk_key = -1;
                    if (cn_obj_malloc(h, (void **)&a->d_sd_walk, walk_words * 4) != hipSuccess) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: device allocation failed");
                    CN_HIP(h, hipMemsetAsync(a->d_sd_walk, 0, (2 + 2 * nc) * 4, h->stream));
                }
                uint32_t *d_wflag = a->d_sd_walk + 2 + nc, *d_wrank = d_wflag + nc, *d_worder = d_wrank + nc, *d_wpart = d_worder + nc;
                SiftArgs S{a->d_bases, a->d_off, a->d_len, d_chunks, (int32_t)nc, T, W, lds_wave, reg_cap,
                           walk_known ? d_worder : nullptr, walk_known ? nullptr : a->d_sd_walk, walk_known ? nullptr : d_wflag,
                           reinterpret_cast<uint32_t *>(d_tot + 256), T / 10 + 1, lmin, CN_DEV_INT("CORNETTO_SIFT_ABL", 0),
                           std::min(65, std::max(1, CN_DEV_INT("CORNETTO_SIFT_DP", 24))), std::min(65, std::max(1, CN_DEV_INT("CORNETTO_SIFT_L2SKIP", 48)))};
                sift_walk_pending = !walk_known;
                SdArgs R = A;
                R.stats = want_stats ? d_tot + 200 : nullptr;
                // Resident waves: as many workgroups as the chip holds at once (LDS is handed out in 1280-byte granules), a share of them
                // when another stream computes beside this one (cornetto_accel_set_share): a launch of one workgroup per chunk keeps
                // the other stream's kernels waiting until it is through (13.1 instead of 9 ms per bench step).  The waves take their
                // chunks from one counter (static strides were measured 10-50 % slower: chunks differ a lot in cost; 64 counters: one was a 14 ns serial point).
                const bool cap_default = reg_cap == SIFT_CAP_DEFAULT && !CN_DEV_INT("CORNETTO_SIFT_GENERIC", 0) && !CN_DEV_INT("CORNETTO_SIFT_ABL", 0);     // (the build with the buffer size as a literal)
                if (h->sift_per_cu == 0 || h->sift_per_cu_default != (cap_default ? 1 : 0)) {
                    int per_cu = 0;
                    const hipError_t oe = cap_default ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sd_sift<false, SIFT_CAP_DEFAULT>, 64 * SIFT_WPB, lds_wave * SIFT_WPB)
                                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sd_sift<false>, 64 * SIFT_WPB, lds_wave * SIFT_WPB);
                    if (oe != hipSuccess || per_cu < 1) per_cu = 8;
                    h->sift_per_cu = per_cu;
                    h->sift_per_cu_default = cap_default ? 1 : 0;
                }
                const int per_cu_all = std::max(1, std::min<int>(h->sift_per_cu, (int)(163840 / ((lds_wave * SIFT_WPB + 1279) / 1280 * 1280))));
                const bool free_now = __atomic_load_n(&h->boost, __ATOMIC_ACQUIRE) != 0;        // (cornetto_accel_boost)
                // (the share of ALL the chip's slots, not of a CU's rounded down: 25 slots per CU make whole waves per CU steps of 4 %.  Measured with it: the
                // other stream's coverage kernel does not degrade gradually between 19 and 20 resident waves per CU — 3.10 ms at 76 %, 3.29 at 77 %
                // (a quarter wave per CU more), 3.55 at 78-80 % — so the balance of a bench step stays at 76 %)
                const size_t cus_n = (size_t)std::max(h->sd_cus, 1);
                const size_t slots_all = (size_t)per_cu_all * cus_n;
                const size_t slots = free_now ? slots_all : std::max<size_t>(cus_n, slots_all * (size_t)h->share / 100);
                const size_t all_blocks = (nc + SIFT_WPB - 1) / SIFT_WPB;
                const unsigned nbk = (unsigned)std::min<size_t>(all_blocks, (size_t)CN_DEV_INT("CORNETTO_SIFT_BLOCKS", (int)slots));
                // The waves left to the other stream: when its owner says it is through (cornetto_accel_boost, from another host thread) while
                // this kernel still runs, they are launched as a second kernel on a second stream — same arguments, same chunk counters: the two
                // launches drain them together — and the stream of this call waits for both.  The helper must not start before the counters,
                // the counts and the walk list of THIS call are reset: it waits for an event recorded behind those memsets, in front of the main launch.
                const unsigned extra = (unsigned)std::min<size_t>(all_blocks > nbk ? all_blocks - nbk : 0, slots_all > slots ? slots_all - slots : 0);
                // (from 85 % on the waves left out are too few to matter — measured: 7.57 ms per step with and without them at 92 % — and without
                // the poll below the rest of the call is queued behind the kernel while it runs.  Round 5: at the shares the probe picks for a
                // balanced step, 72-80 %, the other thread ends a few hundred microseconds before this kernel and the helper buys less than the
                // call's tail loses by being queued only after the poll: 6.66 against 6.8-7.0 ms per step at 76 %, tools/ab_help.sh — so the
                // helper is for shares below 70 % only)
                const bool may_help = phase == 0 && extra > 0 && !want_stats && h->share < CN_DEV_INT("CORNETTO_SDUST_HELP_BELOW", 70);      // (its poll waits for the kernel: not in _begin)
                if (may_help) {
                    if (!h->stream2) {
                        int pr_least = 0, pr_greatest = 0;
                        (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
                        CN_HIP(h, hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, pr_greatest));
                        CN_HIP(h, hipEventCreateWithFlags(&h->ev2, hipEventDisableTiming));
                        CN_HIP(h, hipEventCreateWithFlags(&h->ev1, hipEventDisableTiming));
                    }
                    if (!h->ev3) CN_HIP(h, hipEventCreateWithFlags(&h->ev3, hipEventDisableTiming));
                    CN_HIP(h, hipEventRecord(h->ev1, h->stream));               // the state of this call is set up
                }
                stamp("set up");
                if (want_stats) CN_LAUNCH(h, "sdust_kernel", sd_sift<true><<<dim3(nbk), dim3(64 * SIFT_WPB), lds_wave * SIFT_WPB, h->stream>>>(S, R));
                else if (cap_default) CN_LAUNCH(h, "sdust_kernel", sd_sift<false, SIFT_CAP_DEFAULT><<<dim3(nbk), dim3(64 * SIFT_WPB), lds_wave * SIFT_WPB, h->stream>>>(S, R));
                else CN_LAUNCH(h, "sdust_kernel", sd_sift<false><<<dim3(nbk), dim3(64 * SIFT_WPB), lds_wave * SIFT_WPB, h->stream>>>(S, R));
                __atomic_fetch_add(&h->launch_seq, 1ull, __ATOMIC_RELEASE);      // (cornetto_accel_launch_count)
                if (may_help) {
                    CN_HIP(h, hipEventRecord(h->ev3, h->stream));
                    bool helped = false;
                    while (hipEventQuery(h->ev3) == hipErrorNotReady) {
                        if (!helped && __atomic_load_n(&h->boost, __ATOMIC_ACQUIRE) != 0) {
                            CN_HIP(h, hipStreamWaitEvent(h->stream2, h->ev1, 0));
                            if (cap_default) sd_sift<false, SIFT_CAP_DEFAULT><<<dim3(extra), dim3(64 * SIFT_WPB), lds_wave * SIFT_WPB, h->stream2>>>(S, R);
                            else sd_sift<false><<<dim3(extra), dim3(64 * SIFT_WPB), lds_wave * SIFT_WPB, h->stream2>>>(S, R);
                            dense_pending = true;                                // (every exit from here on joins the second stream: DenseJoin)
                            CN_HIP(h, hipGetLastError());
                            CN_HIP(h, hipEventRecord(h->ev2, h->stream2));
                            helped = true;
                        }
                        sched_yield();
                    }
                    if (helped) CN_HIP(h, hipStreamWaitEvent(h->stream, h->ev2, 0));
                    stamp(helped ? "kernel done (helped)" : "kernel done (polled)");
                }
                if (sift_walk_pending) {
                    CN_TRY(cnscan::exclusive_u32(h, "sdust_order", d_wflag, (int64_t)nc, 1, d_wrank, d_wpart, nullptr));
                    CN_LAUNCH(h, "sdust_order", sd_make_order<<<dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, h->stream>>>(a->d_sd_walk, d_wflag, d_wrank, (int32_t)nc, d_worder));
                }
            } else if (use_w64) {
                // warm-up starts, the order of the queue, the claim flags:
                // flag (nc) + rank (nc) + claim (nc) + perm (nc + 80) + dense list (nc) + scan partials
                uint32_t *d_flag = (uint32_t *)cn_ws(h, WS_SD_PERM, (nc * 5 + 160) * 4 + ((nc + 4095) / 4096 + 1) * 4);
                if (!d_flag) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation failed");
                uint32_t *d_rank = d_flag + nc, *d_claim = d_rank + nc, *d_perm = d_claim + nc, *d_list = d_perm + nc + 160, *d_pp = d_list + nc;
                const unsigned nbs = (unsigned)((nc + 255) / 256);
                A.claim = d_claim;
                A.q_len = (int32_t)nc;
                if (!CN_DEV_INT("CORNETTO_SDUST_ORDER", 1)) {
                    CN_HIP(h, hipMemsetAsync(d_claim, 0, nc * 4, h->stream));
                    CN_HIP(h, hipMemsetAsync(d_cnt, 0, nc * 4, h->stream));      // chunks a lane runs on into publish nothing of their own
                } else {
                    // CORNETTO_SDUST_DENSE: 1 (default) the chunks inside repeat arrays go to sdust_dense when there are enough of
                    // them to pay for it, 2 always (tests), 0 never (they stay in the main kernel's queue, first, one per wave)
                    const int dense_mode = CN_DEV_INT("CORNETTO_SDUST_DENSE", 1);
                    SdPasses ps;
                    ps.P = (uint32_t)std::min(64, std::max(1, CN_DEV_INT("CORNETTO_SDUST_PASSES", 8)));
                    // The plan of a call — which chunks are sampled as low-complexity, the order of the queue, the list for the
                    // dense kernel, the initial claim flags — depends on the resident assembly and the chunk table only: it is
                    // built by the first call and kept with the assembly (like the chunk table itself); later calls copy the
                    // claim flags back and clear the counts, two small asynchronous operations instead of five launches that
                    // queue behind whatever else the device is running.
                    const int64_t plan_key = key * 4 + (dense_mode & 3) + ((int64_t)ps.P << 56);
                    unsigned long long n_dense = 0;
                    if (a->sd_plan_key != plan_key || !a->d_sd_plan) {
                        const unsigned nbp = (unsigned)((nc + 160 + 255) / 256);
                        CN_LAUNCH(h, "sdust_prep", sd_prep<<<dim3(nbp), dim3(256), 0, h->stream>>>(A, d_flag, d_perm));
                        // passes of the queue over the input: a run can grow to `passes` chunks before it meets a queue start
                        {
                            int bits = 0;
                            while ((1u << bits) < ps.P) ++bits;
                            std::vector<std::pair<uint32_t, uint32_t>> key2;      // (bit-reversed phase, phase)
                            for (uint32_t ph = 0; ph < ps.P; ++ph) {
                                uint32_t rev = 0;
                                for (int b = 0; b < bits; ++b) rev |= ((ph >> b) & 1u) << (bits - 1 - b);
                                key2.emplace_back(rev, ph);
                            }
                            std::sort(key2.begin(), key2.end());
                            for (uint32_t t = 0; t < ps.P; ++t) ps.turn[key2[t].second] = (uint8_t)t;
                        }
                        CN_TRY(cnscan::exclusive_u32(h, "sdust_prep", d_flag, (int64_t)nc, 1, d_rank, d_pp, d_tot + 7));
                        // The number of flagged chunks comes back to the host: it decides whether the dense kernel is worth its
                        // latency (one job = one chunk = several milliseconds for a wave)
                        CN_HIP(h, hipMemcpyAsync(p_tot + 200, d_tot + 7, 8, hipMemcpyDeviceToHost, h->stream));
                        CN_HIP(h, hipStreamSynchronize(h->stream));
                        a->sd_flagged = (int64_t)p_tot[200];
                        n_dense = dense_mode ? (unsigned long long)a->sd_flagged : 0;
                        if (dense_mode == 1 && n_dense < std::max<unsigned long long>(1024, nc / 256)) n_dense = 0;
                        CN_LAUNCH(h, "sdust_prep", sd_order<<<dim3(nbs), dim3(256), 0, h->stream>>>(d_flag, d_rank, d_tot + 7, (int32_t)nc, d_perm, ps,
                                                                                               n_dense ? d_list : nullptr, d_claim, (int32_t)a->sd_tail0));
                        // keep {claim, perm, dense list} (contiguous in the workspace) with the assembly
                        if (a->d_sd_plan) { (void)hipFree(a->d_sd_plan); a->d_sd_plan = nullptr; }
                        if (hipMalloc((void **)&a->d_sd_plan, (3 * nc + 160) * 4) != hipSuccess) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: device allocation failed");
                        CN_HIP(h, hipMemcpyAsync(a->d_sd_plan, d_claim, (3 * nc + 160) * 4, hipMemcpyDeviceToDevice, h->stream));
                        a->sd_plan_key = plan_key;
                        a->sd_plan_dense = (int64_t)n_dense;
                    } else {
                        n_dense = (unsigned long long)a->sd_plan_dense;
                        CN_HIP(h, hipMemcpyAsync(d_claim, a->d_sd_plan, nc * 4, hipMemcpyDeviceToDevice, h->stream));
                        CN_HIP(h, hipMemsetAsync(d_cnt, 0, nc * 4, h->stream));
                    }
                    d_perm = a->d_sd_plan + nc;            // (read-only from here on: the assembly's copy)
                    d_list = d_perm + nc + 160;
                    A.perm = d_perm;
                    A.q_len = (int32_t)nc + 160;
                    if (n_dense > 0) {
                        if (!h->stream2) {
                            // (highest priority: where both kernels have workgroups waiting, the dense ones are placed first)
                            int pr_least = 0, pr_greatest = 0;
                            (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
                            CN_HIP(h, hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, pr_greatest));
                            CN_HIP(h, hipEventCreateWithFlags(&h->ev2, hipEventDisableTiming));
                            CN_HIP(h, hipEventCreateWithFlags(&h->ev1, hipEventDisableTiming));
                        }
                        CN_HIP(h, hipEventRecord(h->ev1, h->stream));                  // the dense list is complete
                        CN_HIP(h, hipStreamWaitEvent(h->stream2, h->ev1, 0));
                        const unsigned nbd = (unsigned)std::min<unsigned long long>((n_dense + 63) / 64, 8ull * (unsigned)std::max(h->sd_cus, 1));
                        const bool timed = cn_timed(h, "sdust_dense");
                        cornetto_accel::Rec rd{"sdust_dense", nullptr, nullptr};
                        if (timed) {
                            rd.a = cn_event(h);
                            rd.b = cn_event(h);
                            CN_HIP(h, hipEventRecord(rd.a, h->stream2));
                        }
                        volatile uint32_t *started = reinterpret_cast<volatile uint32_t *>(p_tot + 201);
                        *started = 0;
                        sdust_dense<<<dim3(nbd), dim3(64), 0, h->stream2>>>(A, d_list, (int)n_dense, const_cast<uint32_t *>(started));
                        CN_HIP(h, hipGetLastError());
                        {
                            // wait (bounded: 2 ms) until its first block runs: the main kernel, launched next, fills every
                            // wave slot that is left and would otherwise keep the dense blocks waiting for its last wave
                            struct timespec w0, w1;
                            clock_gettime(CLOCK_MONOTONIC, &w0);
                            while (*started == 0) {
                                clock_gettime(CLOCK_MONOTONIC, &w1);
                                if ((w1.tv_sec - w0.tv_sec) * 1000000000L + (w1.tv_nsec - w0.tv_nsec) > 2000000L) break;
                            }
                        }
                        if (timed) {
                            CN_HIP(h, hipEventRecord(rd.b, h->stream2));
                            h->recs.push_back(rd);
                        }
                        CN_HIP(h, hipEventRecord(h->ev2, h->stream2));
                        dense_pending = true;
                    }
                }
                // as many waves as the chip holds at once (LDS-bound: 18 per CU); each lane works through the queue
                if ((unsigned)sd_waves < nb) nb = (unsigned)sd_waves;
                nb = (nb + SD_WPB - 1) / SD_WPB * SD_WPB;
                if (want_stats) CN_LAUNCH(h, "sdust_kernel", sdust_w64<true><<<dim3(nb / SD_WPB), dim3(64 * SD_WPB), 0, h->stream>>>(A));
                else CN_LAUNCH(h, "sdust_kernel", sdust_w64<false><<<dim3(nb / SD_WPB), dim3(64 * SD_WPB), 0, h->stream>>>(A));
            } else if (W - 2 <= 64) {
                CN_LAUNCH(h, "sdust_kernel", sdust_kernel<64><<<dim3(nb), dim3(64), 0, h->stream>>>(A));
            } else if (W - 2 <= 255) {
                CN_LAUNCH(h, "sdust_kernel", sdust_kernel<256><<<dim3(nb), dim3(64), 0, h->stream>>>(A));
            } else {
                int rc = 512;
                while (rc < W - 2) rc <<= 1;
                const size_t NL = (size_t)nb * 64;
                uint8_t *g = (uint8_t *)cn_ws(h, WS_SD_OFF, NL * ((size_t)rc * 5 + 512) + 64);
                if (!g) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation of %zu bytes failed", NL * ((size_t)rc * 5 + 512));
                uint32_t *g_cw = reinterpret_cast<uint32_t *>(g), *g_cv = g_cw + 64 * NL, *g_slot = g_cv + 64 * NL;
                uint8_t *g_ring = reinterpret_cast<uint8_t *>(g_slot + (size_t)rc * NL);
                CN_LAUNCH(h, "sdust_kernel", sdust_kernel_g<<<dim3(nb), dim3(64), 0, h->stream>>>(A, g_ring, g_cw, g_cv, g_slot, rc));
            }
            if (!sift_on) __atomic_fetch_add(&h->launch_seq, 1ull, __ATOMIC_RELEASE);   // (cornetto_accel_launch_count: the other kernel families count as well)
            if (dense_pending) {
                CN_HIP(h, hipStreamWaitEvent(h->stream, h->ev2, 0));
                dense_pending = false;                 // (joined on the device: the stream's later work waits for it)
            }
            // ordered position of every chunk's intervals (chunks are in contig order) + grand total
            CN_TRY(cnscan::exclusive_u32(h, "sdust_scan", d_cnt, (int64_t)nc, 1, d_off, d_part, d_tot));
            // ---- the rest of the call in one go when the last call for this table left its counts behind (round 4): gather, merge (one
            // launch that reads the number of rows from the device) and the result copy are sized by them, ONE synchronisation, and
            // the counts are checked afterwards — anything that does not fit takes the steps below as before
            if (one_go) {
                size_t n_cap = (size_t)(a->sd_est_rows + a->sd_est_rows / 8 + 4096);
                if (CN_DEV_INT("CORNETTO_SDUST_EST_FORCE", 0) > 0) n_cap = (size_t)CN_DEV_INT("CORNETTO_SDUST_EST_FORCE", 0);   // (tests: an estimate that does not hold)
                const size_t m_cap = (size_t)(a->sd_est_out + a->sd_est_out / 16 + 1024);
                uint8_t *ws = (uint8_t *)cn_ws(h, WS_SD_DST, 2 * n_cap * sizeof(cornetto_ivl_t) + cnivl::ws_bytes(n_cap));
                // (a pinned array whatever its size — the pool's smallest block is 1 MB —: a copy into plain memory is not asynchronous, and
                // cornetto_sdust_asm_begin() returns behind it)
                cornetto_ivl_t *of = (cornetto_ivl_t *)cn_result_alloc(std::max<size_t>(m_cap * sizeof(cornetto_ivl_t), (size_t)1 << 20));
                if (ws && of && n_cap < 0x7fffffffull) {
                    cornetto_ivl_t *d_dst = (cornetto_ivl_t *)(ws + cnivl::ws_bytes(n_cap)), *d_st = d_dst + n_cap;
                    const unsigned nbg = (unsigned)((nc + 255) / 256);
                    CN_LAUNCH(h, "sdust_gather", sdust_gather<<<dim3(nbg), dim3(256), 0, h->stream>>>(d_out, d_cnt, d_off, (uint32_t)cap, d_chunks, (int32_t)nc, d_dst, (uint32_t)n_cap));
                    CN_HIP(h, hipMemsetAsync(d_tot + 9, 0xFF, 8, h->stream));          // (the merge writes the count only when every row had its tile)
                    int rcf = cnivl::merge_fused(h, "sdust_stitch", d_dst, d_tot, (int64_t)n_cap, 0, d_st, d_tot + 9);
                    if (rcf != CORNETTO_OK) { cornetto_free(of); return rcf; }
                    stamp("tail queued");
                    if (hipMemcpyAsync(of, d_st, m_cap * sizeof(cornetto_ivl_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                        hipMemcpyAsync(p_tot, d_tot, 128, hipMemcpyDeviceToHost, h->stream) != hipSuccess) {
                        (void)hipStreamSynchronize(h->stream);
                        cornetto_free(of);
                        return cn_fail(h, CORNETTO_E_HIP, "sdust: stitch / copy back failed");
                    }
                    if (phase == 1) {                                          // cornetto_sdust_asm_end() waits and checks
                        h->sd_pend.state = 1; h->sd_pend.a = a; h->sd_pend.T = T; h->sd_pend.W = W; h->sd_pend.of = of;
                        h->sd_pend.m_cap = m_cap; h->sd_pend.n_cap = n_cap; h->sd_pend.cap = cap; h->sd_pend.key = key; h->sd_pend.walk_pending = sift_walk_pending;
                        return CORNETTO_OK;
                    }
                    const int fin = sd_one_go_finish(h, a, of, p_tot, cap, n_cap, m_cap, key, sift_walk_pending, ivls, n_ivls);
                    if (fin <= 0) {
                        if (fin == 0) stamp("results on the host (one go)");
                        return fin;
                    }
                } else if (of) {
                    cornetto_free(of);
                }
            }
            stamp("main kernel queued");
            CN_HIP(h, hipMemcpyAsync(p_tot, d_tot, want_stats ? 2048 : 128, hipMemcpyDeviceToHost, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));
            stamp("main kernel done");
            if (want_stats) {
                memcpy(h->sd_last, p_tot, 2048);
                if (a->sd_flagged >= 0) h->sd_last[7] = (unsigned long long)a->sd_flagged;   // (counted by the call that built the plan)
                h->sd_last[254] = nb;
                h->sd_last[255] = nc;
            }
            if (env_stats)
                for (int b = 0; b < 32; ++b)
                    if (p_tot[16 + 4 * b])
                        fprintf(stderr, "[sdust stats]   waves that ran %4.1f-%4.1f ms: %6llu, job fetches %llu, find_perfect calls %llu (%llu with candidates)\n", b * 0.5, b * 0.5 + 0.5,
                                p_tot[16 + 4 * b], p_tot[18 + 4 * b], p_tot[19 + 4 * b], p_tot[17 + 4 * b]);
            if (env_stats && sift_on)
                fprintf(stderr, "[sdust stats] sift: tiles %llu, positions with ct > T/10 %llu, after L1 %llu, after L2 %llu; resolve: steps %llu, window reads %llu (behind a gap of 2 / 3-4 / 5-8: %llu / %llu / %llu), dp tiles %llu; passes with candidates %llu; base-by-base steps of chunks with other bytes %llu\n",
                        p_tot[206], p_tot[203], p_tot[204], p_tot[205], p_tot[200], p_tot[201], p_tot[209], p_tot[210], p_tot[211], p_tot[208], p_tot[202], p_tot[207]);
            if (env_stats)
                fprintf(stderr, "[sdust stats] flagged low-complexity %llu; chunks %zu waves %u wave-steps %llu find_perfect calls %llu (%llu with candidates) plain groups %llu; wave time avg %.1f us max %.1f us; queue: %.1f fetch rounds and %.1f us per wave\n", p_tot[7], nc, nb,
                        p_tot[2], p_tot[3], p_tot[10], p_tot[4], nb ? (double)p_tot[5] / nb / 100.0 : 0.0, (double)p_tot[6] / 100.0, nb ? (double)p_tot[11] / nb : 0.0, nb ? (double)p_tot[12] / nb / 100.0 : 0.0);
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
            if (sift_walk_pending && !need_wtab) a->sd_walk_key = key;      // (the list is complete: every chunk was looked at)
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
        stamp("result buffer");
        if (tot > 0) {
            // dense list + stitched list + the merge's scratch.  The chunk lists are merged with the reference's own rule
            // (src/sdust/sdust.c:94-98) on the device: ivlmerge.hpp, distance 0.
            const size_t n = (size_t)tot;
            uint8_t *ws = (uint8_t *)cn_ws(h, WS_SD_DST, 2 * n * sizeof(cornetto_ivl_t) + cnivl::ws_bytes(n));
            if (!ws) { cornetto_free(o); return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation failed"); }
            cornetto_ivl_t *d_dst = (cornetto_ivl_t *)(ws + cnivl::ws_bytes(n)), *d_st = d_dst + n;
            const unsigned nbg = (unsigned)((nc + 255) / 256);
            CN_LAUNCH(h, "sdust_gather", sdust_gather<<<dim3(nbg), dim3(256), 0, h->stream>>>(d_out, d_cnt, d_off, (uint32_t)cap, d_chunks, (int32_t)nc, d_dst));
            CN_TRY(cnivl::merge(h, "sdust_stitch", d_dst, (int64_t)n, 0, ws, d_st, d_tot + 9));
            stamp("gather+stitch queued");
            if (trace) { (void)hipStreamSynchronize(h->stream); stamp("gather+stitch done (trace only)"); }
            if (hipMemcpyAsync(o, d_st, n * sizeof(cornetto_ivl_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipMemcpyAsync(p_tot, d_tot + 9, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) {
                cornetto_free(o);
                return cn_fail(h, CORNETTO_E_HIP, "sdust: stitch / copy back failed");
            }
            n_out = (int64_t)p_tot[0];                 // only the first n_out entries of o are meaningful
            stamp("results on the host");
        }
        if (sift_on) {                                 // what the next call for this table may count on
            a->sd_est_key = est_key;
            a->sd_est_rows = (int64_t)tot;
            a->sd_est_out = n_out;
        }
    }
    cn_timing_end(h);
    if (!o) {
        o = (cornetto_ivl_t *)malloc(sizeof(cornetto_ivl_t));
        if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: host allocation failed");
    }
    if (phase == 1) {                                  // (the call ran to its end inside _begin: _end hands the result over)
        h->sd_pend.state = 2; h->sd_pend.a = a; h->sd_pend.T = T; h->sd_pend.W = W; h->sd_pend.of = o; h->sd_pend.n_done = n_out;
        return CORNETTO_OK;
    }
    *ivls = o;
    *n_ivls = n_out;
    return CORNETTO_OK;
}
}  // namespace

extern "C" {

int cornetto_sdust_asm(cornetto_accel_t *h, const cornetto_asm_t *a_in, int32_t T, int32_t W, cornetto_ivl_t **ivls, int64_t *n_ivls)
{
    if (h && h->sd_pend.state != 0) {                  // a _begin nobody finished: its result is dropped
        cornetto_ivl_t *dv = nullptr;
        int64_t dn = 0;
        if (cornetto_sdust_asm_end(h, reinterpret_cast<const cornetto_asm_t *>(h->sd_pend.a), h->sd_pend.T, h->sd_pend.W, &dv, &dn) == CORNETTO_OK) cornetto_free(dv);
    }
    return sdust_asm_impl(h, a_in, T, W, ivls, n_ivls, 0);
}

int cornetto_sdust_asm_begin(cornetto_accel_t *h, const cornetto_asm_t *a_in, int32_t T, int32_t W)
{
    if (!h || !a_in) return cn_fail(h, CORNETTO_E_ARG, "sdust: bad argument");
    if (h->sd_pend.state != 0) return cn_fail(h, CORNETTO_E_ARG, "sdust: cornetto_sdust_asm_begin() twice without cornetto_sdust_asm_end()");
    cornetto_ivl_t *dv = nullptr;
    int64_t dn = 0;
    return sdust_asm_impl(h, a_in, T, W, &dv, &dn, 1);
}

int cornetto_sdust_asm_end(cornetto_accel_t *h, const cornetto_asm_t *a_in, int32_t T, int32_t W, cornetto_ivl_t **ivls, int64_t *n_ivls)
{
    if (!h || !a_in || !ivls || !n_ivls) return cn_fail(h, CORNETTO_E_ARG, "sdust: bad argument");
    *ivls = nullptr;
    *n_ivls = 0;
    const cornetto_accel::SdPend P = h->sd_pend;
    h->sd_pend = cornetto_accel::SdPend{};
    if (P.state != 0 && (P.a != a_in || P.T != T || P.W != W)) {      // not the call that was begun: that one is finished and dropped first
        if (P.state == 2) cornetto_free(P.of);
        else { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); cornetto_free(P.of); }
        return sdust_asm_impl(h, a_in, T, W, ivls, n_ivls, 0);
    }
    if (P.state == 2) {
        *ivls = (cornetto_ivl_t *)P.of;
        *n_ivls = P.n_done;
        return CORNETTO_OK;
    }
    if (P.state == 1) {
        CN_HIP(h, hipSetDevice(h->device));
        unsigned long long *p_tot = (unsigned long long *)cn_pin(h, PIN_SMALL, 2048);
        if (!p_tot) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: workspace allocation failed");
        const int fin = sd_one_go_finish(h, const_cast<cornetto_asm_t *>(a_in), (cornetto_ivl_t *)P.of, p_tot, P.cap, P.n_cap, P.m_cap, P.key, P.walk_pending, ivls, n_ivls);
        if (fin <= 0) return fin;
    }
    return sdust_asm_impl(h, a_in, T, W, ivls, n_ivls, 0);
}

int cornetto_accel_sdust_stats(cornetto_accel_t *h, int enable, uint64_t *out, int cap)
{
    if (!h) return CORNETTO_E_ARG;
    h->sd_stats = enable ? 1 : 0;
    int n = 0;
    if (out)
        for (; n < cap && n < 256; ++n) out[n] = h->sd_last[n];
    return n;
}

// process-wide handle for the sdust() / sdust_core() drop-ins
static cornetto_accel_t *g_handle = nullptr;

// = sdust_buf_t (src/sdust/sdust.c:54-59) as far as a caller can see it: the owner of the result array of sdust_core()
struct cornetto_sdust_buf {
    uint64_t *res;      // plain malloc / realloc memory (sdust() hands it to a caller who free()s it)
    int64_t cap;        // entries allocated
};

cornetto_sdust_buf_t *cornetto_sdust_buf_init(void *km)
{
    if (km) return nullptr;                       // kalloc pools are not supported (the reference passes 0: src/sdust/sdust.c:199)
    return (cornetto_sdust_buf_t *)calloc(1, sizeof(cornetto_sdust_buf));
}

void cornetto_sdust_buf_destroy(cornetto_sdust_buf_t *buf)
{
    if (!buf) return;
    free(buf->res);
    free(buf);
}

const uint64_t *cornetto_sdust_core(const uint8_t *seq, int l_seq, int T, int W, int *n, cornetto_sdust_buf_t *buf)
{
    if (n) *n = -1;
    if (!seq || !n || !buf) return nullptr;
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
    if (ni > buf->cap || !buf->res) {
        const int64_t cap = ni > 16 ? ni : 16;
        uint64_t *r = (uint64_t *)realloc(buf->res, (size_t)cap * sizeof(uint64_t));
        if (!r) { cornetto_free(iv); return nullptr; }
        buf->res = r;
        buf->cap = cap;
    }
    for (int64_t i = 0; i < ni; ++i) buf->res[i] = (uint64_t)(uint32_t)iv[i].start << 32 | (uint32_t)iv[i].finish;   // :91
    cornetto_free(iv);            // library memory (pinned pool for large results): never free()
    *n = (int)ni;
    return buf->res;
}

uint64_t *cornetto_sdust(void *km, const uint8_t *seq, int l_seq, int T, int W, int *n)
{
    // src/sdust/sdust.c:162-171: a fresh buf, sdust_core, the result array detached from the buf and handed to the caller
    if (n) *n = -1;
    if (km || !seq || !n) return nullptr;
    cornetto_sdust_buf_t *buf = cornetto_sdust_buf_init(nullptr);
    if (!buf) return nullptr;
    uint64_t *ret = const_cast<uint64_t *>(cornetto_sdust_core(seq, l_seq, T, W, n, buf));
    buf->res = nullptr;
    cornetto_sdust_buf_destroy(buf);
    return ret;
}

}  // extern "C"

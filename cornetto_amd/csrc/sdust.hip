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

namespace {

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
};

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
#undef RING
#undef CW
#undef CV
#undef SLOT
}

__global__ void sdust_gather(const uint2 *in, const uint32_t *cnt, const int64_t *dst_off, uint32_t cap,
                             int32_t n_chunks, uint2 *dst)
{
    // one wavefront per chunk row
    const int cid = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (cid >= n_chunks) return;
    const uint32_t n = cnt[cid];
    const uint2 *src = in + (size_t)cid * cap;
    uint2 *d = dst + dst_off[cid];
    for (uint32_t i = threadIdx.x & 63; i < n; i += 64) d[i] = src[i];
}

int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

}  // namespace

extern "C" {

int cornetto_sdust_asm(cornetto_accel_t *h, const cornetto_asm_t *a, int32_t T, int32_t W, cornetto_ivl_t **ivls,
                       int64_t *n_ivls)
{
    if (!h || !a || !ivls || !n_ivls) return cn_fail(h, CORNETTO_E_ARG, "sdust: bad argument");
    *ivls = nullptr;
    *n_ivls = 0;
    if (W < 3 || W > 258) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: -w %d outside 3..258 (the reference crashes below 3)", W);
    if (T < 0 || T > (1 << 20)) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: -t %d outside 0..2^20", T);
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);

    // chunking: enough lanes to fill the chip a few times over; CORNETTO_SDUST_CHUNK overrides (tests use
    // tiny chunks to stress the speculative start)
    int64_t chunk = env_int("CORNETTO_SDUST_CHUNK", 0);
    if (chunk <= 0) {
        const int64_t target_lanes = 256LL * 8 * 64 * 4;
        chunk = a->total / target_lanes;
        chunk = std::max<int64_t>(1024, std::min<int64_t>(16384, chunk));
    }
    chunk = std::max<int64_t>(16, chunk);
    std::vector<SdChunk> chunks;
    for (int32_t c = 0; c < a->n; ++c)
        for (int64_t s = 0; s < a->len[c]; s += chunk)
            chunks.push_back(SdChunk{c, (int32_t)s, (int32_t)std::min<int64_t>(a->len[c], s + chunk)});
    const size_t nc = chunks.size();
    std::vector<cornetto_ivl_t> res;
    if (nc > 0) {
        if (nc > 0x7fffffffull) return cn_fail(h, CORNETTO_E_UNSUPPORTED, "sdust: too many chunks");
        DevBuf d_chunks, d_out, d_cnt, d_off, d_dst;
        if (d_chunks.alloc(nc * sizeof(SdChunk)) != hipSuccess || d_cnt.alloc(nc * 4) != hipSuccess)
            return cn_fail(h, CORNETTO_E_NOMEM, "sdust: device allocation failed");
        CN_HIP(h, hipMemcpyAsync(d_chunks.p, chunks.data(), nc * sizeof(SdChunk), hipMemcpyHostToDevice, h->stream));
        uint32_t cap = (uint32_t)std::max<int64_t>(16, chunk / 32);
        cap = (uint32_t)env_int("CORNETTO_SDUST_CAP", (int)cap);
        std::vector<uint32_t> cnt(nc);
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (d_out.alloc(nc * (size_t)cap * sizeof(uint2)) != hipSuccess)
                return cn_fail(h, CORNETTO_E_NOMEM, "sdust: device allocation of %zu bytes failed", nc * (size_t)cap * sizeof(uint2));
            SdArgs A{a->d_bases, a->d_off, a->d_len, d_chunks.as<SdChunk>(), (int32_t)nc, T, W, d_out.as<uint2>(), d_cnt.as<uint32_t>(), cap};
            const unsigned nb = (unsigned)((nc + 63) / 64);
            if (W - 2 <= 64) CN_LAUNCH(h, "sdust_kernel", sdust_kernel<64><<<dim3(nb), dim3(64), 0, h->stream>>>(A));
            else CN_LAUNCH(h, "sdust_kernel", sdust_kernel<256><<<dim3(nb), dim3(64), 0, h->stream>>>(A));
            CN_HIP(h, hipMemcpyAsync(cnt.data(), d_cnt.p, nc * 4, hipMemcpyDeviceToHost, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));
            const uint32_t mx = *std::max_element(cnt.begin(), cnt.end());
            if (mx <= cap) break;
            if (attempt == 1) return cn_fail(h, CORNETTO_E_HIP, "sdust: chunk produced %u intervals after resizing to %u", mx, cap);
            cap = mx;   // rerun with room for the densest chunk: results are never truncated
        }
        std::vector<int64_t> off(nc + 1, 0);
        for (size_t i = 0; i < nc; ++i) off[i + 1] = off[i] + cnt[i];
        const int64_t tot = off[nc];
        std::vector<uint2> flat((size_t)tot);
        if (tot > 0) {
            if (d_off.alloc(nc * 8) != hipSuccess || d_dst.alloc((size_t)tot * sizeof(uint2)) != hipSuccess)
                return cn_fail(h, CORNETTO_E_NOMEM, "sdust: device allocation failed");
            CN_HIP(h, hipMemcpyAsync(d_off.p, off.data(), nc * 8, hipMemcpyHostToDevice, h->stream));
            const unsigned nb = (unsigned)((nc + 3) / 4);
            CN_LAUNCH(h, "sdust_gather", sdust_gather<<<dim3(nb), dim3(256), 0, h->stream>>>(d_out.as<uint2>(), d_cnt.as<uint32_t>(), d_off.as<int64_t>(), cap, (int32_t)nc, d_dst.as<uint2>()));
            CN_HIP(h, hipMemcpyAsync(flat.data(), d_dst.p, (size_t)tot * sizeof(uint2), hipMemcpyDeviceToHost, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));
        }
        // stitch chunk lists in order with the reference's merge rule (src/sdust/sdust.c:94-98)
        res.reserve((size_t)tot);
        int32_t cur_ctg = -1;
        for (size_t ci = 0; ci < nc; ++ci) {
            const int32_t ctg = chunks[ci].ctg;
            for (int64_t j = off[ci]; j < off[ci + 1]; ++j) {
                const int32_t s = (int32_t)flat[j].x, f = (int32_t)flat[j].y;
                if (ctg == cur_ctg && !res.empty() && s <= res.back().finish) {
                    if (f > res.back().finish) res.back().finish = f;
                } else {
                    res.push_back(cornetto_ivl_t{ctg, s, f});
                    cur_ctg = ctg;
                }
            }
        }
    }
    cn_timing_end(h);
    cornetto_ivl_t *o = (cornetto_ivl_t *)malloc((res.size() ? res.size() : 1) * sizeof(cornetto_ivl_t));
    if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "sdust: host allocation failed");
    if (!res.empty()) memcpy(o, res.data(), res.size() * sizeof(cornetto_ivl_t));
    *ivls = o;
    *n_ivls = (int64_t)res.size();
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

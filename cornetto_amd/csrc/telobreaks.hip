// telobreaks.hip — "telomere breaks": low-complexity runs that contain a telomere repeat with its 100-base flanks.
// Replaces the bitset loops of src/telomere_breaks.c:79-148 of the reference; the text
// parsing and the name table stay on the host (cli/telobreaks_main.c), as does the print order, which is the bucket
// order of the reference's khash table (cornetto_khash_str_order below restates it).
//
// The reference keeps one bit per base and per contig and walks bits one at a time.  Same bitsets here, 64 bits per
// word, all contigs in one array (every contig starts on a word):
//   tb_fill   one thread per BED interval: OR its bits in (edge words atomically, inner words whole)        :79-90
//   tb_mark   one thread per telofind row: are [start-100, end+100) all set?  then walk left / right over words to
//             the ends of the run (ffs on the inverted words) and OR the run into the second bitset          :95-128
//   tb_count / tb_emit   runs of the second bitset: a run starts where a bit is set and its predecessor is not
//             (word-level shifts with a carry from the neighbouring word), per-tile counts, device scan, dense
//             records {contig, first - 1 clamped at 0, last} = what :140-142 prints                          :133-148
// All of it is HBM-bound word traffic: 2 x 1/8 byte per base cleared, the second bitset read twice.
#include <algorithm>
#include <cstring>
#include <string>

#include "common.hpp"
#include "scan.hpp"

namespace {

constexpr int TB_MIN_TEL = 24;          // MIN_TEL, src/telomere_breaks.c:10
constexpr int TB_FLANK = 100;           // :102-103
constexpr int TB_TILE_WORDS = 1024;     // words of one contig per workgroup of tb_count / tb_emit (256 threads x 4)

struct TbArgs {
    const int32_t *ctg_len;
    const int64_t *woff;                // first word of every contig
    int32_t n_ctg;
    unsigned long long *bits, *fin;
    uint32_t *err;                      // set when a coordinate lies outside its contig (undefined behaviour in the reference)
};

__device__ __forceinline__ unsigned long long tb_mask_from(int b) { return ~0ull << b; }                   // bits b..63
__device__ __forceinline__ unsigned long long tb_mask_below(int b) { return b >= 64 ? ~0ull : ~(~0ull << b); }   // bits 0..b-1

// OR the bits [a, b) of a contig's bitset (a < b)
__device__ void tb_or_range(unsigned long long *w, int a, int b)
{
    const int wa = a >> 6, wb = (b - 1) >> 6;
    if (wa == wb) {
        atomicOr(&w[wa], tb_mask_from(a & 63) & tb_mask_below(((b - 1) & 63) + 1));
        return;
    }
    atomicOr(&w[wa], tb_mask_from(a & 63));
    for (int i = wa + 1; i < wb; ++i) w[i] = ~0ull;        // whole words: every writer stores the same value
    atomicOr(&w[wb], tb_mask_below(((b - 1) & 63) + 1));
}

__global__ void tb_fill(TbArgs A, const cornetto_ivl_t *sd, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cornetto_ivl_t v = sd[i];
    if (v.ctg < 0 || v.ctg >= A.n_ctg) return;                                   // name not in the lens file (:83)
    // An interval that reaches beyond its contig's end is what sdust itself prints for a low-complexity run at the end of a contig — a telomere —
    // (the finish of such an interval is up to W beyond the last base: src/sdust/sdust.c:88-102 does not cut it), and the reference sets those bits
    // beyond its bitset (:85) and never reads them (:103,:118,:136 stop at the length): they are cut here.  A negative start has no such reading.
    if (v.start < 0) { atomicOr(A.err, 1u); return; }
    const int len = A.ctg_len[v.ctg];
    const int fin = v.finish > len ? len : v.finish;
    if (v.start >= fin) return;
    tb_or_range(A.bits + A.woff[v.ctg], v.start, fin);
}

// are the bits [a, b) all set?
__device__ bool tb_all_set(const unsigned long long *w, int a, int b)
{
    if (a >= b) return true;                                                     // the reference's loop runs zero times
    const int wa = a >> 6, wb = (b - 1) >> 6;
    for (int i = wa; i <= wb; ++i) {
        unsigned long long need = ~0ull;
        if (i == wa) need &= tb_mask_from(a & 63);
        if (i == wb) need &= tb_mask_below(((b - 1) & 63) + 1);
        if ((w[i] & need) != need) return false;
    }
    return true;
}

__global__ void tb_mark(TbArgs A, const cornetto_telrow_t *tel, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cornetto_telrow_t t = tel[i];
    if (t.matched < TB_MIN_TEL) return;                                          // :98
    if (t.ctg < 0 || t.ctg >= A.n_ctg) return;                                   // :100
    const int len = A.ctg_len[t.ctg];
    if (t.start < 0 || t.end > len || t.start >= t.end) { atomicOr(A.err, 2u); return; }
    const unsigned long long *w = A.bits + A.woff[t.ctg];
    const int a = t.start - TB_FLANK < 0 ? 0 : t.start - TB_FLANK;               // :102
    const int b = t.end + TB_FLANK > len ? len : t.end + TB_FLANK;               // :103
    if (!tb_all_set(w, a, b)) return;
    // :114-121: the run of set bits around [start, end).  Left: the highest clear bit below `start`.
    int lo = 0;
    for (int wi = (t.start - 1) >> 6, first = 1; t.start > 0 && wi >= 0; --wi, first = 0) {
        unsigned long long z = ~w[wi];
        if (first) z &= tb_mask_below(((t.start - 1) & 63) + 1);                 // only bits < start
        if (z) { lo = wi * 64 + 63 - __builtin_clzll(z) + 1; break; }
    }
    // right: the lowest clear bit at or above `end`
    int hi = len;
    const int nw = (len + 63) >> 6;
    for (int wi = t.end >> 6, first = 1; t.end < len && wi < nw; ++wi, first = 0) {
        unsigned long long z = ~w[wi];
        if (first) z &= tb_mask_from(t.end & 63);
        if (z) { const int p = wi * 64 + __builtin_ctzll(z); hi = p < len ? p : len; break; }
    }
    tb_or_range(A.fin + A.woff[t.ctg], lo, hi);
}

struct TbTile {
    int32_t ctg, w0;                    // contig and first word (within the contig) of the tile
};

// starts / ends of runs inside one word, given the neighbouring bits
__device__ __forceinline__ void tb_edges(const unsigned long long *w, int wi, int nw, unsigned long long &starts, unsigned long long &ends)
{
    const unsigned long long x = w[wi];
    const unsigned long long prev = wi > 0 ? w[wi - 1] >> 63 : 0ull, next = wi + 1 < nw ? w[wi + 1] & 1ull : 0ull;
    starts = x & ~((x << 1) | prev);
    ends = x & ~((x >> 1) | (next << 63));
}

__global__ __launch_bounds__(256) void tb_count(TbArgs A, const TbTile *tiles, uint32_t *n_start, uint32_t *n_end)
{
    __shared__ uint32_t ws[4], we[4];
    const TbTile t = tiles[blockIdx.x];
    const unsigned long long *w = A.fin + A.woff[t.ctg];
    const int nw = (A.ctg_len[t.ctg] + 63) >> 6;
    uint32_t cs = 0, ce = 0;
    for (int k = 0; k < TB_TILE_WORDS / 256; ++k) {
        const int wi = t.w0 + k * 256 + (int)threadIdx.x;
        if (wi < nw) {
            unsigned long long s, e;
            tb_edges(w, wi, nw, s, e);
            cs += (uint32_t)__popcll(s);
            ce += (uint32_t)__popcll(e);
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        cs += __shfl_xor(cs, d);
        ce += __shfl_xor(ce, d);
    }
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = cs; we[threadIdx.x >> 6] = ce; }
    __syncthreads();
    if (threadIdx.x == 0) {
        n_start[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
        n_end[blockIdx.x] = we[0] + we[1] + we[2] + we[3];
    }
}

// the k-th run start and the k-th run end of the whole array belong to the same run (runs do not nest); one
// wavefront per tile walks its words in order (runs are rare: a few per million bases)
__global__ __launch_bounds__(64) void tb_emit(TbArgs A, const TbTile *tiles, const uint32_t *off_start, const uint32_t *off_end,
                                               const uint32_t *n_start, const uint32_t *n_end, cornetto_ivl_t *out)
{
    const TbTile t = tiles[blockIdx.x];
    if (n_start[blockIdx.x] == 0 && n_end[blockIdx.x] == 0) return;
    const unsigned long long *w = A.fin + A.woff[t.ctg];
    const int nw = (A.ctg_len[t.ctg] + 63) >> 6, lane = threadIdx.x;
    uint32_t ks = off_start[blockIdx.x], ke = off_end[blockIdx.x];
    for (int base = 0; base < TB_TILE_WORDS; base += 64) {
        const int wi = t.w0 + base + lane;
        unsigned long long s = 0, e = 0;
        if (wi < nw) tb_edges(w, wi, nw, s, e);
        // exclusive prefix of the per-lane counts over the wave
        uint32_t cs = (uint32_t)__popcll(s), ce = (uint32_t)__popcll(e), ps = cs, pe = ce;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t a = __shfl_up(ps, d), b = __shfl_up(pe, d);
            if (lane >= d) { ps += a; pe += b; }
        }
        uint32_t is = ks + ps - cs, ie = ke + pe - ce;
        while (s) {
            const int p = wi * 64 + __builtin_ctzll(s);
            s &= s - 1;
            out[is].ctg = t.ctg;
            out[is].start = p - 1 < 0 ? 0 : p - 1;                                // :140
            ++is;
        }
        while (e) {
            out[ie].finish = wi * 64 + __builtin_ctzll(e);                        // end - 1, :142
            e &= e - 1;
            ++ie;
        }
        ks += __shfl(ps, 63);
        ke += __shfl(pe, 63);
    }
}

}  // namespace

extern "C" {

int cornetto_telobreaks(cornetto_accel_t *h, const int32_t *ctg_len, int32_t n_ctg, const cornetto_ivl_t *sd, int64_t n_sd,
                        const cornetto_telrow_t *tel, int64_t n_tel, cornetto_ivl_t **out, int64_t *n_out)
{
    if (!h || !out || !n_out || n_ctg < 0 || n_sd < 0 || n_tel < 0 || (n_ctg > 0 && !ctg_len) || (n_sd > 0 && !sd) || (n_tel > 0 && !tel))
        return cn_fail(h, CORNETTO_E_ARG, "telobreaks: bad argument");
    *out = nullptr;
    *n_out = 0;
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    // word layout and tiles
    std::vector<int64_t> woff((size_t)n_ctg + 1, 0);
    std::vector<TbTile> tiles;
    for (int32_t c = 0; c < n_ctg; ++c) {
        if (ctg_len[c] < 0) return cn_fail(h, CORNETTO_E_ARG, "telobreaks: contig %d has a negative length", c);
        const int64_t nw = ((int64_t)ctg_len[c] + 63) >> 6;
        woff[c + 1] = woff[c] + nw;
        for (int64_t w0 = 0; w0 < nw; w0 += TB_TILE_WORDS) tiles.push_back(TbTile{c, (int32_t)w0});
    }
    const size_t W = (size_t)woff[n_ctg], nt = tiles.size();
    cornetto_ivl_t *o = nullptr;
    int64_t n = 0;
    if (W > 0 && n_tel > 0 && n_sd > 0) {
        // bitsets | word offsets | lengths | tiles | counts + offsets + scan partials | records in
        const size_t npart = (nt + 4095) / 4096 + 1;
        const size_t bytes = W * 16 + ((size_t)n_ctg + 1) * 8 + (size_t)n_ctg * 4 + 8 + nt * sizeof(TbTile) + (nt * 4 + npart) * 4 + 64 +
                             (size_t)n_sd * sizeof(cornetto_ivl_t) + (size_t)n_tel * sizeof(cornetto_telrow_t) + 64;
        uint8_t *ws = (uint8_t *)cn_ws(h, WS_TB, bytes);
        unsigned long long *d_small = (unsigned long long *)cn_ws(h, WS_TB_SMALL, 64);
        unsigned long long *p_small = (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
        if (!ws || !d_small || !p_small) return cn_fail(h, CORNETTO_E_NOMEM, "telobreaks: workspace allocation of %zu bytes failed", bytes);
        unsigned long long *d_bits = (unsigned long long *)ws, *d_fin = d_bits + W;
        int64_t *d_woff = (int64_t *)(d_fin + W);
        TbTile *d_tiles = (TbTile *)(d_woff + n_ctg + 1);
        uint32_t *d_ns = (uint32_t *)(d_tiles + nt), *d_ne = d_ns + nt, *d_os = d_ne + nt, *d_oe = d_os + nt, *d_part = d_oe + nt;
        int32_t *d_len = (int32_t *)(d_part + npart);
        cornetto_ivl_t *d_sd = (cornetto_ivl_t *)(d_len + n_ctg + 1);
        cornetto_telrow_t *d_tel = (cornetto_telrow_t *)(d_sd + n_sd);
        TbArgs A{d_len, d_woff, n_ctg, d_bits, d_fin, reinterpret_cast<uint32_t *>(d_small + 2)};
        // (the host arrays below are locals or the caller's: whatever happens, nothing returns before the stream has drained)
        const int rc_q = [&]() -> int {
            CN_HIP(h, hipMemsetAsync(d_bits, 0, W * 16, h->stream));
            CN_HIP(h, hipMemsetAsync(d_small, 0, 64, h->stream));
            CN_HIP(h, hipMemcpyAsync(d_woff, woff.data(), ((size_t)n_ctg + 1) * 8, hipMemcpyHostToDevice, h->stream));
            CN_HIP(h, hipMemcpyAsync(d_len, ctg_len, (size_t)n_ctg * 4, hipMemcpyHostToDevice, h->stream));
            CN_HIP(h, hipMemcpyAsync(d_tiles, tiles.data(), nt * sizeof(TbTile), hipMemcpyHostToDevice, h->stream));
            CN_HIP(h, hipMemcpyAsync(d_sd, sd, (size_t)n_sd * sizeof(cornetto_ivl_t), hipMemcpyHostToDevice, h->stream));
            CN_HIP(h, hipMemcpyAsync(d_tel, tel, (size_t)n_tel * sizeof(cornetto_telrow_t), hipMemcpyHostToDevice, h->stream));
            CN_LAUNCH(h, "tb_fill", tb_fill<<<dim3((unsigned)((n_sd + 255) / 256)), dim3(256), 0, h->stream>>>(A, d_sd, n_sd));
            CN_LAUNCH(h, "tb_mark", tb_mark<<<dim3((unsigned)((n_tel + 255) / 256)), dim3(256), 0, h->stream>>>(A, d_tel, n_tel));
            CN_LAUNCH(h, "tb_count", tb_count<<<dim3((unsigned)nt), dim3(256), 0, h->stream>>>(A, d_tiles, d_ns, d_ne));
            CN_TRY(cnscan::exclusive_u32(h, "tb_scan", d_ns, (int64_t)nt, 1, d_os, d_part, d_small));
            CN_TRY(cnscan::exclusive_u32(h, "tb_scan", d_ne, (int64_t)nt, 1, d_oe, d_part, d_small + 1));
            CN_HIP(h, hipMemcpyAsync(p_small, d_small, 64, hipMemcpyDeviceToHost, h->stream));
            return CORNETTO_OK;
        }();
        if (rc_q != CORNETTO_OK) {
            (void)hipStreamSynchronize(h->stream);
            return rc_q;
        }
        CN_HIP(h, hipStreamSynchronize(h->stream));        // (woff / tiles are locals: the copies above are done)
        if (reinterpret_cast<uint32_t *>(p_small + 2)[0] != 0)
            return cn_fail(h, CORNETTO_E_FORMAT, "telobreaks: %s with coordinates outside its contig (the reference indexes its bitset unchecked)",
                           (reinterpret_cast<uint32_t *>(p_small + 2)[0] & 1u) ? "low-complexity interval" : "telomere row");
        if (p_small[0] != p_small[1]) return cn_fail(h, CORNETTO_E_HIP, "telobreaks: %llu run starts, %llu run ends", p_small[0], p_small[1]);
        n = (int64_t)p_small[0];
        o = (cornetto_ivl_t *)cn_result_alloc(((size_t)n ? (size_t)n : 1) * sizeof(cornetto_ivl_t));
        if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "telobreaks: host allocation failed");
        if (n > 0) {
            cornetto_ivl_t *d_out = (cornetto_ivl_t *)cn_ws(h, WS_TB_OUT, (size_t)n * sizeof(cornetto_ivl_t));
            if (!d_out) { cornetto_free(o); return cn_fail(h, CORNETTO_E_NOMEM, "telobreaks: workspace allocation failed"); }
            int rc = CORNETTO_OK;
            {
                cornetto_accel::Rec r{"tb_emit", cn_event(h), cn_event(h)};
                (void)hipEventRecord(r.a, h->stream);
                tb_emit<<<dim3((unsigned)nt), dim3(64), 0, h->stream>>>(A, d_tiles, d_os, d_oe, d_ns, d_ne, d_out);
                (void)hipEventRecord(r.b, h->stream);
                h->recs.push_back(r);
            }
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(o, d_out, (size_t)n * sizeof(cornetto_ivl_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess)
                rc = cn_fail(h, CORNETTO_E_HIP, "telobreaks: emit / copy back failed");
            if (rc != CORNETTO_OK) { cornetto_free(o); return rc; }
        }
    }
    cn_timing_end(h);
    if (!o) {
        o = (cornetto_ivl_t *)malloc(sizeof(cornetto_ivl_t));
        if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "telobreaks: host allocation failed");
    }
    *out = o;
    *n_out = n;
    return CORNETTO_OK;
}

// ---- khash v0.2.8 (src/khash.h of the reference), string keys, insertions only: which bucket every key ends up in.
// X31 hash on signed char (:395-400); tables of 4, 8, 16 ... buckets, grown when size reaches 0.77 of them (:308-318)
// by re-inserting the old buckets in index order, a displaced key taking the place of the one it meets ("kick-out",
// :278-297); quadratic probing i += ++step (:238, :326).
static uint32_t kh_x31(const char *s)
{
    uint32_t x = (uint32_t)(int)(signed char)*s;
    if (x) for (++s; *s; ++s) x = (x << 5) - x + (uint32_t)(int)(signed char)*s;
    return x;
}

int32_t cornetto_khash_str_order(const char *const *names, int32_t n, int32_t *slot, int32_t *order)
{
    if (n < 0 || (n > 0 && (!names || !slot || !order))) return -1;
    std::vector<int32_t> bucket;                 // id in every bucket, -1 = empty
    std::vector<const char *> key;               // key of every id (the first spelling inserted)
    uint32_t size = 0, upper = 0;
    for (int32_t it = 0; it < n; ++it) {
        if (size >= upper) {
            uint32_t nb = (uint32_t)bucket.size() + 1;
            --nb; nb |= nb >> 1; nb |= nb >> 2; nb |= nb >> 4; nb |= nb >> 8; nb |= nb >> 16; ++nb;
            if (nb < 4) nb = 4;
            if (size < (uint32_t)(nb * 0.77 + 0.5)) {
                const uint32_t ob = (uint32_t)bucket.size(), mask = nb - 1;
                std::vector<int32_t> tab(bucket);
                tab.resize(nb, -1);
                std::vector<char> placed(nb, 0), pending(ob, 0);
                for (uint32_t j = 0; j < ob; ++j) pending[j] = bucket[j] >= 0;
                for (uint32_t j = 0; j < ob; ++j) {
                    if (!pending[j]) continue;
                    int32_t id = tab[j];
                    pending[j] = 0;
                    for (;;) {
                        uint32_t i = kh_x31(key[id]) & mask, step = 0;
                        while (placed[i]) i = (i + (++step)) & mask;
                        placed[i] = 1;
                        if (i < ob && pending[i]) {
                            std::swap(id, tab[i]);
                            pending[i] = 0;
                        } else {
                            tab[i] = id;
                            break;
                        }
                    }
                }
                for (uint32_t j = 0; j < nb; ++j) if (!placed[j]) tab[j] = -1;
                bucket.swap(tab);
                upper = (uint32_t)(nb * 0.77 + 0.5);
            }
        }
        const uint32_t mask = (uint32_t)bucket.size() - 1;
        uint32_t i = kh_x31(names[it]) & mask, step = 0;
        while (bucket[i] >= 0 && strcmp(key[bucket[i]], names[it]) != 0) i = (i + (++step)) & mask;
        if (bucket[i] < 0) {
            bucket[i] = (int32_t)key.size();
            key.push_back(names[it]);
            ++size;
        }
        slot[it] = bucket[i];
    }
    int32_t k = 0;
    for (size_t b = 0; b < bucket.size(); ++b) if (bucket[b] >= 0) order[k++] = bucket[b];
    return (int32_t)key.size();
}

}  // extern "C"

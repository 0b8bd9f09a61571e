// bgin.hip — device ingest of the two lock-step per-base bedgraphs of (no)boringbits; replaces the
// fscanf loop of get_depths() (src/boringbits_main.c:204-287), which is ~95 % of the reference's wall time.
//
// The reference reads both files with fscanf("%s\t%d\t%d\t%d\n"): a record is FOUR WHITESPACE-SEPARATED TOKENS,
// whatever the line structure.  That is what is implemented here, on the device, for text resident in HBM:
//   tk_count / tk_scatter  token starts (non-white byte after a white byte) of a text buffer: per 16-byte lane
//                          piece a white-space bit mask, popcount, block scan, device scan of the tile counts,
//                          then the positions are scattered in order (two streaming reads of the text);
//   bg_records             one thread per record index r: parses tokens 4r..4r+3 of BOTH files, applies the
//                          reference's checks in its order (4 columns :209-222, same name/start/end :224,
//                          contiguity from the first record of a contig :229-253, end == start+1 :256-259,
//                          clamp to 65535 :261-268), stores the two uint16 depths, reports contig starts.
//                          Contig boundaries and the "previous position" are local properties (a record starts a
//                          contig iff its name differs from the previous record's; the expected start is 1 after a
//                          contig start and previous start + 1 otherwise), so no sequential state is needed.
// The host side streams arbitrary pieces of the two files: unconsumed tail bytes and the last two consumed
// records (context for the local rules) are carried to the next piece.
#include <algorithm>
#include <string>

#include "common.hpp"
#include "scan.hpp"

namespace {

constexpr int TK_THREADS = 256;
constexpr int TK_TILE = TK_THREADS * 16;

__device__ __forceinline__ bool is_ws(uint32_t c) { return (c - 9u < 5u) | (c == 32u); }   // isspace() in the C locale

// bit i = byte i of the 16-byte piece at `pos` is white space; bytes at or beyond n count as white space
__device__ __forceinline__ uint32_t ws_mask16(const uint8_t *text, int64_t pos, int64_t n)
{
    uint32_t m = 0;
    if (pos + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + pos);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) m |= (uint32_t)is_ws((w[i >> 2] >> (8 * (i & 3))) & 0xFFu) << i;
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) m |= (uint32_t)(pos + i >= n || is_ws(text[pos + i])) << i;
    }
    return m;
}

__device__ __forceinline__ uint32_t tokstart_mask(const uint8_t *text, int64_t pos, int64_t n)
{
    if (pos >= n) return 0;
    const uint32_t ws = ws_mask16(text, pos, n);
    const uint32_t prev = pos == 0 ? 1u : (uint32_t)is_ws(text[pos - 1]);
    return ~ws & ((ws << 1) | prev) & 0xFFFFu;
}

__global__ __launch_bounds__(TK_THREADS) void tk_count(const uint8_t *text, int64_t n, uint32_t *tile_cnt)
{
    __shared__ uint32_t w[TK_THREADS / 64];
    const int64_t pos = ((int64_t)blockIdx.x * TK_THREADS + threadIdx.x) * 16;
    uint32_t c = (uint32_t)__popc(tokstart_mask(text, pos, n));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(TK_THREADS) void tk_scatter(const uint8_t *text, int64_t n, const uint32_t *tile_off, uint32_t *tok)
{
    __shared__ uint32_t wt[TK_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t pos = ((int64_t)blockIdx.x * TK_THREADS + t) * 16;
    uint32_t m = tokstart_mask(text, pos, n);
    const uint32_t c = (uint32_t)__popc(m);
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wt[wv] = inc;
    __syncthreads();
    uint32_t idx = tile_off[blockIdx.x] + inc - c;
    for (int i = 0; i < wv; ++i) idx += wt[i];
    while (m) {
        const int b = __ffs((int)m) - 1;
        m &= m - 1;
        tok[idx++] = (uint32_t)(pos + b);
    }
}

enum { BG_OK = 0, BG_COLUMNS_A = 1, BG_COLUMNS_B = 2, BG_ORDER = 3, BG_INCREMENTAL = 4, BG_ENDSTART = 5 };

struct BgRec {
    uint32_t name_off, name_len;
    int32_t st, end, depth;
    int32_t nfields;      // converted fields, like the return value of the reference's fscanf
};

// %d of one token: optional sign, digits up to the next white space; anything else fails the conversion
__device__ __forceinline__ bool parse_int(const uint8_t *text, int64_t n, uint32_t pos, int32_t *out)
{
    int64_t p = pos;
    bool neg = false;
    if (p < n && (text[p] == '-' || text[p] == '+')) neg = text[p++] == '-';
    if (p >= n || text[p] < '0' || text[p] > '9') return false;
    uint32_t v = 0;
    while (p < n && text[p] >= '0' && text[p] <= '9') v = v * 10u + (uint32_t)(text[p++] - '0');
    if (p < n && !is_ws(text[p])) return false;
    *out = (int32_t)(neg ? 0u - v : v);
    return true;
}

__device__ __forceinline__ BgRec parse_rec(const uint8_t *text, int64_t n, const uint32_t *tok, int64_t r)
{
    BgRec x;
    x.name_off = tok[4 * r];
    uint32_t e = x.name_off;
    while (e < n && !is_ws(text[e])) ++e;
    x.name_len = e - x.name_off;
    x.st = x.end = x.depth = 0;
    x.nfields = 1;
    if (parse_int(text, n, tok[4 * r + 1], &x.st)) {
        x.nfields = 2;
        if (parse_int(text, n, tok[4 * r + 2], &x.end)) {
            x.nfields = 3;
            if (parse_int(text, n, tok[4 * r + 3], &x.depth)) x.nfields = 4;
        }
    }
    return x;
}

__device__ __forceinline__ bool same_name(const uint8_t *a, uint32_t ao, uint32_t al, const uint8_t *b, uint32_t bo, uint32_t bl)
{
    if (al != bl) return false;
    for (uint32_t i = 0; i < al; ++i)
        if (a[ao + i] != b[bo + i]) return false;
    return true;
}

__device__ __forceinline__ uint32_t name_of(const uint8_t *text, int64_t n, const uint32_t *tok, int64_t r, uint32_t *len)
{
    const uint32_t o = tok[4 * r];
    uint32_t e = o;
    while (e < n && !is_ws(text[e])) ++e;
    *len = e - o;
    return o;
}

struct BgArgs {
    const uint8_t *ta, *tb;
    int64_t na, nb;
    const uint32_t *toka, *tokb;
    int64_t nrec;            // records available in both buffers
    int32_t skip;            // leading records that are context only (already consumed)
    int32_t file_start;      // record 0 of the buffer is the first record of the file (the context, at most two
                             // records, reaches back to it exactly when fewer than three records were consumed)
    int64_t base;            // global index of record `skip`
    uint16_t *da, *db;       // [capacity] global depth arrays
    unsigned long long *err; // min over records of (global index << 3 | kind), ~0 when clean; then a, b detail words
    int32_t *err_detail;     // [2 * 8]: per kind two ints
    uint32_t *n_break;
    uint4 *breaks;           // {global index lo, hi, name offset in ta, name length}
    uint32_t break_cap;
    unsigned long long *n_clamp;
    unsigned long long *neg_corr;   // [2]: what the stored uint16 of a NEGATIVE depth / mq depth is above the value itself (a multiple of 65536), summed
    // ... and one by one, for the contig each belongs to (cornetto_cov_shard() deals contigs to devices): {global index << 1 | file, the difference}; neg_n counts
    // every one, the list holds the first neg_cap (more than that in ONE call: the feed is refused — such a file is not coverage)
    unsigned long long *neg_list;
    uint32_t *neg_n;
    uint32_t neg_cap;
};

__global__ __launch_bounds__(256) void bg_records(BgArgs A)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= A.nrec || r < A.skip) return;
    const BgRec a = parse_rec(A.ta, A.na, A.toka, r);
    const BgRec b = parse_rec(A.tb, A.nb, A.tokb, r);
    const int64_t gi = A.base + (r - A.skip);
    int kind = BG_OK, d0 = 0, d1 = 0;
    if (a.nfields != 4) { kind = BG_COLUMNS_A; d0 = a.nfields; }                 // :209-212
    else if (b.nfields != 4) { kind = BG_COLUMNS_B; d0 = b.nfields; }            // :219-222
    else if (!same_name(A.ta, a.name_off, a.name_len, A.tb, b.name_off, b.name_len) || a.st != b.st || a.end != b.end)
        kind = BG_ORDER;                                                         // :224-227
    bool first = false;
    if (kind == BG_OK) {
        // a record starts a contig iff its name differs from the previous record's (:229)
        if (r == 0) {
            first = A.file_start != 0;      // (r == 0 && !file_start cannot reach here: context records are skipped)
        } else {
            uint32_t pl;
            const uint32_t po = name_of(A.ta, A.na, A.toka, r - 1, &pl);
            first = !same_name(A.ta, a.name_off, a.name_len, A.ta, po, pl);
            if (!first) {
                // expected start: 1 right after a contig start (prev_pos was reset to 0 there, :246), previous start + 1 otherwise
                bool prev_first;
                if (r - 1 == 0) {
                    prev_first = A.file_start != 0;   // r - 1 == 0 < skip only happens while record 0 is the file's first
                } else {
                    uint32_t ql;
                    const uint32_t qo = name_of(A.ta, A.na, A.toka, r - 2, &ql);
                    prev_first = !same_name(A.ta, po, pl, A.ta, qo, ql);
                }
                int32_t pst = 0;
                (void)parse_int(A.ta, A.na, A.toka[4 * (r - 1) + 1], &pst);
                const int32_t prev_pos = prev_first ? 0 : pst;
                if (prev_pos + 1 != a.st) { kind = BG_INCREMENTAL; d0 = prev_pos; d1 = a.st; }   // :249-252
            }
        }
    }
    if (kind == BG_OK && a.st + 1 != a.end) { kind = BG_ENDSTART; d0 = a.st; d1 = a.end; }      // :256-259
    if (kind != BG_OK) {
        const unsigned long long key = ((unsigned long long)gi << 3) | (unsigned)kind;
        const unsigned long long old = atomicMin(A.err, key);
        if (key < old) {   // best effort detail for the message; the smallest key wins the exit decision
            A.err_detail[2 * kind] = d0;
            A.err_detail[2 * kind + 1] = d1;
        }
        return;
    }
    int32_t da = a.depth, db = b.depth;
    unsigned clamp = 0;
    if (da > 65535) { da = 65535; ++clamp; }                                     // :261-268
    if (db > 65535) { db = 65535; ++clamp; }
    if (clamp) atomicAdd(A.n_clamp, (unsigned long long)clamp);
    A.da[gi] = (uint16_t)da;                                                     // negative values wrap as in the reference
    A.db[gi] = (uint16_t)db;
    // ... in the ARRAYS (:282-283); the reference's totals take the int itself (:285-286: tot_depth += depth1): the difference goes with the
    // coverage object, and cornetto_cov_prepare() takes it off the sums of the arrays
    if (da < 0) {
        const unsigned long long d = (unsigned long long)((long long)(uint16_t)da - (long long)da);
        atomicAdd(A.neg_corr, d);
        const uint32_t k = atomicAdd(A.neg_n, 1u);
        if (k < A.neg_cap) { A.neg_list[2 * k] = (unsigned long long)gi << 1; A.neg_list[2 * k + 1] = d; }
    }
    if (db < 0) {
        const unsigned long long d = (unsigned long long)((long long)(uint16_t)db - (long long)db);
        atomicAdd(A.neg_corr + 1, d);
        const uint32_t k = atomicAdd(A.neg_n, 1u);
        if (k < A.neg_cap) { A.neg_list[2 * k] = ((unsigned long long)gi << 1) | 1ull; A.neg_list[2 * k + 1] = d; }
    }
    if (first) {
        const uint32_t k = atomicAdd(A.n_break, 1u);
        if (k < A.break_cap) A.breaks[k] = make_uint4((uint32_t)(gi & 0xFFFFFFFFll), (uint32_t)(gi >> 32), a.name_off, a.name_len);
    }
}

// contig segments of the flat arrays -> 64-element aligned layout of cornetto_cov_t
__global__ void bg_layout(const uint16_t *src, const int64_t *src_off, const int64_t *dst_off, const int32_t *len, int32_t n_ctg,
                          uint16_t *dst)
{
    // grid.y is limited to 65535: contigs are taken with a grid stride (read-level coverage sets have more)
    for (int c = blockIdx.y; c < n_ctg; c += gridDim.y) {
        const int64_t n = len[c];
        const uint16_t *s = src + src_off[c];
        uint16_t *d = dst + dst_off[c];
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
    }
}

}  // namespace

struct cornetto_bgin {
    std::string pend[2];           // per file: context records (already consumed) + bytes not yet consumed
    int32_t ctx = 0;               // context records at the head of pend[]
    bool started = false;          // at least one record consumed
    int64_t n_rec = 0;             // records consumed so far
    uint16_t *d_a = nullptr, *d_b = nullptr;
    int64_t cap = 0;
    struct Brk { int64_t index; std::string name; };
    std::vector<Brk> breaks;
    unsigned long long n_clamp = 0;
    unsigned long long neg_corr[2] = {0, 0};   // (see BgArgs::neg_corr)
    std::vector<unsigned long long> neg;       // {record index << 1 | file, difference} of every negative value (BgArgs::neg_list)
    cornetto_bgerr_t err{0, 0, 0, 0};
    bool finished = false;
    int64_t left_mq = 0;      // tokens of cov-mq behind the last record when cov-total ended (the reference never looks at them: :204-207)
    // cornetto_bgin_prefetch(): the pieces of the feeds to come on their way to the device while this feed's kernels run — two staging sets: one
    // is being filled while the feed in front of it still copies out of the other
    hipStream_t pf_stream = nullptr;
    struct Staged {
        hipEvent_t ev = nullptr;
        uint8_t *buf[2] = {nullptr, nullptr};    // on the device, per file
        size_t cap[2] = {0, 0};
        const char *src[2] = {nullptr, nullptr}; // what is staged: the caller's pointers and lengths (a feed with the same ones takes it from here)
        int64_t n[2] = {0, 0};
        bool valid = false;
        uint64_t seq = 0;
    } pf[2];
    uint64_t pf_seq = 0;
};

namespace {

int tokenize(cornetto_accel_t *h, const uint8_t *d_text, int64_t n, int slot_tok, int slot_cnt, uint32_t **tok_out, int64_t *ntok)
{
    *ntok = 0;
    *tok_out = nullptr;
    if (n <= 0) return CORNETTO_OK;
    const int64_t nt = (n + TK_TILE - 1) / TK_TILE;
    uint32_t *d_cnt = (uint32_t *)cn_ws(h, slot_cnt, ((size_t)2 * nt + (nt + 4095) / 4096 + 8) * 4 + 16);
    unsigned long long *p_tot = (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
    if (!d_cnt || !p_tot) return cn_fail(h, CORNETTO_E_NOMEM, "bedgraph ingest: workspace allocation failed");
    uint32_t *d_off = d_cnt + nt, *d_part = d_off + nt;
    unsigned long long *d_tot = reinterpret_cast<unsigned long long *>(((uintptr_t)(d_part + (nt + 4095) / 4096 + 1) + 7) & ~(uintptr_t)7);
    CN_LAUNCH(h, "tk_count", tk_count<<<dim3((unsigned)nt), dim3(TK_THREADS), 0, h->stream>>>(d_text, n, d_cnt));
    CN_TRY(cnscan::exclusive_u32(h, "tk_scan", d_cnt, nt, 1, d_off, d_part, d_tot));
    CN_HIP(h, hipMemcpyAsync(p_tot, d_tot, 8, hipMemcpyDeviceToHost, h->stream));
    CN_HIP(h, hipStreamSynchronize(h->stream));
    const int64_t total = (int64_t)p_tot[0];
    uint32_t *d_tok = (uint32_t *)cn_ws(h, slot_tok, ((size_t)total + 8) * 4);
    if (!d_tok) return cn_fail(h, CORNETTO_E_NOMEM, "bedgraph ingest: workspace allocation failed");
    if (total) CN_LAUNCH(h, "tk_scatter", tk_scatter<<<dim3((unsigned)nt), dim3(TK_THREADS), 0, h->stream>>>(d_text, n, d_off, d_tok));
    *tok_out = d_tok;
    *ntok = total;
    return CORNETTO_OK;
}

}  // namespace

extern "C" {

void *cornetto_pinned_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void cornetto_pinned_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int cornetto_bgin_open(cornetto_accel_t *h, cornetto_bgin_t **out)
{
    if (!h || !out) return cn_fail(h, CORNETTO_E_ARG, "bgin_open: bad argument");
    *out = new (std::nothrow) cornetto_bgin;
    return *out ? CORNETTO_OK : cn_fail(h, CORNETTO_E_NOMEM, "bgin_open: host allocation failed");
}

void cornetto_bgin_close(cornetto_accel_t *h, cornetto_bgin_t *b)
{
    if (!b) return;
    if (h) (void)hipSetDevice(h->device);
    if (b->d_a) (void)hipFree(b->d_a);
    if (b->d_b) (void)hipFree(b->d_b);
    if (b->pf_stream) { (void)hipStreamSynchronize(b->pf_stream); (void)hipStreamDestroy(b->pf_stream); }
    for (auto &st : b->pf) {
        if (st.ev) (void)hipEventDestroy(st.ev);
        for (int f = 0; f < 2; ++f)
            if (st.buf[f]) (void)hipFree(st.buf[f]);
    }
    delete b;
}

int cornetto_bgin_prefetch(cornetto_accel_t *h, cornetto_bgin_t *b, const char *tot, int64_t n_tot, const char *mq, int64_t n_mq)
{
    if (!h || !b || n_tot < 0 || n_mq < 0 || (n_tot > 0 && !tot) || (n_mq > 0 && !mq)) return cn_fail(h, CORNETTO_E_ARG, "bgin_prefetch: bad argument");
    if (b->finished || b->err.kind) return CORNETTO_OK;
    CN_HIP(h, hipSetDevice(h->device));
    if (!b->pf_stream) {
        CN_HIP(h, hipStreamCreateWithFlags(&b->pf_stream, hipStreamNonBlocking));
        for (auto &st : b->pf) CN_HIP(h, hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    }
    // the set that is free, else the older one (a prefetch nobody fed: it is written over once its own copy and whatever reads it are through)
    cornetto_bgin::Staged *st = !b->pf[0].valid ? &b->pf[0] : !b->pf[1].valid ? &b->pf[1] : (b->pf[0].seq < b->pf[1].seq ? &b->pf[0] : &b->pf[1]);
    if (st->valid) CN_HIP(h, hipStreamSynchronize(b->pf_stream));
    st->valid = false;
    // (the feed that took this set last copied out of it on the handle's stream and synchronised that stream before it returned: nothing reads it now)
    const char *src[2] = {tot, mq};
    const int64_t nn[2] = {n_tot, n_mq};
    for (int f = 0; f < 2; ++f) {
        if ((size_t)nn[f] > st->cap[f]) {
            if (st->buf[f]) (void)hipFree(st->buf[f]);
            st->buf[f] = nullptr;
            st->cap[f] = 0;
            const size_t cap = ((size_t)nn[f] + (1u << 20)) & ~(size_t)4095;
            if (hipMalloc((void **)&st->buf[f], cap) != hipSuccess) return cn_fail(h, CORNETTO_E_NOMEM, "bgin_prefetch: device allocation of %zu bytes failed", cap);
            st->cap[f] = cap;
        }
        if (nn[f]) CN_HIP(h, hipMemcpyAsync(st->buf[f], src[f], (size_t)nn[f], hipMemcpyHostToDevice, b->pf_stream));
        st->src[f] = src[f];
        st->n[f] = nn[f];
    }
    CN_HIP(h, hipEventRecord(st->ev, b->pf_stream));
    st->valid = true;
    st->seq = ++b->pf_seq;
    return CORNETTO_OK;
}

void cornetto_bgin_pending(const cornetto_bgin_t *b, int64_t *pend_tot, int64_t *pend_mq)
{
    if (pend_tot) *pend_tot = b ? (int64_t)b->pend[0].size() : 0;
    if (pend_mq) *pend_mq = b ? (int64_t)b->pend[1].size() : 0;
}

int64_t cornetto_bgin_unmatched_mq(const cornetto_bgin_t *b) { return b && b->finished ? b->left_mq : 0; }

const cornetto_bgerr_t *cornetto_bgin_error(const cornetto_bgin_t *b) { return b ? &b->err : nullptr; }

int cornetto_bgin_done(const cornetto_bgin_t *b) { return b && b->finished; }

int cornetto_bgin_feed(cornetto_accel_t *h, cornetto_bgin_t *b, const char *tot, int64_t n_tot, const char *mq, int64_t n_mq, int final)
{
    if (!h || !b || n_tot < 0 || n_mq < 0 || (n_tot > 0 && !tot) || (n_mq > 0 && !mq)) return cn_fail(h, CORNETTO_E_ARG, "bgin_feed: bad argument");
    if (b->finished) return CORNETTO_OK;   // cov-total is exhausted: whatever follows in cov-mq is ignored, as in the reference
    if (b->err.kind) return CORNETTO_E_FORMAT;
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    const char *src[2] = {tot, mq};
    const int64_t nnew[2] = {n_tot, n_mq};
    int64_t n[2];
    uint8_t *d_text[2];
    for (int f = 0; f < 2; ++f) {
        n[f] = (int64_t)b->pend[f].size() + nnew[f];
        if (n[f] > 0xF0000000ll) return cn_fail(h, CORNETTO_E_ARG, "bgin_feed: more than 3.75 GiB pending for one file; feed smaller pieces");
        d_text[f] = (uint8_t *)cn_ws(h, f ? WS_BG_TEXT_B : WS_BG_TEXT_A, (size_t)n[f] + 64);
        if (!d_text[f]) return cn_fail(h, CORNETTO_E_NOMEM, "bgin_feed: workspace allocation failed");
        if (!b->pend[f].empty()) CN_HIP(h, hipMemcpyAsync(d_text[f], b->pend[f].data(), b->pend[f].size(), hipMemcpyHostToDevice, h->stream));
    }
    // the new bytes: staged on the device by cornetto_bgin_prefetch() (the same pointers and lengths) — a copy inside the device behind what is
    // left over from the last feed — or from the host now
    cornetto_bgin::Staged *st = nullptr;
    for (auto &x : b->pf)
        if (x.valid && x.src[0] == tot && x.src[1] == mq && x.n[0] == n_tot && x.n[1] == n_mq) st = &x;
    if (st) {
        CN_HIP(h, hipStreamWaitEvent(h->stream, st->ev, 0));
        st->valid = false;
    }
    for (int f = 0; f < 2; ++f) {
        if (!nnew[f]) continue;
        if (st) CN_HIP(h, hipMemcpyAsync(d_text[f] + b->pend[f].size(), st->buf[f], (size_t)nnew[f], hipMemcpyDeviceToDevice, h->stream));
        else CN_HIP(h, hipMemcpyAsync(d_text[f] + b->pend[f].size(), src[f], (size_t)nnew[f], hipMemcpyHostToDevice, h->stream));
    }
    // a token cut by the end of the buffer is not complete yet (unless this is the end of the file)
    auto last_byte = [&](int f) -> int {
        if (n[f] == 0) return ' ';
        return nnew[f] ? (unsigned char)src[f][nnew[f] - 1] : (unsigned char)b->pend[f].back();
    };
    CN_TRACE("bgin_feed: copies queued");
    uint32_t *d_tok[2];
    int64_t ntok[2], all_tok[2];
    const bool eof[2] = {(final & 1) != 0, (final & 2) != 0};
    for (int f = 0; f < 2; ++f) {
        CN_TRY(tokenize(h, d_text[f], n[f], f ? WS_BG_TOK_B : WS_BG_TOK_A, f ? WS_BG_CNT_B : WS_BG_CNT_A, &d_tok[f], &ntok[f]));
        const int lb = last_byte(f);
        const bool open_token = !((unsigned)(lb - 9) < 5u || lb == 32);
        all_tok[f] = ntok[f];
        if (!eof[f] && open_token && ntok[f] > 0) --ntok[f];   // the last token may continue in the next piece
    }
    CN_TRACE("bgin_feed: tokenised");
    int64_t nrec = std::min(ntok[0] / 4, ntok[1] / 4);
    const int64_t fresh = nrec - b->ctx;   // records consumed by this call
    // small device block: err key, n_break, n_clamp, details
    unsigned long long *d_small = (unsigned long long *)cn_ws(h, WS_BG_SMALL, 256);
    unsigned long long *p_small = (unsigned long long *)cn_pin(h, PIN_SMALL, 256);
    if (!d_small || !p_small) return cn_fail(h, CORNETTO_E_NOMEM, "bgin_feed: workspace allocation failed");
    uint32_t break_cap = 1u << 16;
    std::vector<uint4> brk;
    if (fresh > 0) {
        if (b->n_rec + fresh > b->cap) {   // grow the flat depth arrays (amortised doubling)
            int64_t ncap = std::max<int64_t>(b->cap * 2, b->n_rec + fresh + (1 << 20));
            uint16_t *na = nullptr, *nb = nullptr;
            if (hipMalloc((void **)&na, (size_t)ncap * 2) != hipSuccess || hipMalloc((void **)&nb, (size_t)ncap * 2) != hipSuccess)
                return cn_fail(h, CORNETTO_E_NOMEM, "bgin_feed: cannot grow the depth arrays to %lld positions", (long long)ncap);
            if (b->n_rec) {
                CN_HIP(h, hipMemcpyAsync(na, b->d_a, (size_t)b->n_rec * 2, hipMemcpyDeviceToDevice, h->stream));
                CN_HIP(h, hipMemcpyAsync(nb, b->d_b, (size_t)b->n_rec * 2, hipMemcpyDeviceToDevice, h->stream));
                CN_HIP(h, hipStreamSynchronize(h->stream));
            }
            if (b->d_a) (void)hipFree(b->d_a);
            if (b->d_b) (void)hipFree(b->d_b);
            b->d_a = na;
            b->d_b = nb;
            b->cap = ncap;
        }
        for (int attempt = 0; attempt < 2; ++attempt) {
            // (the contig-start list, and behind it the list of negative values: 16 bytes each)
            constexpr uint32_t NEG_CAP = 1u << 16;
            uint4 *d_brk = (uint4 *)cn_ws(h, WS_BG_BRK, ((size_t)break_cap + NEG_CAP) * sizeof(uint4));
            if (!d_brk) return cn_fail(h, CORNETTO_E_NOMEM, "bgin_feed: workspace allocation failed");
            unsigned long long *d_neg = reinterpret_cast<unsigned long long *>(d_brk + break_cap);
            CN_HIP(h, hipMemsetAsync(d_small, 0, 256, h->stream));
            CN_HIP(h, hipMemsetAsync(d_small, 0xFF, 8, h->stream));
            BgArgs A{d_text[0], d_text[1], n[0], n[1], d_tok[0], d_tok[1], nrec, b->ctx, (b->n_rec - b->ctx == 0) ? 1 : 0, b->n_rec, b->d_a, b->d_b,
                     d_small, reinterpret_cast<int32_t *>(d_small + 4), reinterpret_cast<uint32_t *>(d_small + 1), d_brk, break_cap, d_small + 2, d_small + 14,
                     d_neg, reinterpret_cast<uint32_t *>(d_small + 16), NEG_CAP};
            CN_LAUNCH(h, "bg_records", bg_records<<<dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, h->stream>>>(A));
            CN_HIP(h, hipMemcpyAsync(p_small, d_small, 256, hipMemcpyDeviceToHost, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));
            const uint32_t nb = (uint32_t)(p_small[1] & 0xFFFFFFFFull);
            if (nb > break_cap) {   // more contig starts than room: exact rerun (the kernel is idempotent)
                if (attempt == 1) return cn_fail(h, CORNETTO_E_HIP, "bgin_feed: contig-start list overflow");
                break_cap = nb;
                continue;
            }
            brk.resize(nb);
            if (nb) CN_HIP(h, hipMemcpy(brk.data(), d_brk, (size_t)nb * sizeof(uint4), hipMemcpyDeviceToHost));
            const uint32_t nneg = (uint32_t)(p_small[16] & 0xFFFFFFFFull);
            if (nneg > NEG_CAP && p_small[0] == ~0ull)
                return cn_fail(h, CORNETTO_E_UNSUPPORTED, "bgin_feed: %u negative depth values in one piece of text (at most %u are kept apart for the totals)", nneg, NEG_CAP);
            if (nneg && p_small[0] == ~0ull) {
                const size_t at = b->neg.size();
                b->neg.resize(at + 2 * (size_t)nneg);
                CN_HIP(h, hipMemcpy(b->neg.data() + at, d_neg, (size_t)nneg * 16, hipMemcpyDeviceToHost));
            }
            break;
        }
        if (p_small[0] != ~0ull) {   // the record with the smallest index that fails a check decides (the reference stops there)
            const int kind = (int)(p_small[0] & 7);
            const int32_t *det = reinterpret_cast<const int32_t *>(p_small + 4);
            b->err.kind = kind;
            b->err.record = (int64_t)(p_small[0] >> 3);
            b->err.a = det[2 * kind];
            b->err.b = det[2 * kind + 1];
            cn_timing_end(h);
            return CORNETTO_E_FORMAT;
        }
        b->n_clamp += p_small[2];
        b->neg_corr[0] += p_small[14];
        b->neg_corr[1] += p_small[15];
    }
    CN_TRACE("bgin_feed: records");
    // offsets needed for the carry: start of the first unconsumed token, start of the context (last two consumed records)
    int64_t cut[2], ctx_start[2];
    const int new_ctx = (int)std::min<int64_t>(2, nrec);
    for (int f = 0; f < 2; ++f) {
        uint32_t v[2] = {0, 0};
        // token index of the first unconsumed token / first context token
        const int64_t t_cut = 4 * nrec, t_ctx = 4 * (nrec - new_ctx);
        if (t_cut < all_tok[f]) CN_HIP(h, hipMemcpy(&v[0], d_tok[f] + t_cut, 4, hipMemcpyDeviceToHost));
        if (nrec > 0) CN_HIP(h, hipMemcpy(&v[1], d_tok[f] + t_ctx, 4, hipMemcpyDeviceToHost));
        cut[f] = t_cut < all_tok[f] ? (int64_t)v[0] : n[f];
        ctx_start[f] = nrec > 0 ? (int64_t)v[1] : cut[f];
    }
    // contig names: copy from the host bytes of file 0 before they go away
    auto byte_at = [&](int f, int64_t off, int64_t len) -> std::string {   // bytes [off, off+len) of (pend[f] | new data)
        std::string s;
        if (len <= 0) return s;
        s.reserve((size_t)len);
        const int64_t np = (int64_t)b->pend[f].size();
        if (off < np) s.append(b->pend[f], (size_t)off, (size_t)std::min<int64_t>(len, np - off));
        if (off + len > np) {
            const int64_t o2 = std::max<int64_t>(off, np) - np;
            s.append(src[f] + o2, (size_t)(off + len - np - o2));
        }
        return s;
    };
    for (const uint4 &x : brk) {
        cornetto_bgin::Brk k;
        k.index = (int64_t)x.x | ((int64_t)x.y << 32);
        k.name = byte_at(0, x.z, x.w);
        b->breaks.push_back(std::move(k));
    }
    {
        // end-of-file logic of the reference's loop (:204-222): it stops cleanly when cov-total is exhausted; a
        // cov-total record without a cov-mq partner is "not in the same order" (or a short cov-mq record)
        const int64_t complete_a = ntok[0] / 4, left_a = ntok[0] - 4 * complete_a, left_b = ntok[1] - 4 * nrec;
        const int64_t at = b->n_rec + std::max<int64_t>(0, fresh);
        if (eof[1] && complete_a > nrec) {
            b->err = cornetto_bgerr_t{(left_b >= 1 && left_b <= 3) ? BG_COLUMNS_B : BG_ORDER, at, (int32_t)left_b, 0};
            cn_timing_end(h);
            return CORNETTO_E_FORMAT;
        }
        if (eof[0] && complete_a == nrec) {
            if (left_a >= 1) {
                // a trailing partial record of cov-total: fscanf converts 1..3 fields, then cov-mq must still deliver
                b->err = cornetto_bgerr_t{BG_COLUMNS_A, at, (int32_t)left_a, 0};
                cn_timing_end(h);
                return CORNETTO_E_FORMAT;
            }
            b->finished = true;
            b->left_mq = left_b > 0 ? left_b : 0;
        }
    }
    // carry: context records + everything not consumed
    for (int f = 0; f < 2; ++f) {
        const int64_t from = nrec > 0 ? ctx_start[f] : (b->ctx ? 0 : cut[f]);
        std::string np = byte_at(f, from, n[f] - from);
        b->pend[f].swap(np);
    }
    if (fresh > 0) {
        b->n_rec += fresh;
        b->started = true;
        b->ctx = new_ctx;
    }
    CN_TRACE("bgin_feed: carry");
    cn_timing_end(h);
    return CORNETTO_OK;
}

int cornetto_bgin_finish(cornetto_accel_t *h, cornetto_bgin_t *b, cornetto_cov_t **cov, int32_t *n_ctg, char ***names, int64_t *n_clamped)
{
    if (!h || !b || !cov || !n_ctg || !names) return cn_fail(h, CORNETTO_E_ARG, "bgin_finish: bad argument");
    if (!b->finished) return cn_fail(h, CORNETTO_E_ARG, "bgin_finish: feed(..., final = 1) has not succeeded");
    CN_HIP(h, hipSetDevice(h->device));
    *cov = nullptr;
    *n_ctg = 0;
    *names = nullptr;
    if (n_clamped) *n_clamped = (int64_t)b->n_clamp;
    std::sort(b->breaks.begin(), b->breaks.end(), [](const cornetto_bgin::Brk &x, const cornetto_bgin::Brk &y) { return x.index < y.index; });
    const int32_t nc = (int32_t)b->breaks.size();
    cornetto_cov_t *c = new (std::nothrow) cornetto_cov;
    if (!c) return cn_fail(h, CORNETTO_E_NOMEM, "bgin_finish: host allocation failed");
    c->n = nc;
    c->sum_corr[0] = b->neg_corr[0];
    c->sum_corr[1] = b->neg_corr[1];
    if (!b->neg.empty()) {                         // ... and per contig (the breaks are sorted by record index): cornetto_cov_shard()
        c->ctg_corr.assign(2 * (size_t)nc, 0ull);
        for (size_t i = 0; i + 1 < b->neg.size(); i += 2) {
            const int64_t gi = (int64_t)(b->neg[i] >> 1);
            size_t lo = 0, hi = (size_t)nc;           // the last contig whose first record is <= gi
            while (lo + 1 < hi) {
                const size_t mid = (lo + hi) / 2;
                if (b->breaks[mid].index <= gi) lo = mid; else hi = mid;
            }
            if (nc) c->ctg_corr[2 * lo + (b->neg[i] & 1ull)] += b->neg[i + 1];
        }
    }
    std::vector<int64_t> src_off(nc);
    int64_t pos = 0;
    for (int32_t i = 0; i < nc; ++i) {
        const int64_t s = b->breaks[i].index, e = i + 1 < nc ? b->breaks[i + 1].index : b->n_rec;
        if (e - s > INT32_MAX) { delete c; return cn_fail(h, CORNETTO_E_UNSUPPORTED, "bgin_finish: contig %d has more than 2^31-1 positions", i); }
        src_off[i] = s;
        c->off.push_back(pos);
        c->len.push_back((int32_t)(e - s));
        c->total += e - s;
        pos = cn_align_up(pos + (e - s), 64);
    }
    const size_t bytes = (size_t)(pos + 256) * sizeof(uint16_t);
    size_t ntab = (size_t)(nc > 0 ? nc : 1);
    int64_t *d_src = nullptr;
    if (hipMalloc(&c->owned_d, bytes) != hipSuccess || hipMalloc(&c->owned_q, bytes) != hipSuccess ||
        hipMalloc((void **)&c->d_off, ntab * 8) != hipSuccess || hipMalloc((void **)&c->d_len, ntab * 4) != hipSuccess ||
        hipMalloc((void **)&d_src, ntab * 8) != hipSuccess) {
        cornetto_cov_free(h, c);
        return cn_fail(h, CORNETTO_E_NOMEM, "bgin_finish: device allocation failed");
    }
    c->d_depth = (const uint16_t *)c->owned_d;
    c->d_mq = (const uint16_t *)c->owned_q;
    hipError_t e = hipMemsetAsync(c->owned_d, 0, bytes, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->owned_q, 0, bytes, h->stream);
    if (nc) {
        if (e == hipSuccess) e = hipMemcpyAsync(c->d_off, c->off.data(), (size_t)nc * 8, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->d_len, c->len.data(), (size_t)nc * 4, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_src, src_off.data(), (size_t)nc * 8, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            bg_layout<<<dim3(64, (unsigned)std::min<int32_t>(nc, 32768)), dim3(256), 0, h->stream>>>(b->d_a, d_src, c->d_off, c->d_len, nc, (uint16_t *)c->owned_d);
            bg_layout<<<dim3(64, (unsigned)std::min<int32_t>(nc, 32768)), dim3(256), 0, h->stream>>>(b->d_b, d_src, c->d_off, c->d_len, nc, (uint16_t *)c->owned_q);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_src);
    if (e != hipSuccess) {
        cornetto_cov_free(h, c);
        return cn_fail(h, CORNETTO_E_HIP, "bgin_finish: %s", hipGetErrorString(e));
    }
    // the flat arrays are no longer needed
    (void)hipFree(b->d_a);
    (void)hipFree(b->d_b);
    b->d_a = b->d_b = nullptr;
    b->cap = 0;
    char **nm = (char **)malloc(ntab * sizeof(char *));
    if (!nm) { cornetto_cov_free(h, c); return cn_fail(h, CORNETTO_E_NOMEM, "bgin_finish: host allocation failed"); }
    for (int32_t i = 0; i < nc; ++i) {
        nm[i] = (char *)malloc(b->breaks[i].name.size() + 1);
        if (nm[i]) memcpy(nm[i], b->breaks[i].name.c_str(), b->breaks[i].name.size() + 1);
    }
    *cov = c;
    *n_ctg = nc;
    *names = nm;
    return CORNETTO_OK;
}

}  // extern "C"

// fastq.hip — FASTQ record splitter on the device (`cornetto seq -m` + per-read sdust,
// docs/protocol.md:185, reader src/kseq.h:184-224).
//
// The reference frames records with klib's kseq, one byte-stream state machine on one core (~0.5 Gbases/s).  ONT /
// PacBio FASTQ is written as plain four-line records, and for such a record the state machine reduces to local rules
// on the four lines — so the text is indexed in parallel instead:
//   fq_nl_count / fq_nl_scatter   newline positions of the piece (16 bytes per thread, ballot-free bit tricks,
//                                 per-tile counts -> scan -> ordered scatter), ~1 B/byte read, HBM bound
//   fq_records                    one thread per four lines: the checks under which kseq_read (src/kseq.h:184-224)
//                                 reads exactly these four lines as one record, the name / comment split of
//                                 ks_getuntil(KS_SEP_SPACE) (:195-196), the '\r' rule of :138, the `seq -m` length
//                                 test (src/seq.c:120); the first group that is not a plain record is reported and
//                                 everything from there on is left to the caller's sequential reader
//   fq_pack                       bases of the kept reads -> the 64-byte aligned layout of cornetto_asm_t, so that
//                                 sdust / telofind run on them without a host-side copy per read
// Nothing here guesses: a piece either is a sequence of plain records (then the result is what kseq returns, record
// for record — tests/test_gpu_fastq.py against the oracle's kseq restatement) or the caller is told where it stops
// being one.
#include "common.hpp"
#include "scan.hpp"

namespace {

constexpr int FQ_THREADS = 256;
constexpr int FQ_TILE = FQ_THREADS * 16;

// bit i = byte i of the 16-byte piece at `pos` is '\n'
__device__ __forceinline__ uint32_t nl_mask16(const uint8_t *text, int64_t pos, int64_t n)
{
    uint32_t m = 0;
    if (pos + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + pos);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t x = w[k] ^ 0x0A0A0A0Au;                       // zero byte <=> newline
            const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);   // 0x80 in every zero byte (exact)
            m |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * k);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) m |= (uint32_t)(pos + i < n && text[pos + i] == '\n') << i;
    }
    return m;
}

__global__ __launch_bounds__(FQ_THREADS) void fq_nl_count(const uint8_t *text, int64_t n, uint32_t *tile_cnt)
{
    __shared__ uint32_t w[FQ_THREADS / 64];
    const int64_t pos = ((int64_t)blockIdx.x * FQ_THREADS + threadIdx.x) * 16;
    uint32_t c = pos < n ? (uint32_t)__popc(nl_mask16(text, pos, n)) : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(FQ_THREADS) void fq_nl_scatter(const uint8_t *text, int64_t n, const uint32_t *tile_off, uint32_t *nl)
{
    __shared__ uint32_t wt[FQ_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t pos = ((int64_t)blockIdx.x * FQ_THREADS + t) * 16;
    uint32_t m = pos < n ? nl_mask16(text, pos, n) : 0u;
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
        nl[idx++] = (uint32_t)(pos + b);
    }
}

__device__ __forceinline__ bool fq_space(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }   // isspace(), C locale

// Line k of the piece is [k == 0 ? 0 : nl[k-1] + 1, nl[k]); with `virt` the text ends inside its last line and that line's
// end is n itself (kseq treats the end of the input as the end of the line, src/kseq.h:127-131).
struct FqArgs {
    const uint8_t *text;
    int64_t n;
    const uint32_t *nl;
    int64_t n_nl;      // real newlines
    int64_t n_rec;     // groups of four lines to look at
    int32_t min_len;
    cornetto_fqrec_t *recs;
    uint32_t *first_bad;   // smallest group that is not a plain record
    uint32_t *ends;        // [n_rec] offset just behind the group
};

__global__ __launch_bounds__(256) void fq_records(FqArgs A)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= A.n_rec) return;
    const uint8_t *t = A.text;
    int64_t e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = 4 * r + k < A.n_nl ? (int64_t)A.nl[4 * r + k] : A.n;
    const int64_t s0 = r == 0 ? 0 : (int64_t)A.nl[4 * r - 1] + 1, s1 = e[0] + 1, s2 = e[1] + 1, s3 = e[2] + 1;
    bool ok = t[s0] == '@';                                            // kseq_read :189-193 takes the next '@' or '>' anywhere: it must be right here
    // :201-205 the sequence ends at a line that begins with '>', '+' or '@'; an empty line is skipped
    if (e[1] > s1) ok = ok && t[s1] != '>' && t[s1] != '+' && t[s1] != '@';
    ok = ok && e[2] > s2 && t[s2] == '+';                              // one sequence line only, then the separator
    int64_t l1 = e[1] - s1, l3 = e[3] - s3;
    if (l1 > 1 && t[e[1] - 1] == '\r') --l1;                           // :138, for the sequence as a whole
    if (l3 > 1 && t[e[3] - 1] == '\r') --l3;
    ok = ok && l1 == l3 && l1 <= 0x7fffffffLL;                         // :221-223: the first quality line must complete the record
    // name / comment, :195-196
    int64_t j = s0 + 1;
    while (j < e[0] && !fq_space(t[j])) ++j;
    int64_t cl = j < e[0] ? e[0] - (j + 1) : 0;
    if (cl > 1 && t[e[0] - 1] == '\r') --cl;
    ok = ok && (j - (s0 + 1)) <= 0x7fffffffLL && cl <= 0x7fffffffLL;
    cornetto_fqrec_t rec;
    rec.head = s0;
    rec.seq = s1;
    rec.qual = s3;
    rec.len = (int32_t)l1;
    rec.name_len = (int32_t)(j - (s0 + 1));
    rec.comment_len = (int32_t)cl;
    rec.keep = l1 >= A.min_len ? 1 : 0;
    A.recs[r] = rec;
    A.ends[r] = (uint32_t)(e[3] < A.n ? e[3] + 1 : A.n);
    if (!ok) atomicMin(A.first_bad, (uint32_t)r);
}

typedef uint32_t __attribute__((aligned(1))) fq_u32u;

// one workgroup per read (grid stride): bases of read i -> bases[off[i] .. off[i] + len[i])
__global__ __launch_bounds__(256) void fq_pack(const uint8_t *text, const int64_t *src, const int64_t *off, const int32_t *len, int32_t n_reads,
                                               uint8_t *bases)
{
    for (int32_t i = blockIdx.x; i < n_reads; i += gridDim.x) {
        const uint8_t *s = text + src[i];
        uint8_t *d = bases + off[i];
        const int32_t L = len[i];
        const int32_t L16 = L & ~15;
        for (int32_t k = threadIdx.x * 16; k < L16; k += 256 * 16) {
            const fq_u32u *p = reinterpret_cast<const fq_u32u *>(s + k);
            uint4 v;
            v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
            *reinterpret_cast<uint4 *>(d + k) = v;
        }
        for (int32_t k = L16 + threadIdx.x; k < L; k += 256) d[k] = s[k];
    }
}


// ---- FASTA (kseq_read in front of telofind / sdust / fa2bed) -----------------------------
// A FASTA piece is plain when no line begins with '@' or '+' (kseq would switch to its FASTQ reading there,
// src/kseq.h:201,:213): then a record is a '>' line and every following line up to the next '>' line, empty lines count
// nothing (:202) and a line's trailing '\r' is dropped (:138).  Per line: header flag and payload length; two scans give
// every line its record and its place in the record's sequence; one pass over the text moves the payload bytes into the
// resident layout.
struct FaLines {
    const uint8_t *text;
    int64_t n;
    const uint32_t *nl;
    int64_t n_nl, n_lines;
    uint32_t *hdr;        // [n_lines + 1] 1 = the line begins with '>'
    uint32_t *pay;        // [n_lines + 1] bytes of the line that are sequence
    uint32_t *bad_line;   // smallest line at which the piece stops being plain FASTA
};

__device__ __forceinline__ int64_t fa_line_start(const uint32_t *nl, int64_t k) { return k == 0 ? 0 : (int64_t)nl[k - 1] + 1; }

__global__ __launch_bounds__(256) void fa_lines(FaLines A)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k > A.n_lines) return;
    uint32_t hdr = 0, pay = 0;
    if (k < A.n_lines) {
        const int64_t s = fa_line_start(A.nl, k), e = k < A.n_nl ? (int64_t)A.nl[k] : A.n;
        const int64_t len = e - s;
        if (len > 0) {
            const uint8_t c = A.text[s];
            if (c == '>') hdr = 1;
            else if (c == '@' || c == '+') atomicMin(A.bad_line, (uint32_t)k);
            else pay = (uint32_t)(len - (A.text[e - 1] == '\r' ? 1 : 0));
        }
    }
    A.hdr[k] = hdr;      // entry n_lines = 0: the exclusive scans then end with the totals
    A.pay[k] = pay;
}

// header line of every record; a line that is exactly "\r" while its record holds nothing yet would be kept as a base
// by kseq (:138 drops the '\r' only when more than one byte is held): not plain
__global__ __launch_bounds__(256) void fa_heads(FaLines A, const uint32_t *H, const uint32_t *P, uint32_t *hdr_line, int pass)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= A.n_lines) return;
    if (pass == 0) {
        if (A.hdr[k]) hdr_line[H[k]] = (uint32_t)k;
        return;
    }
    if (A.hdr[k] || A.pay[k] != 0 || H[k] == 0) return;
    const int64_t s = fa_line_start(A.nl, k), e = k < A.n_nl ? (int64_t)A.nl[k] : A.n;
    if (e - s == 1 && A.text[s] == '\r' && P[k] == P[hdr_line[H[k] - 1]]) atomicMin(A.bad_line, (uint32_t)k);
}

__global__ __launch_bounds__(256) void fa_records(FaLines A, const uint32_t *P, const uint32_t *hdr_line, int64_t n_rec, cornetto_farec_t *recs)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rec) return;
    const int64_t k = hdr_line[r], next = r + 1 < n_rec ? (int64_t)hdr_line[r + 1] : A.n_lines;
    const int64_t s = fa_line_start(A.nl, k), e = k < A.n_nl ? (int64_t)A.nl[k] : A.n;
    int64_t j = s + 1;
    while (j < e && !fq_space(A.text[j])) ++j;
    cornetto_farec_t rec;
    rec.head = s;
    rec.len = (int64_t)P[next] - (int64_t)P[k];
    rec.name_len = (int32_t)((j - (s + 1)) > 0x7fffffffLL ? 0x7fffffff : (j - (s + 1)));
    rec.pad = 0;
    recs[r] = rec;
}

// destination of every sequence line of the records that are used: offset of its first payload byte in the resident
// layout, or -1 (header lines, lines of records that are left to the caller)
__global__ __launch_bounds__(256) void fa_linedst(FaLines A, const uint32_t *H, const uint32_t *P, const uint32_t *hdr_line, const int64_t *rec_off,
                                                  int64_t n_use, int64_t *dst)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= A.n_lines) return;
    int64_t d = -1;
    if (!A.hdr[k] && A.pay[k] && H[k] >= 1 && (int64_t)H[k] - 1 < n_use) {
        const int64_t r = (int64_t)H[k] - 1;
        d = rec_off[r] + ((int64_t)P[k] - (int64_t)P[hdr_line[r]]);
    }
    dst[k] = d;
}

// one pass over the text, 16 bytes per thread: the line of the first byte is the number of newlines in front of it
// (per-tile offsets of the newline index + an in-tile scan, exactly as fq_nl_scatter finds its output slot)
__global__ __launch_bounds__(FQ_THREADS) void fa_copy(FaLines A, const uint32_t *tile_off, const int64_t *dst, uint8_t *bases)
{
    __shared__ uint32_t wt[FQ_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t pos = ((int64_t)blockIdx.x * FQ_THREADS + t) * 16;
    uint32_t m = pos < A.n ? nl_mask16(A.text, pos, A.n) : 0u;
    const uint32_t c = (uint32_t)__popc(m);
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wt[wv] = inc;
    __syncthreads();
    int64_t k = (int64_t)tile_off[blockIdx.x] + inc - c;
    for (int i = 0; i < wv; ++i) k += wt[i];
    if (pos >= A.n) return;
    int64_t s = fa_line_start(A.nl, k);
    int64_t d = dst[k];
    uint32_t pay = A.pay[k];
    const int nb = A.n - pos < 16 ? (int)(A.n - pos) : 16;
    if (m == 0 && nb == 16) {
        if (d < 0) return;
        const int64_t o = pos - s;                       // offset of my first byte inside the line
        if (o + 16 <= (int64_t)pay) {
            const uint4 v = *reinterpret_cast<const uint4 *>(A.text + pos);
            fq_u32u *q = reinterpret_cast<fq_u32u *>(bases + d + o);      // the destination has no alignment
            q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
            return;
        }
    }
    for (int b = 0; b < nb; ++b) {
        const int64_t p = pos + b;
        if ((m >> b) & 1u) {                             // a newline: the next byte starts line k + 1
            ++k;
            s = p + 1;
            if (k < A.n_lines) {
                d = dst[k];
                pay = A.pay[k];
            } else {
                d = -1;
            }
            continue;
        }
        if (d >= 0 && p - s < (int64_t)pay) bases[d + (p - s)] = A.text[p];
    }
}

}  // namespace

extern "C" int cornetto_fastq_split(cornetto_accel_t *h, const char *text, int64_t n, int final, int32_t min_len, cornetto_fqrec_t **recs,
                                    int64_t *n_recs, int64_t *consumed, int32_t *plain, cornetto_asm_t **reads)
{
    if (!h || n < 0 || (n > 0 && !text) || !recs || !n_recs || !consumed || !plain)
        return cn_fail(h, CORNETTO_E_ARG, "fastq_split: bad argument");
    if (n > 0xFFFFFF00LL) return cn_fail(h, CORNETTO_E_ARG, "fastq_split: pieces are limited to 2^32-256 bytes (got %lld)", (long long)n);
    *recs = nullptr;
    *n_recs = 0;
    *consumed = 0;
    *plain = 1;
    if (reads) *reads = nullptr;
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    int64_t n_use = 0, used_bytes = 0;
    uint8_t *d_text = nullptr;
    cornetto_fqrec_t *d_recs = nullptr;
    if (n > 0) {
        const int64_t nt = (n + FQ_TILE - 1) / FQ_TILE;
        d_text = (uint8_t *)cn_ws(h, WS_FQ_TEXT, (size_t)n + 64);
        uint32_t *d_cnt = (uint32_t *)cn_ws(h, WS_FQ_CNT, ((size_t)2 * nt + (nt + 4095) / 4096 + 16) * 4 + 32);
        unsigned long long *p_small = (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
        if (!d_text || !d_cnt || !p_small) return cn_fail(h, CORNETTO_E_NOMEM, "fastq_split: workspace allocation failed");
        uint32_t *d_off = d_cnt + nt, *d_part = d_off + nt;
        unsigned long long *d_tot = reinterpret_cast<unsigned long long *>(((uintptr_t)(d_part + (nt + 4095) / 4096 + 1) + 7) & ~(uintptr_t)7);
        uint32_t *d_bad = reinterpret_cast<uint32_t *>(d_tot + 1);
        CN_HIP(h, hipMemcpyAsync(d_text, text, (size_t)n, hipMemcpyHostToDevice, h->stream));
        CN_LAUNCH(h, "fq_nl_count", fq_nl_count<<<dim3((unsigned)nt), dim3(FQ_THREADS), 0, h->stream>>>(d_text, n, d_cnt));
        CN_TRY(cnscan::exclusive_u32(h, "fq_scan", d_cnt, nt, 1, d_off, d_part, d_tot));
        CN_HIP(h, hipMemcpyAsync(p_small, d_tot, 8, hipMemcpyDeviceToHost, h->stream));
        CN_HIP(h, hipStreamSynchronize(h->stream));
        const int64_t n_nl = (int64_t)p_small[0];
        const bool virt = final && text[n - 1] != '\n';
        const int64_t n_lines = n_nl + (virt ? 1 : 0);
        const int64_t n_rec = n_lines / 4;
        if (n_rec > 0x7fffffffLL) return cn_fail(h, CORNETTO_E_ARG, "fastq_split: more than 2^31-1 records in one piece");
        if (n_rec > 0) {
            uint32_t *d_nl = (uint32_t *)cn_ws(h, WS_FQ_NL, ((size_t)n_nl + 8) * 4);
            d_recs = (cornetto_fqrec_t *)cn_ws(h, WS_FQ_RECS, (size_t)n_rec * sizeof(cornetto_fqrec_t));
            uint32_t *d_ends = (uint32_t *)cn_ws(h, WS_FQ_ENDS, (size_t)n_rec * 4);
            if (!d_nl || !d_recs || !d_ends) return cn_fail(h, CORNETTO_E_NOMEM, "fastq_split: workspace allocation failed");
            CN_HIP(h, hipMemsetAsync(d_bad, 0xFF, 4, h->stream));
            if (n_nl) CN_LAUNCH(h, "fq_nl_scatter", fq_nl_scatter<<<dim3((unsigned)nt), dim3(FQ_THREADS), 0, h->stream>>>(d_text, n, d_off, d_nl));
            FqArgs A{d_text, n, d_nl, n_nl, n_rec, min_len, d_recs, d_bad, d_ends};
            CN_LAUNCH(h, "fq_records", fq_records<<<dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, h->stream>>>(A));
            uint32_t *p_bad = reinterpret_cast<uint32_t *>(p_small + 1);
            CN_HIP(h, hipMemcpyAsync(p_bad, d_bad, 4, hipMemcpyDeviceToHost, h->stream));
            CN_HIP(h, hipStreamSynchronize(h->stream));
            n_use = (int64_t)p_bad[0] < n_rec ? (int64_t)p_bad[0] : n_rec;
            if (n_use < n_rec) *plain = 0;
            if (n_use > 0) {
                uint32_t *p_end = p_bad + 1;
                CN_HIP(h, hipMemcpyAsync(p_end, d_ends + (n_use - 1), 4, hipMemcpyDeviceToHost, h->stream));
                cornetto_fqrec_t *out = (cornetto_fqrec_t *)cn_result_alloc((size_t)n_use * sizeof(cornetto_fqrec_t));
                if (!out) return cn_fail(h, CORNETTO_E_NOMEM, "fastq_split: result allocation failed");
                if (hipMemcpyAsync(out, d_recs, (size_t)n_use * sizeof(cornetto_fqrec_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                    hipStreamSynchronize(h->stream) != hipSuccess) {
                    cornetto_free(out);
                    return cn_fail(h, CORNETTO_E_HIP, "fastq_split: copying the record table failed");
                }
                used_bytes = (int64_t)p_end[0];
                *recs = out;
            }
        }
        // at the end of the input whatever follows the last plain record (blank lines, a cut-off record) is the reader's
        if (final && used_bytes < n && *plain) *plain = 0;
    }
    *n_recs = n_use;
    *consumed = used_bytes;
    if (reads) {
        const cornetto_fqrec_t *R = *recs;
        std::vector<int32_t> lens;
        std::vector<int64_t> src;
        for (int64_t i = 0; i < n_use; ++i)
            if (R[i].keep) {
                lens.push_back(R[i].len);
                src.push_back(R[i].seq);
            }
        cornetto_asm_t *a = nullptr;
        int rc = cn_asm_alloc(h, lens.data(), (int32_t)lens.size(), &a);
        if (rc == CORNETTO_OK && !lens.empty()) {
            int64_t *d_src = (int64_t *)cn_ws(h, WS_FQ_SRC, lens.size() * 8);
            if (!d_src) rc = cn_fail(h, CORNETTO_E_NOMEM, "fastq_split: workspace allocation failed");
            else if (hipMemcpyAsync(d_src, src.data(), src.size() * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = CORNETTO_E_HIP;
            if (rc == CORNETTO_OK) {
                const unsigned blocks = (unsigned)(lens.size() < 16384 ? lens.size() : 16384);
                cornetto_accel::Rec r_{"fq_pack", cn_event(h), cn_event(h)};
                (void)hipEventRecord(r_.a, h->stream);
                fq_pack<<<dim3(blocks), dim3(256), 0, h->stream>>>(d_text, d_src, a->d_off, a->d_len, (int32_t)lens.size(), (uint8_t *)a->owned);
                (void)hipEventRecord(r_.b, h->stream);
                h->recs.push_back(r_);
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) rc = cn_fail(h, CORNETTO_E_HIP, "fastq_split: fq_pack failed");
            }
        }
        if (rc != CORNETTO_OK) {
            if (a) cornetto_asm_free(h, a);
            cornetto_free(*recs);
            *recs = nullptr;
            *n_recs = 0;
            *consumed = 0;
            return rc;
        }
        *reads = a;
    }
    cn_timing_end(h);
    return CORNETTO_OK;
}


// the framing of FASTA text that lies on the device (d_text: n bytes, readable up to n + 64; text_in != NULL: uploaded here first)
static int fasta_split_core(cornetto_accel_t *h, const char *text_in, uint8_t *d_text_in, char first_ch, char last_ch, int64_t n, int final, cornetto_farec_t **recs,
                            int64_t *n_recs, int64_t *consumed, int32_t *plain, cornetto_asm_t **seqs)
{
    if (n > 0xFFFFFF00LL) return cn_fail(h, CORNETTO_E_ARG, "fasta_split: pieces are limited to 2^32-256 bytes (got %lld)", (long long)n);
    *recs = nullptr;
    *n_recs = 0;
    *consumed = 0;
    *plain = 1;
    if (seqs) *seqs = nullptr;
    if (n == 0) return CORNETTO_OK;
    if (first_ch != '>') {      // kseq_read looks for the first '>' or '@' anywhere (:189-193): the caller's reader does that
        *plain = 0;
        return CORNETTO_OK;
    }
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    CN_TRACE("fasta_split: enter");
    const int64_t nt = (n + FQ_TILE - 1) / FQ_TILE;
    uint8_t *d_text = d_text_in ? d_text_in : (uint8_t *)cn_ws(h, WS_FQ_TEXT, (size_t)n + 64);
    uint32_t *d_cnt = (uint32_t *)cn_ws(h, WS_FQ_CNT, ((size_t)2 * nt + (nt + 4095) / 4096 + 16) * 4 + 64);
    unsigned long long *p_small = (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
    if (!d_text || !d_cnt || !p_small) return cn_fail(h, CORNETTO_E_NOMEM, "fasta_split: workspace allocation failed");
    CN_TRACE("fasta_split: text workspace");
    uint32_t *d_off = d_cnt + nt, *d_part = d_off + nt;
    unsigned long long *d_tot = reinterpret_cast<unsigned long long *>(((uintptr_t)(d_part + (nt + 4095) / 4096 + 1) + 7) & ~(uintptr_t)7);
    uint32_t *d_bad = reinterpret_cast<uint32_t *>(d_tot + 3);
    if (text_in) CN_HIP(h, hipMemcpyAsync(d_text, text_in, (size_t)n, hipMemcpyHostToDevice, h->stream));
    CN_LAUNCH(h, "fq_nl_count", fq_nl_count<<<dim3((unsigned)nt), dim3(FQ_THREADS), 0, h->stream>>>(d_text, n, d_cnt));
    CN_TRY(cnscan::exclusive_u32(h, "fq_scan", d_cnt, nt, 1, d_off, d_part, d_tot));
    CN_HIP(h, hipMemcpyAsync(p_small, d_tot, 8, hipMemcpyDeviceToHost, h->stream));
    CN_HIP(h, hipStreamSynchronize(h->stream));
    CN_TRACE("fasta_split: text up, newlines counted");
    const int64_t n_nl = (int64_t)p_small[0];
    const bool virt = last_ch != '\n';     // the bytes behind the last newline are a line too (complete only if `final`)
    const int64_t n_lines = n_nl + (virt ? 1 : 0);
    // per line: newline offset, header flag, payload, their exclusive scans (n_lines + 1 entries: the last holds the totals),
    // destination; per record: header line
    const size_t nl1 = (size_t)n_lines + 1;
    uint32_t *d_nl = (uint32_t *)cn_ws(h, WS_FQ_NL, (nl1 + 8) * 4);
    uint32_t *d_lw = (uint32_t *)cn_ws(h, WS_FQ_ENDS, (5 * nl1 + 2 * ((nl1 + 4095) / 4096 + 2) + 16) * 4);
    int64_t *d_dst = (int64_t *)cn_ws(h, WS_FQ_SRC, nl1 * 8);
    if (!d_nl || !d_lw || !d_dst) return cn_fail(h, CORNETTO_E_NOMEM, "fasta_split: workspace allocation failed");
    uint32_t *d_hdr = d_lw, *d_pay = d_hdr + nl1, *d_H = d_pay + nl1, *d_P = d_H + nl1, *d_hl = d_P + nl1, *d_pp = d_hl + nl1;
    CN_HIP(h, hipMemsetAsync(d_bad, 0xFF, 4, h->stream));
    if (n_nl) CN_LAUNCH(h, "fq_nl_scatter", fq_nl_scatter<<<dim3((unsigned)nt), dim3(FQ_THREADS), 0, h->stream>>>(d_text, n, d_off, d_nl));
    FaLines A{d_text, n, d_nl, n_nl, n_lines, d_hdr, d_pay, d_bad};
    const unsigned nbl = (unsigned)((nl1 + 255) / 256);
    CN_LAUNCH(h, "fa_lines", fa_lines<<<dim3(nbl), dim3(256), 0, h->stream>>>(A));
    CN_TRY(cnscan::exclusive_u32(h, "fa_scan", d_hdr, (int64_t)nl1, 1, d_H, d_pp, d_tot + 1));
    CN_TRY(cnscan::exclusive_u32(h, "fa_scan", d_pay, (int64_t)nl1, 1, d_P, d_pp, d_tot + 2));
    CN_LAUNCH(h, "fa_heads", fa_heads<<<dim3(nbl), dim3(256), 0, h->stream>>>(A, d_H, d_P, d_hl, 0));
    CN_LAUNCH(h, "fa_heads", fa_heads<<<dim3(nbl), dim3(256), 0, h->stream>>>(A, d_H, d_P, d_hl, 1));
    uint32_t *p_u = reinterpret_cast<uint32_t *>(p_small + 4);
    CN_HIP(h, hipMemcpyAsync(p_small, d_tot + 1, 16, hipMemcpyDeviceToHost, h->stream));
    CN_HIP(h, hipMemcpyAsync(p_u, d_bad, 4, hipMemcpyDeviceToHost, h->stream));
    CN_HIP(h, hipStreamSynchronize(h->stream));
    CN_TRACE("fasta_split: lines, heads");
    const int64_t n_rec = (int64_t)p_small[0];
    if (p_small[1] > 0xFFFFFFFFull) return cn_fail(h, CORNETTO_E_ARG, "fasta_split: sequence bytes of one piece exceed 2^32");
    int64_t n_use = final ? n_rec : n_rec - 1;            // the last record of a piece that is not the last one may go on
    if (p_u[0] != 0xFFFFFFFFu) {                          // not plain from the record of that line on
        *plain = 0;
        uint32_t *p_r = p_u + 1;
        CN_HIP(h, hipMemcpyAsync(p_r, d_H + p_u[0], 4, hipMemcpyDeviceToHost, h->stream));
        CN_HIP(h, hipStreamSynchronize(h->stream));
        const int64_t r_bad = (int64_t)p_r[0] - 1;        // the bad line is never a header line: H = records begun before it
        if (r_bad < n_use) n_use = r_bad;
    }
    if (n_use < 0) n_use = 0;
    if (n_rec > 0x7fffffffLL) return cn_fail(h, CORNETTO_E_ARG, "fasta_split: more than 2^31-1 records in one piece");
    cornetto_farec_t *out = nullptr;
    int64_t used = 0;
    if (n_rec > 0) {
        cornetto_farec_t *d_recs = (cornetto_farec_t *)cn_ws(h, WS_FQ_RECS, (size_t)n_rec * sizeof(cornetto_farec_t));
        if (!d_recs) return cn_fail(h, CORNETTO_E_NOMEM, "fasta_split: workspace allocation failed");
        CN_LAUNCH(h, "fa_records", fa_records<<<dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, h->stream>>>(A, d_P, d_hl, n_rec, d_recs));
        out = (cornetto_farec_t *)cn_result_alloc((size_t)n_rec * sizeof(cornetto_farec_t));
        if (!out) return cn_fail(h, CORNETTO_E_NOMEM, "fasta_split: result allocation failed");
        if (hipMemcpyAsync(out, d_recs, (size_t)n_rec * sizeof(cornetto_farec_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) {
            cornetto_free(out);
            return cn_fail(h, CORNETTO_E_HIP, "fasta_split: copying the record table failed");
        }
        for (int64_t i = 0; i < n_use; ++i)
            if (out[i].len > 0x7fffffffLL) {              // kseq's int length (src/kseq.h:185): the caller's reader reports it
                n_use = i;
                *plain = 0;
                break;
            }
        used = n_use < n_rec ? out[n_use].head : n;
    }
    if (final && used < n) *plain = 0;
    if (n_use == 0) {
        cornetto_free(out);
        out = nullptr;
    }
    *recs = out;
    *n_recs = n_use;
    *consumed = used;
    CN_TRACE("fasta_split: record table");
    if (seqs) {
        std::vector<int32_t> lens((size_t)n_use);
        for (int64_t i = 0; i < n_use; ++i) lens[(size_t)i] = (int32_t)out[i].len;
        cornetto_asm_t *a = nullptr;
        int rc = cn_asm_alloc(h, lens.data(), (int32_t)n_use, &a);
        CN_TRACE("fasta_split: assembly allocated");
        if (rc == CORNETTO_OK && n_use > 0) {
            cornetto_accel::Rec r1{"fa_linedst", cn_event(h), cn_event(h)}, r2{"fa_copy", cn_event(h), cn_event(h)};
            (void)hipEventRecord(r1.a, h->stream);
            fa_linedst<<<dim3(nbl), dim3(256), 0, h->stream>>>(A, d_H, d_P, d_hl, a->d_off, n_use, d_dst);
            (void)hipEventRecord(r1.b, h->stream);
            (void)hipEventRecord(r2.a, h->stream);
            fa_copy<<<dim3((unsigned)nt), dim3(FQ_THREADS), 0, h->stream>>>(A, d_off, d_dst, (uint8_t *)a->owned);
            (void)hipEventRecord(r2.b, h->stream);
            h->recs.push_back(r1);
            h->recs.push_back(r2);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) rc = cn_fail(h, CORNETTO_E_HIP, "fasta_split: fa_copy failed");
        }
        if (rc != CORNETTO_OK) {
            if (a) cornetto_asm_free(h, a);
            cornetto_free(out);
            *recs = nullptr;
            *n_recs = 0;
            *consumed = 0;
            return rc;
        }
        *seqs = a;
        CN_TRACE("fasta_split: sequences copied");
    }
    cn_timing_end(h);
    return CORNETTO_OK;
}

extern "C" int cornetto_fasta_split(cornetto_accel_t *h, const char *text, int64_t n, int final, cornetto_farec_t **recs, int64_t *n_recs,
                                    int64_t *consumed, int32_t *plain, cornetto_asm_t **seqs)
{
    if (!h || n < 0 || (n > 0 && !text) || !recs || !n_recs || !consumed || !plain)
        return cn_fail(h, CORNETTO_E_ARG, "fasta_split: bad argument");
    return fasta_split_core(h, text, nullptr, n > 0 ? text[0] : 0, n > 0 ? text[n - 1] : 0, n, final, recs, n_recs, consumed, plain, seqs);
}

// ---- a text put on the device slab by slab (include/cornetto_accel.h: cornetto_text_*) ------------------------------------------------
struct cornetto_text {
    uint8_t *d = nullptr;
    int64_t cap = 0;
    hipStream_t q[4] = {nullptr, nullptr, nullptr, nullptr};     // copy queues: one copy in flight moves ~28 GB/s over PCIe, several ~45
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // the last copy of slot s (the caller's slab ring has up to four slabs)
};

extern "C" int cornetto_text_open(cornetto_accel_t *h, int64_t capacity, cornetto_text_t **out)
{
    if (!h || !out || capacity < 1 || capacity > 0xFFFFFF00LL) return cn_fail(h, CORNETTO_E_ARG, "text_open: bad argument");
    *out = nullptr;
    CN_HIP(h, hipSetDevice(h->device));
    cornetto_text_t *t = new (std::nothrow) cornetto_text;
    if (!t) return cn_fail(h, CORNETTO_E_NOMEM, "text_open: host allocation failed");
    bool ok = hipMalloc((void **)&t->d, (size_t)capacity + 256) == hipSuccess;
    for (int i = 0; ok && i < 4; ++i) ok = hipStreamCreateWithFlags(&t->q[i], hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 4; ++i) ok = hipEventCreateWithFlags(&t->ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        cornetto_text_free(h, t);
        return cn_fail(h, CORNETTO_E_NOMEM, "text_open: device allocation of %lld bytes failed", (long long)capacity);
    }
    t->cap = capacity;
    *out = t;
    return CORNETTO_OK;
}

extern "C" void cornetto_text_free(cornetto_accel_t *h, cornetto_text_t *t)
{
    if (!t) return;
    if (h) (void)hipSetDevice(h->device);
    for (int i = 0; i < 4; ++i)
        if (t->q[i]) { (void)hipStreamSynchronize(t->q[i]); (void)hipStreamDestroy(t->q[i]); }
    for (int i = 0; i < 4; ++i)
        if (t->ev[i]) (void)hipEventDestroy(t->ev[i]);
    if (t->d) (void)hipFree(t->d);
    delete t;
}

extern "C" int cornetto_text_put(cornetto_accel_t *h, cornetto_text_t *t, const char *slab, int64_t n, int64_t at, int slot)
{
    if (!h || !t || !slab || n < 0 || at < 0 || at + n > t->cap || slot < 0 || slot > 3) return cn_fail(h, CORNETTO_E_ARG, "text_put: bad argument");
    if (n == 0) return CORNETTO_OK;
    CN_HIP(h, hipSetDevice(h->device));
    hipStream_t q = t->q[slot];
    CN_HIP(h, hipMemcpyAsync(t->d + at, slab, (size_t)n, hipMemcpyHostToDevice, q));
    CN_HIP(h, hipEventRecord(t->ev[slot], q));
    return CORNETTO_OK;
}

extern "C" int cornetto_text_wait(cornetto_accel_t *h, cornetto_text_t *t, int slot)
{
    if (!h || !t || slot < 0 || slot > 3) return cn_fail(h, CORNETTO_E_ARG, "text_wait: bad argument");
    CN_HIP(h, hipSetDevice(h->device));
    CN_HIP(h, hipEventSynchronize(t->ev[slot]));
    return CORNETTO_OK;
}

extern "C" int cornetto_fasta_split_text(cornetto_accel_t *h, cornetto_text_t *t, int64_t n, int final, cornetto_farec_t **recs, int64_t *n_recs, int64_t *consumed,
                                         int32_t *plain, cornetto_asm_t **seqs)
{
    if (!h || !t || n < 0 || n > t->cap || !recs || !n_recs || !consumed || !plain) return cn_fail(h, CORNETTO_E_ARG, "fasta_split_text: bad argument");
    CN_HIP(h, hipSetDevice(h->device));
    for (int i = 0; i < 4; ++i) CN_HIP(h, hipStreamSynchronize(t->q[i]));      // every slab is on the device
    char ends[2] = {0, 0};
    if (n > 0) {
        CN_HIP(h, hipMemcpy(&ends[0], t->d, 1, hipMemcpyDeviceToHost));
        CN_HIP(h, hipMemcpy(&ends[1], t->d + n - 1, 1, hipMemcpyDeviceToHost));
    }
    return fasta_split_core(h, nullptr, t->d, ends[0], ends[1], n, final, recs, n_recs, consumed, plain, seqs);
}

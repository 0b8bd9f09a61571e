/* host_backend.c — the product's own HOST path of the panel scans: plain sequential C99, no device, no library call.
 *
 * Selected explicitly — `--accel=no` (the switch the reference left for this: src/boringbits_main.c:627-632,
 * src/cornetto.c:323-325) or CORNETTO_ACCEL=no for the sub-commands that have no such option — never silently: without
 * that choice a box without a usable GPU still gets the "cannot open HIP device" exit.  It exists so that the binary runs
 * BASELINE's configuration 1 ("CPU only ... plumbing, no GPU") as written and so that `--accel=no` means what it says.
 * It is not a tuned CPU implementation and not the timed baseline of bench.py (that is the unmodified reference).
 *
 * Every unit takes and returns the C ABI's record types (include/cornetto_accel.h), so the sub-command mains print the
 * host results with the code that prints the device results.  Written for this file; nothing under oracle/ is used. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cli.h"

/* ------------------------------------------------------------------------------------------------ the switch */
static int g_host = -1;

int cli_host_mode(void)
{
    if (g_host < 0) {
        const char *e = getenv("CORNETTO_ACCEL");
        g_host = e && (!strcmp(e, "no") || !strcmp(e, "n") || !strcmp(e, "0") || !strcmp(e, "cpu") || !strcmp(e, "host"));
    }
    return g_host;
}

void cli_host_set(int on) { g_host = on ? 1 : 0; }

/* ------------------------------------------------------------------------------------------------ telofind
 * src/find_telomere.c:44-74 on one record: the contig is compared upper-cased (:76-81; the motif is taken as given),
 * runs are greedy and never overlap (:52-57), strand 0 = the motif, then strand 1 = its reverse complement (:61-72). */
typedef struct {
    cornetto_hit_t *a;
    int64_t n, cap;
} hitbuf_t;

static inline uint8_t fold(uint8_t c) { return (uint8_t)(c >= 'a' && c <= 'z' ? c - 32 : c); }

static inline int motif_at(const uint8_t *s, const uint8_t *pat, int64_t k)
{
    for (int64_t j = 0; j < k; ++j)
        if (fold(s[j]) != pat[j]) return 0;
    return 1;
}

static void strand_runs(const uint8_t *s, int64_t n, const uint8_t *pat, int64_t k, int32_t ctg, int32_t strand, hitbuf_t *out)
{
    int64_t p = 0;
    while (p + k <= n) {
        while (p + k <= n && !motif_at(s + p, pat, k)) ++p;      /* the next occurrence at or behind p (:49) */
        if (p + k > n) break;
        int64_t e = p + k;
        while (e + k <= n && motif_at(s + e, pat, k)) e += k;     /* :52-55 */
        if (out->n == out->cap) {
            out->cap = out->cap ? out->cap * 2 : 256;
            out->a = (cornetto_hit_t *)cli_xrealloc(out->a, (size_t)out->cap * sizeof(*out->a));
        }
        out->a[out->n].ctg = ctg;
        out->a[out->n].strand = strand;
        out->a[out->n].start = (int32_t)p;
        out->a[out->n].end = (int32_t)e;
        out->n++;
        p = e + 1;                                                /* :57 */
    }
}

void cli_host_telofind(const uint8_t *seq, int64_t len, const char *motif, int32_t ctg, cornetto_hit_t **hits, int64_t *n_hits, int64_t *cap_hits)
{
    hitbuf_t b = {*hits, *n_hits, *cap_hits};
    const int64_t k = (int64_t)strlen(motif);
    const uint8_t *nul = (const uint8_t *)memchr(seq, 0, (size_t)len);   /* strstr() of the reference ends at a NUL byte */
    const int64_t n = nul ? (int64_t)(nul - seq) : len;
    if (k > 0 && k <= n) {
        uint8_t *rc = (uint8_t *)cli_xmalloc((size_t)k + 1);
        for (int64_t i = 0; i < k; ++i) {                         /* :24-42: A<->T, C<->G, anything else as it is */
            const char c = motif[k - 1 - i];
            rc[i] = (uint8_t)(c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c);
        }
        strand_runs(seq, n, (const uint8_t *)motif, k, ctg, 0, &b);
        strand_runs(seq, n, rc, k, ctg, 1, &b);
        free(rc);
    }
    *hits = b.a;
    *n_hits = b.n;
    *cap_hits = b.cap;
}

/* ------------------------------------------------------------------------------------------------ telowin
 * src/telomere_windows.c:69-79 (marks) and :28-43 (windows): one bit per base, windows of 1000 every 200. */
static inline int64_t bits_in(const uint64_t *w, int64_t a, int64_t b) /* set bits in [a, b) */
{
    if (a >= b) return 0;
    const int64_t wa = a >> 6, wb = (b - 1) >> 6;
    const uint64_t ma = ~0ull << (a & 63), mb = ~0ull >> (63 - ((b - 1) & 63));
    if (wa == wb) return __builtin_popcountll(w[wa] & ma & mb);
    int64_t c = __builtin_popcountll(w[wa] & ma) + __builtin_popcountll(w[wb] & mb);
    for (int64_t i = wa + 1; i < wb; ++i) c += __builtin_popcountll(w[i]);
    return c;
}

static inline void bits_set(uint64_t *w, int64_t a, int64_t b) /* [a, b) */
{
    if (a >= b) return;
    const int64_t wa = a >> 6, wb = (b - 1) >> 6;
    const uint64_t ma = ~0ull << (a & 63), mb = ~0ull >> (63 - ((b - 1) & 63));
    if (wa == wb) {
        w[wa] |= ma & mb;
        return;
    }
    w[wa] |= ma;
    for (int64_t i = wa + 1; i < wb; ++i) w[i] = ~0ull;
    w[wb] |= mb;
}

void cli_host_telowin(const cornetto_hit_t *hits, int64_t n_hits, const int32_t *lens, int32_t n_ctg, double thr_adj, cornetto_win_t **wins, int64_t *n_wins)
{
    cornetto_win_t *o = NULL;
    int64_t n = 0, cap = 0, h = 0;
    for (int32_t c = 0; c < n_ctg; ++c) {
        const int64_t len = lens[c];
        uint64_t *mark = (uint64_t *)calloc((size_t)(len / 64 + 2), sizeof(uint64_t));
        if (!mark) {
            CLI_ERROR("%s", "out of memory");
            exit(EXIT_FAILURE);
        }
        for (; h < n_hits && hits[h].ctg == c; ++h) bits_set(mark, hits[h].start, hits[h].end);
        for (int64_t i = 0; i <= len; i += 200) {                                  /* :31 */
            const int64_t hi = i + 1000 < len ? i + 1000 : len;
            const int64_t car = bits_in(mark, i, hi);                              /* :33-35 */
            const int64_t den = i + 1000 < len ? 1000 : len - i;                   /* :36 */
            if ((double)car / (double)den >= thr_adj) {                            /* :37 (0 / 0 = NaN: no window) */
                if (n == cap) {
                    cap = cap ? cap * 2 : 256;
                    o = (cornetto_win_t *)cli_xrealloc(o, (size_t)cap * sizeof(*o));
                }
                o[n].ctg = c;
                o[n].start = (int32_t)i;
                o[n].end = (int32_t)(i + den);
                o[n].car = (int32_t)car;
                ++n;
            }
            if (i + 1000 >= len) break;                                            /* :40 */
        }
        free(mark);
    }
    *wins = o;
    *n_wins = n;
}

/* ------------------------------------------------------------------------------------------------ sdust
 * src/sdust/sdust.c:130-160 on one record, base by base.  State: the FIFO of the last <= W - 2 three-letter words with
 * its pair score (rw), the suffix of it in which no word occurs more than 2T/10 times (length L, pair score rv), the
 * perfect intervals found so far that may still grow (P, by descending start), the result list. */
typedef struct {
    int32_t start, finish, r, l;
} perfect_t;

typedef struct {
    uint8_t *fifo;
    int cap, head, n;
    int cw[64], cv[64];
    int rw, rv, L;
    perfect_t *P;
    int nP, capP;
    cornetto_ivl_t *out;
    int64_t n_out, cap_out, first_out; /* first_out: where this record's intervals begin (merging never reaches further back) */
    int32_t ctg;
} dust_t;

static inline int dust_word(const dust_t *d, int i) { return d->fifo[(d->head + i) % d->cap]; }

/* :88-102: the interval with the smallest start has left the window for good -> into the result, merged with the last
 * one when they overlap or touch; then every perfect interval that starts in front of `start` is dropped */
static void dust_retire(dust_t *d, int start)
{
    if (d->nP == 0 || d->P[d->nP - 1].start >= start) return;
    const perfect_t *p = &d->P[d->nP - 1];
    if (d->n_out > d->first_out && p->start <= d->out[d->n_out - 1].finish) {
        if (p->finish > d->out[d->n_out - 1].finish) d->out[d->n_out - 1].finish = p->finish;
    } else {
        if (d->n_out == d->cap_out) {
            d->cap_out = d->cap_out ? d->cap_out * 2 : 256;
            d->out = (cornetto_ivl_t *)cli_xrealloc(d->out, (size_t)d->cap_out * sizeof(*d->out));
        }
        d->out[d->n_out].ctg = d->ctg;
        d->out[d->n_out].start = p->start;
        d->out[d->n_out].finish = p->finish;
        d->n_out++;
    }
    while (d->nP > 0 && d->P[d->nP - 1].start < start) d->nP--;
}

/* :66-86 */
static void dust_push(dust_t *d, int t, int T, int W)
{
    if (d->n >= W - 2) {
        const int x = d->fifo[d->head];
        d->head = (d->head + 1) % d->cap;
        d->n--;
        d->rw -= --d->cw[x];
        if (d->L > d->n) {
            d->L--;
            d->rv -= --d->cv[x];
        }
    }
    d->fifo[(d->head + d->n) % d->cap] = (uint8_t)t;
    d->n++;
    d->L++;
    d->rw += d->cw[t]++;
    d->rv += d->cv[t]++;
    if (d->cv[t] * 10 > T << 1) {
        int x;
        do {
            x = dust_word(d, d->n - d->L);
            d->rv -= --d->cv[x];
            d->L--;
        } while (x != t);
    }
}

/* :104-128: every suffix of the window that is longer than the L-suffix, from short to long */
static void dust_perfect(dust_t *d, int T, int start)
{
    int c[64];
    memcpy(c, d->cv, sizeof(c));
    int r = d->rv, best_r = 0, best_l = 0;
    for (int i = d->n - d->L - 1; i >= 0; --i) {
        const int t = dust_word(d, i);
        r += c[t]++;
        const int l = d->n - i - 1;
        if (r * 10 <= T * l) continue;
        int j = 0;
        for (; j < d->nP && d->P[j].start >= i + start; ++j)
            if (best_r == 0 || d->P[j].r * best_l > best_r * d->P[j].l) {
                best_r = d->P[j].r;
                best_l = d->P[j].l;
            }
        if (best_r == 0 || r * best_l >= best_r * l) {
            best_r = r;
            best_l = l;
            if (d->nP == d->capP) {
                d->capP = d->capP ? d->capP * 2 : 64;
                d->P = (perfect_t *)cli_xrealloc(d->P, (size_t)d->capP * sizeof(*d->P));
            }
            memmove(d->P + j + 1, d->P + j, (size_t)(d->nP - j) * sizeof(*d->P));
            d->nP++;
            d->P[j].start = i + start;
            d->P[j].finish = d->n + 2 + start;
            d->P[j].r = r;
            d->P[j].l = l;
        }
    }
}

static inline int base_code(uint8_t c) /* seq_nt4_table, :23-40: ACGT in either case, the bytes 0..3 themselves, 4 otherwise */
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return c < 4 ? c : 4;
    }
}

int cli_host_sdust(const uint8_t *seq, int64_t len, int T, int W, int32_t ctg, cornetto_ivl_t **ivls, int64_t *n_ivls, int64_t *cap_ivls)
{
    if (W < 3 || W > 1026 || T < 0 || T > (1 << 20) || len > 0x7fffffffLL) return -1;   /* (the limits of the device path) */
    dust_t d;
    memset(&d, 0, sizeof(d));
    d.cap = W + 2;
    d.fifo = (uint8_t *)cli_xmalloc((size_t)d.cap);
    d.out = *ivls;
    d.n_out = d.first_out = *n_ivls;
    d.cap_out = *cap_ivls;
    d.ctg = ctg;
    int l = 0;
    unsigned t = 0;
    for (int64_t i = 0; i <= len; ++i) {                                  /* :141: one step past the end flushes P */
        const int b = i < len ? base_code(seq[i]) : 4;
        if (b < 4) {
            ++l;
            t = (t << 2 | (unsigned)b) & 63u;
            if (l >= 3) {
                const int start = (l - W > 0 ? l - W : 0) + (int)(i + 1 - l);          /* :146 */
                dust_retire(&d, start);
                dust_push(&d, (int)t, T, W);
                if (d.rw * 10 > d.L * T) dust_perfect(&d, T, start);                  /* :149 */
            }
        } else {
            int start = (l - W + 1 > 0 ? l - W + 1 : 0) + (int)(i + 1 - l);            /* :152 */
            while (d.nP) dust_retire(&d, start++);                                     /* :153 */
            l = 0;                                                                     /* :154: the words and their counts stay */
            t = 0;
        }
    }
    free(d.fifo);
    free(d.P);
    *ivls = d.out;
    *n_ivls = d.n_out;
    *cap_ivls = d.cap_out;
    return 0;
}

/* ------------------------------------------------------------------------------------------------ get_depths
 * src/boringbits_main.c:180-301: the two per-base bedgraphs read in lock-step the way fscanf("%s\t%d\t%d\t%d\n") reads
 * them — a record is four white-space separated tokens whatever the lines look like — with the reference's checks in
 * its order.  A small buffered byte stream with the two conversions, not fscanf: no 10 000-byte name buffer to overrun. */
typedef struct {
    FILE *f;
    unsigned char *buf;
    size_t n, at;
    int eof;
} bgs_t;

static inline int bgs_peek(bgs_t *s)
{
    if (s->at == s->n) {
        if (s->eof) return -1;
        s->n = fread(s->buf, 1, 1 << 20, s->f);
        s->at = 0;
        if (s->n == 0) {
            s->eof = 1;
            return -1;
        }
    }
    return s->buf[s->at];
}

static inline int is_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

static void bgs_skip_space(bgs_t *s)
{
    int c;
    while ((c = bgs_peek(s)) >= 0 && is_space(c)) s->at++;
}

/* %s: 1 converted, 0 end of input */
static int bgs_string(bgs_t *s, cli_str_t *name)
{
    bgs_skip_space(s);
    int c = bgs_peek(s);
    if (c < 0) return 0;
    name->l = 0;
    while ((c = bgs_peek(s)) >= 0 && !is_space(c)) {
        if (name->l + 2 > name->m) {
            name->m = name->m ? name->m * 2 : 64;
            name->s = (char *)cli_xrealloc(name->s, name->m);
        }
        name->s[name->l++] = (char)c;
        s->at++;
    }
    name->s[name->l] = 0;
    return 1;
}

/* %d: 1 converted, 0 no number here (or end of input) */
static int bgs_int(bgs_t *s, int *v)
{
    bgs_skip_space(s);
    int c = bgs_peek(s), neg = 0;
    if (c == '-' || c == '+') {
        neg = c == '-';
        s->at++;
        c = bgs_peek(s);
    }
    if (c < '0' || c > '9') return 0;
    long long x = 0;
    while ((c = bgs_peek(s)) >= '0' && c <= '9') {
        if (x < (1ll << 40)) x = x * 10 + (c - '0');
        s->at++;
    }
    *v = (int)(neg ? -x : x);
    return 1;
}

/* one record -> converted fields (4 = a record; 0 with nothing read = end of the file, what fscanf reports as EOF) */
static int bgs_record(bgs_t *s, cli_str_t *name, int *st, int *end, int *depth)
{
    if (!bgs_string(s, name)) return -1;
    if (!bgs_int(s, st)) return 1;
    if (!bgs_int(s, end)) return 2;
    if (!bgs_int(s, depth)) return 3;
    return 4;
}

void cli_host_get_depths(FILE *ft, FILE *fq, cli_host_cov_t *out)
{
    bgs_t a = {ft, (unsigned char *)cli_xmalloc(1 << 20), 0, 0, 0}, b = {fq, (unsigned char *)cli_xmalloc(1 << 20), 0, 0, 0};
    cli_str_t na = {0, 0, 0}, nb = {0, 0, 0};
    memset(out, 0, sizeof(*out));
    int32_t cap_ctg = 0;
    int64_t cap_pos = 0;
    int prev_pos = 0;
    double tot = 0, tot_mq = 0, tot_len = 0;
    for (;;) {
        int st1, end1, d1, st2, end2, d2;
        int r = bgs_record(&a, &na, &st1, &end1, &d1);
        if (r < 0) break;                                                              /* :205-207 */
        if (r != 4) {
            CLI_ERROR("The depth files should have 4 columns. Had %d.", r);            /* :209-212 */
            exit(EXIT_FAILURE);
        }
        r = bgs_record(&b, &nb, &st2, &end2, &d2);
        if (r < 0) {
            CLI_ERROR("%s", "The two files are not in the same order");               /* :214-217 */
            exit(EXIT_FAILURE);
        }
        if (r != 4) {
            CLI_ERROR("The depth files should have 4 columns. Had %d.", r);            /* :219-222 */
            exit(EXIT_FAILURE);
        }
        if (strcmp(na.s, nb.s) != 0 || st1 != st2 || end1 != end2) {
            CLI_ERROR("%s", "The two files are not in the same order");               /* :224-227 */
            exit(EXIT_FAILURE);
        }
        if (out->n_ctg == 0 || strcmp(na.s, out->names[out->n_ctg - 1]) != 0) {       /* :229-246 (prev_ctg starts as "") */
            if (out->n_ctg == 0 && na.l == 0) { /* cannot happen: %s never converts an empty token */ }
            if (out->n_ctg == cap_ctg) {
                cap_ctg = cap_ctg ? cap_ctg * 2 : 256;
                out->names = (char **)cli_xrealloc(out->names, (size_t)cap_ctg * sizeof(char *));
                out->lens = (int32_t *)cli_xrealloc(out->lens, (size_t)cap_ctg * sizeof(int32_t));
                out->first = (int64_t *)cli_xrealloc(out->first, ((size_t)cap_ctg + 1) * sizeof(int64_t));
            }
            out->names[out->n_ctg] = cli_xstrdup(na.s);
            out->lens[out->n_ctg] = 0;
            out->first[out->n_ctg] = out->n_pos;
            out->n_ctg++;
            prev_pos = 0;
        } else {
            if (prev_pos + 1 != st1) {
                CLI_ERROR("The depth files should be incremantal at one base resolution. Found %d to %d", prev_pos, st1);   /* :249-252 */
                exit(EXIT_FAILURE);
            }
            prev_pos++;
        }
        if (st1 + 1 != end1) {
            CLI_ERROR("The depth files should have end=start+1. Found %d to %d", st1, end1);   /* :256-259 */
            exit(EXIT_FAILURE);
        }
        if (d1 > 65535) {                                                             /* :261-268 */
            d1 = 65535;
            out->n_clamped++;
        }
        if (d2 > 65535) {
            d2 = 65535;
            out->n_clamped++;
        }
        if (out->lens[out->n_ctg - 1] == 0x7fffffff) {
            CLI_ERROR("contig %s has more than 2^31-1 positions", na.s);
            exit(EXIT_FAILURE);
        }
        if (out->n_pos == cap_pos) {
            cap_pos = cap_pos ? cap_pos * 2 : 1 << 20;
            out->depth = (uint16_t *)cli_xrealloc(out->depth, (size_t)cap_pos * sizeof(uint16_t));
            out->mq = (uint16_t *)cli_xrealloc(out->mq, (size_t)cap_pos * sizeof(uint16_t));
        }
        out->depth[out->n_pos] = (uint16_t)d1;                                        /* (a negative value wraps, as there) */
        out->mq[out->n_pos] = (uint16_t)d2;
        out->n_pos++;
        out->lens[out->n_ctg - 1]++;
        tot += d1;                                                                    /* :283-285 */
        tot_mq += d2;
        tot_len++;
    }
    if (out->first) out->first[out->n_ctg] = out->n_pos;
    out->sum_depth = tot;
    out->sum_mq = tot_mq;
    out->positions = tot_len;
    free(a.buf);
    free(b.buf);
    free(na.s);
    free(nb.s);
}

void cli_host_cov_free(cli_host_cov_t *c)
{
    for (int32_t i = 0; i < c->n_ctg; ++i) free(c->names[i]);
    free(c->names);
    free(c->lens);
    free(c->first);
    free(c->depth);
    free(c->mq);
    memset(c, 0, sizeof(*c));
}

/* ------------------------------------------------------------------------------------------------ get_regs + selection
 * src/boringbits_main.c:322-378 and the predicates of :425-445 / :463-481.  The window sums slide (what leaves, what
 * enters) in 32-bit unsigned arithmetic: the same value as the reference's `int` sums modulo 2^32. */
void cli_host_cov_select(const cli_host_cov_t *c, int w, int inc, int32_t lo, int32_t hi, float low_mq, int32_t edge_len, int32_t min_ctg_len, int boring,
                         cornetto_regrec_t **recs, int64_t *n_recs)
{
    cornetto_regrec_t *o = NULL;
    int64_t n = 0, cap = 0;
    for (int32_t ci = 0; ci < c->n_ctg; ++ci) {
        const int32_t len = c->lens[ci];
        if (boring ? !(len > min_ctg_len) : len < min_ctg_len) continue;                /* :467 / :428 */
        const uint16_t *dp = c->depth + c->first[ci], *mq = c->mq + c->first[ci];
        int32_t n_reg = (len - w + inc - 1) / inc + 1;                                 /* :338 */
        if (n_reg < 1) n_reg = 1;                                                      /* :339 */
        uint32_t sd = 0, sq = 0;
        int64_t a = 0, b = 0; /* the sums hold [a, b) */
        for (int32_t j = 0; j < n_reg; ++j) {
            const int64_t st = (int64_t)j * inc;
            int64_t end = st + w;
            if (end > len) end = len;                                                  /* :349-351 */
            /* (st < end: the caller has run the asserts of :353 / :368 over every contig — cornetto_regs_assert) */
            if (st >= b) {
                sd = sq = 0;
                a = b = st;
            }
            for (; a < st; ++a) {
                sd -= dp[a];
                sq -= mq[a];
            }
            for (; b < end; ++b) {
                sd += dp[b];
                sq += mq[b];
            }
            const int32_t depth = (int32_t)sd / (int32_t)(end - st), mqd = (int32_t)sq / (int32_t)(end - st);   /* :360-361 */
            const int fun = depth < lo || depth > hi || (mqd / (double)depth) < low_mq;                         /* :439 */
            const int sel = boring ? (st > edge_len && end < len - edge_len && !fun) : fun;                      /* :473-474 */
            if (!sel) continue;
            if (n == cap) {
                cap = cap ? cap * 2 : 4096;
                o = (cornetto_regrec_t *)cli_xrealloc(o, (size_t)cap * sizeof(*o));
            }
            o[n].ctg = ci;
            o[n].st = (int32_t)st;
            o[n].end = (int32_t)end;
            o[n].depth = depth;
            o[n].mq_depth = mqd;
            ++n;
        }
    }
    *recs = o;
    *n_recs = n;
}

/* the first three steps of scripts/create-cornetto.sh (:41-47) on the selected windows: `bedtools merge -d dist` of rows
 * that are already ordered (a row opens a new interval iff it starts more than dist behind the furthest end so far),
 * then the rows of at least min_len */
void cli_host_merge_windows(const cornetto_regrec_t *recs, int64_t n_recs, int32_t dist, int32_t min_len, cornetto_ivl_t **ivls, int64_t *n_ivls)
{
    cornetto_ivl_t *o = (cornetto_ivl_t *)cli_xmalloc(((size_t)n_recs + 1) * sizeof(*o));
    int64_t n = 0;
    for (int64_t i = 0; i < n_recs; ++i) {
        if (n > 0 && o[n - 1].ctg == recs[i].ctg && (int64_t)recs[i].st <= (int64_t)o[n - 1].finish + dist) {
            if (recs[i].end > o[n - 1].finish) o[n - 1].finish = recs[i].end;
        } else {
            if (n > 0 && o[n - 1].finish - o[n - 1].start < min_len) --n;
            o[n].ctg = recs[i].ctg;
            o[n].start = recs[i].st;
            o[n].finish = recs[i].end;
            ++n;
        }
    }
    if (n > 0 && o[n - 1].finish - o[n - 1].start < min_len) --n;
    *ivls = o;
    *n_ivls = n;
}

/* ------------------------------------------------------------------------------------------------ telobreaks
 * src/telomere_breaks.c:79-148: one bit per base for the low-complexity intervals, a second bit set for the runs of it
 * that hold a telomere row with its 100-base flanks; the runs of the second set are the output. */
static inline int bit_at(const uint64_t *w, int64_t i) { return (int)(w[i >> 6] >> (i & 63)) & 1; }

int cli_host_telobreaks(const int32_t *ctg_len, int32_t n_ctg, const cornetto_ivl_t *sd, int64_t n_sd, const cornetto_telrow_t *tel, int64_t n_tel,
                        cornetto_ivl_t **out, int64_t *n_out)
{
    uint64_t **low = (uint64_t **)cli_xmalloc(((size_t)n_ctg + 1) * sizeof(*low)), **fin = (uint64_t **)cli_xmalloc(((size_t)n_ctg + 1) * sizeof(*fin));
    for (int32_t c = 0; c < n_ctg; ++c) {
        low[c] = (uint64_t *)calloc((size_t)(ctg_len[c] / 64 + 2), 8);
        fin[c] = (uint64_t *)calloc((size_t)(ctg_len[c] / 64 + 2), 8);
        if (!low[c] || !fin[c]) {
            CLI_ERROR("%s", "out of memory");
            exit(EXIT_FAILURE);
        }
    }
    int rc = 0;
    for (int64_t i = 0; i < n_sd && !rc; ++i) {                                       /* :79-90 */
        if (sd[i].ctg < 0 || sd[i].ctg >= n_ctg) continue;
        /* (beyond the contig's end: sdust's own intervals at a contig's end reach there; the reference sets those bits and never reads them) */
        const int64_t fin_i = sd[i].finish > ctg_len[sd[i].ctg] ? ctg_len[sd[i].ctg] : sd[i].finish;
        if (sd[i].start < 0) rc = -1;
        else if (sd[i].start < fin_i) bits_set(low[sd[i].ctg], sd[i].start, fin_i);
    }
    for (int64_t i = 0; i < n_tel && !rc; ++i) {                                      /* :95-128 */
        if (tel[i].matched < 24 || tel[i].ctg < 0 || tel[i].ctg >= n_ctg) continue;   /* MIN_TEL, :10,:98 */
        const int64_t len = ctg_len[tel[i].ctg];
        if (tel[i].start < 0 || tel[i].end > len || tel[i].start >= tel[i].end) {
            rc = -1;
            break;
        }
        const uint64_t *b = low[tel[i].ctg];
        const int64_t a0 = tel[i].start - 100 < 0 ? 0 : tel[i].start - 100, a1 = tel[i].end + 100 > len ? len : tel[i].end + 100;   /* :102-103 */
        if (bits_in(b, a0, a1) != a1 - a0) continue;
        int64_t s = tel[i].start, e = tel[i].end;                                     /* :114-121 */
        while (s > 0 && bit_at(b, s - 1)) --s;
        while (e < len && bit_at(b, e)) ++e;
        bits_set(fin[tel[i].ctg], s, e);
    }
    cornetto_ivl_t *o = NULL;
    int64_t n = 0, cap = 0;
    for (int32_t c = 0; c < n_ctg && !rc; ++c) {                                      /* :133-148 */
        const int64_t len = ctg_len[c];
        for (int64_t i = 0; i < len; ++i) {
            if (!fin[c][i >> 6]) {
                i |= 63;
                continue;
            }
            if (!bit_at(fin[c], i)) continue;
            int64_t e = i;
            while (e < len && bit_at(fin[c], e)) ++e;
            if (n == cap) {
                cap = cap ? cap * 2 : 256;
                o = (cornetto_ivl_t *)cli_xrealloc(o, (size_t)cap * sizeof(*o));
            }
            o[n].ctg = c;
            o[n].start = (int32_t)(i - 1 < 0 ? 0 : i - 1);                             /* :139-141 */
            o[n].finish = (int32_t)(e - 1);
            ++n;
            i = e;
        }
    }
    for (int32_t c = 0; c < n_ctg; ++c) {
        free(low[c]);
        free(fin[c]);
    }
    free(low);
    free(fin);
    if (rc) {
        free(o);
        o = NULL;
        n = 0;
    }
    *out = o;
    *n_out = n;
    return rc;
}

/*
 * oracle/oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY; see oracle.h for the rules and the parity pin).
 *
 * A from-scratch, sequential C restatement of the reference's per-contig algorithms.  The algorithmic
 * STRUCTURE of the reference is kept on purpose (it doubles as the timed 1-core CPU baseline "port" in
 * bench.py): two whole-contig search passes for telofind, a byte mark array + 5x re-read for telowin,
 * the sequential sDUST recurrence with an explicit perfect-interval list, and O(n*w/inc) window sums.
 * All file:line citations are relative to /root/reference/.
 */
#define _POSIX_C_SOURCE 200809L
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------------
 * telofind — src/find_telomere.c
 * ---------------------------------------------------------------------------------------------- */

/* src/find_telomere.c:24-42 — reverse, complement A<->T C<->G, copy anything else */
void orc_revcomp(const char *motif, char *out)
{
    size_t k = strlen(motif);
    for (size_t i = 0; i < k; ++i) {
        char c = motif[k - 1 - i];
        out[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
    }
    out[k] = 0;
}

typedef struct { orc_hit_t *a; int64_t n, m; } hitvec_t;

static void hv_push(hitvec_t *v, int64_t s, int64_t e, int strand)
{
    if (v->n == v->m) {
        v->m = v->m ? v->m * 2 : 64;
        v->a = (orc_hit_t *)realloc(v->a, (size_t)v->m * sizeof(orc_hit_t));
    }
    v->a[v->n].start = s; v->a[v->n].end = e; v->a[v->n].strand = strand; v->a[v->n].pad = 0;
    v->n++;
}

/* One strand of src/find_telomere.c:49-58 (identical loop at :63-72): find the next occurrence at or
 * after pos (strstr), extend while the next k bytes equal the pattern (strncmp), emit, resume at end+1. */
static void one_strand(const char *q, int64_t len, const char *pat, int strand, hitvec_t *out)
{
    size_t k = strlen(pat);
    int64_t pos = 0;
    while (pos <= len) {
        const char *hit = strstr(q + pos, pat);      /* q is NUL-terminated at q[len] */
        if (!hit) break;
        pos = hit - q;
        int64_t start = pos;
        while (strncmp(q + pos, pat, k) == 0) pos += (int64_t)k;
        hv_push(out, start, pos, strand);
        pos++;                                       /* src/find_telomere.c:57 */
    }
}

int orc_telofind(const uint8_t *seq, int64_t len, const char *motif, orc_hit_t **hits, int64_t *n_hits)
{
    *hits = 0; *n_hits = 0;
    size_t k = strlen(motif);
    if (k == 0) return -1;                           /* the reference would never terminate */
    char *q = (char *)malloc((size_t)len + 1);
    if (!q) return -2;
    /* src/find_telomere.c:76-81 — toupper() of the whole contig ("C" locale: a-z only) */
    for (int64_t i = 0; i < len; ++i) {
        uint8_t c = seq[i];
        q[i] = (c >= 'a' && c <= 'z') ? (char)(c - 32) : (char)c;
    }
    q[len] = 0;
    hitvec_t v = {0, 0, 0};
    one_strand(q, len, motif, 0, &v);
    char *rc = (char *)malloc(k + 1);
    orc_revcomp(motif, rc);
    one_strand(q, len, rc, 1, &v);
    free(rc); free(q);
    *hits = v.a; *n_hits = v.n;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * telowin — src/telomere_windows.c
 * ---------------------------------------------------------------------------------------------- */

double orc_telowin_threshold(double threshold, double identity_percent)
{
    double identity = identity_percent / 100;        /* src/telomere_windows.c:53 */
    return threshold * pow(identity, 6);             /* :54 — always 6, whatever the motif length */
}

int orc_telowin(const orc_hit_t *hits, int64_t n_hits, int32_t contig_len, double thr_adj,
                orc_win_t **wins, int64_t *n_wins)
{
    *wins = 0; *n_wins = 0;
    uint8_t *b = (uint8_t *)calloc(contig_len > 0 ? (size_t)contig_len : 1, 1);   /* :72 */
    if (!b) return -2;
    for (int64_t h = 0; h < n_hits; ++h)             /* :75-79 */
        for (int64_t i = hits[h].start; i < hits[h].end; ++i) b[i] = 1;
    int64_t m = 0, n = 0; orc_win_t *w = 0;
    const int WIN = 1000;                            /* :13 */
    for (int i = 0; i <= contig_len; i += WIN / 5) { /* :31 */
        int car = 0;
        for (int j = i; j < i + WIN && j < contig_len; ++j) if (b[j]) car++;      /* :33-35 */
        int den = (i + WIN < contig_len) ? WIN : contig_len - i;                   /* :36 */
        if ((double)car / den >= thr_adj) {          /* :37 */
            if (n == m) { m = m ? m * 2 : 64; w = (orc_win_t *)realloc(w, (size_t)m * sizeof(orc_win_t)); }
            w[n].start = i; w[n].end = i + den; w[n].car = car; w[n].pad = 0; n++;
        }
        if (i + WIN >= contig_len) break;            /* :40 */
    }
    free(b);
    *wins = w; *n_wins = n;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * sdust — src/sdust/sdust.c
 * ---------------------------------------------------------------------------------------------- */

#define NWORD 64   /* SD_WTOT = 4^3, src/sdust/sdust.c:9-10 */

typedef struct { int start, finish, r, l; } pint_t;          /* :13-16 */

typedef struct {
    /* the sliding window of 3-mers (a FIFO of at most W-2 words), :42 kdq_t(int) *w */
    int *ring; int cap, front, size;
    int cw[NWORD], cv[NWORD];
    int rw, rv, L;
    pint_t *P; int nP, mP;                                   /* :43 */
    uint64_t *res; int nres, mres;                           /* :44 */
} sd_t;

static int nt4(uint8_t c)                                    /* :23-40 seq_nt4_table */
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default:
        if (c < 4) return c;                                 /* table rows 0: bytes 0..3 map to themselves */
        return 4;
    }
}

static int ring_at(const sd_t *s, int i) { return s->ring[(s->front + i) % s->cap]; }

/* :66-86 shift_window */
static void sd_shift_window(sd_t *s, int t, int T, int W)
{
    if (s->size >= W - 3 + 1) {                              /* :69 */
        int x = s->ring[s->front];
        s->front = (s->front + 1) % s->cap; s->size--;
        s->rw -= --s->cw[x];                                 /* :71 */
        if (s->L > s->size) { --s->L; s->rv -= --s->cv[x]; } /* :72-73 */
    }
    s->ring[(s->front + s->size) % s->cap] = t; s->size++;   /* :75 */
    ++s->L;
    s->rw += s->cw[t]++;                                     /* :77 */
    s->rv += s->cv[t]++;                                     /* :78 */
    if (s->cv[t] * 10 > T << 1) {                            /* :79 */
        int x;
        do {
            x = ring_at(s, s->size - s->L);                  /* :81 */
            s->rv -= --s->cv[x];
            --s->L;
        } while (x != t);
    }
}

/* :88-102 save_masked_regions */
static void sd_save(sd_t *s, int start)
{
    if (s->nP == 0 || s->P[s->nP - 1].start >= start) return;        /* :92 */
    pint_t *p = &s->P[s->nP - 1];
    int saved = 0;
    if (s->nres) {
        int st = (int)(s->res[s->nres - 1] >> 32), f = (int)(uint32_t)s->res[s->nres - 1];
        if (p->start <= f) {                                 /* :96 overlapping or adjacent */
            saved = 1;
            s->res[s->nres - 1] = (uint64_t)st << 32 | (uint32_t)(f > p->finish ? f : p->finish);
        }
    }
    if (!saved) {                                            /* :99 */
        if (s->nres == s->mres) { s->mres = s->mres ? s->mres * 2 : 64; s->res = (uint64_t *)realloc(s->res, (size_t)s->mres * 8); }
        s->res[s->nres++] = (uint64_t)p->start << 32 | (uint32_t)p->finish;
    }
    int i;
    for (i = s->nP - 1; i >= 0 && s->P[i].start < start; --i) {}      /* :100 */
    s->nP = i + 1;
}

/* :104-128 find_perfect */
static void sd_find_perfect(sd_t *s, int T, int start)
{
    int c[NWORD], r = s->rv, max_r = 0, max_l = 0;
    memcpy(c, s->cv, sizeof(c));
    for (int i = s->size - s->L - 1; i >= 0; --i) {          /* :108 */
        int t = ring_at(s, i);
        r += c[t]++;
        int new_r = r, new_l = s->size - i - 1;              /* :111 */
        if (new_r * 10 > T * new_l) {                        /* :112 */
            int j;
            for (j = 0; j < s->nP && s->P[j].start >= i + start; ++j) {          /* :113 */
                pint_t *p = &s->P[j];
                if (max_r == 0 || p->r * max_l > max_r * p->l) { max_r = p->r; max_l = p->l; }
            }
            if (max_r == 0 || new_r * max_l >= max_r * new_l) {                  /* :118 */
                max_r = new_r; max_l = new_l;
                if (s->nP == s->mP) { s->mP = s->mP ? s->mP * 2 : 64; s->P = (pint_t *)realloc(s->P, (size_t)s->mP * sizeof(pint_t)); }
                memmove(&s->P[j + 1], &s->P[j], (size_t)(s->nP - j) * sizeof(pint_t));
                ++s->nP;
                s->P[j].start = i + start; s->P[j].finish = s->size + 2 + start;  /* :123 */
                s->P[j].r = new_r; s->P[j].l = new_l;
            }
        }
    }
}

uint64_t *orc_sdust(const uint8_t *seq, int32_t l_seq, int32_t T, int32_t W, int32_t *n)
{
    sd_t s; memset(&s, 0, sizeof(s));
    s.cap = (W > 3 ? W : 3) + 2;
    s.ring = (int *)malloc((size_t)s.cap * sizeof(int));
    int l = 0; unsigned t = 0;
    for (int i = 0; i <= l_seq; ++i) {                       /* :141 — note <= : sentinel at i == l_seq */
        int b = i < l_seq ? nt4(seq[i]) : 4;
        if (b < 4) {
            ++l; t = (t << 2 | (unsigned)b) & (NWORD - 1);   /* :144 */
            if (l >= 3) {
                int start = (l - W > 0 ? l - W : 0) + (i + 1 - l);               /* :146 */
                sd_save(&s, start);
                sd_shift_window(&s, (int)t, T, W);
                if (s.rw * 10 > s.L * T) sd_find_perfect(&s, T, start);           /* :149-150 */
            }
        } else {
            int start = (l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l);           /* :152 */
            while (s.nP) sd_save(&s, start++);               /* :153 */
            l = 0; t = 0;                                    /* :154 — window and counters NOT reset */
        }
    }
    free(s.ring); free(s.P);
    *n = s.nres;
    if (!s.res) s.res = (uint64_t *)malloc(8);
    return s.res;
}

/* ------------------------------------------------------------------------------------------------
 * (no)boringbits window stage — src/boringbits_main.c
 * ---------------------------------------------------------------------------------------------- */

int32_t orc_n_reg(int32_t length, int32_t window_size, int32_t window_inc)
{
    int32_t n = (length - window_size + window_inc - 1) / window_inc + 1;   /* :338, C truncation */
    return n < 1 ? 1 : n;                                                    /* :339 */
}

/* the asserts of get_regs(): the loop of :346-353 and the two lines behind it (:368-369), statement for statement, without the sums.
 * 0 = the reference gets through, else the line of the assert that raises SIGABRT */
int orc_regs_assert(int32_t length, int32_t window_size, int32_t window_inc)
{
    int32_t n = orc_n_reg(length, window_size, window_inc);
    int32_t st = 0, end = 0;
    for (int32_t j = 0; j < n; ++j) {                        /* :346 */
        st = j * window_inc;                                 /* :347 */
        end = st + window_size;                              /* :348 */
        if (end > length) end = length;                      /* :349-351 */
        if (!(st < end)) return 353;                         /* :353 */
    }
    if (!(end == length)) return 368;                        /* :368 */
    if (!(st < end)) return 369;                             /* :369 */
    return 0;
}

void orc_get_regs(const uint16_t *depth, const uint16_t *mq_depth, int32_t length,
                  int32_t window_size, int32_t window_inc, orc_reg_t *regs)
{
    int32_t n = orc_n_reg(length, window_size, window_inc);
    for (int32_t j = 0; j < n; ++j) {                        /* :346 */
        int32_t st = j * window_inc, end = st + window_size;
        if (end > length) end = length;                      /* :349-351 */
        /* int accumulators as in the reference (:354-359); unsigned here so that an (absurdly large
         * -w) overflow wraps the way gcc -O2 makes the reference wrap instead of being undefined */
        uint32_t d = 0, q = 0;
        for (int32_t k = st; k < end; ++k) { d += depth[k]; q += mq_depth[k]; }
        regs[j].st = st; regs[j].end = end;
        regs[j].depth = (int32_t)d / (end - st);             /* :360 */
        regs[j].mq_depth = (int32_t)q / (end - st);          /* :361 */
    }
}

int32_t orc_mean_depth(double tot, double n) { return (int32_t)round(tot / n); }     /* :293-294 */

int32_t orc_threshold(float factor, int32_t mean) { return (int32_t)round(factor * mean); }  /* :518-519 */

int orc_is_fun(int32_t depth, int32_t mq_depth, int32_t lo, int32_t hi, float low_mq)
{
    return depth < lo || depth > hi || (mq_depth / (double)depth) < low_mq;      /* :439 */
}

/* ------------------------------------------------------------------------------------------------
 * bigenough — src/bigenough_main.c
 * ---------------------------------------------------------------------------------------------- */

int orc_bigenough_keep(int32_t covlen, int32_t start, int32_t end, int32_t threshold)
{
    /* :206  r->covlen > (r->end - r->start) * threshold / 100   with int operands: the product wraps
     * (two's complement, as gcc -O2 compiles it) and the division truncates toward zero */
    int32_t prod = (int32_t)((uint32_t)(end - start) * (uint32_t)threshold);
    return covlen > prod / 100;
}

/* ------------------------------------------------------------------------------------------------
 * telobreaks — src/telomere_breaks.c
 * ---------------------------------------------------------------------------------------------- */

static void tb_set(unsigned char *b, int i) { b[i / 8] |= (unsigned char)(1 << (i % 8)); }     /* :33-35 */
static int tb_get(const unsigned char *b, int i) { return b[i / 8] & (1 << (i % 8)); }           /* :37-39 */

int orc_telobreaks(const int32_t *ctg_len, int32_t n_ctg, const orc_span_t *sd, int64_t n_sd,
                   const orc_telrow_t *tel, int64_t n_tel, orc_span_t **out, int64_t *n_out)
{
    unsigned char **bits = (unsigned char **)calloc((size_t)n_ctg + 1, sizeof(*bits));
    unsigned char **fin = (unsigned char **)calloc((size_t)n_ctg + 1, sizeof(*fin));
    orc_span_t *o = NULL;
    int64_t no = 0, cap = 0;
    int rc = 0;
    for (int32_t c = 0; c < n_ctg; ++c) {                                 /* create_scaffold, :25-31 */
        bits[c] = (unsigned char *)calloc((size_t)ceil(ctg_len[c] / 8.0) + 1, 1);
        fin[c] = (unsigned char *)calloc((size_t)ceil(ctg_len[c] / 8.0) + 1, 1);
    }
    for (int64_t i = 0; i < n_sd && rc == 0; ++i) {                       /* :79-90 */
        const orc_span_t v = sd[i];
        if (v.ctg < 0 || v.ctg >= n_ctg) continue;
        /* (an interval that ends beyond the contig — sdust prints such intervals at a contig's end —: the reference sets those bits beyond its
         * bitset (:85) and never reads them (:103,:118,:136): here they are not set) */
        if (v.start < 0) { rc = -1; break; }
        for (int j = v.start; j < v.end && j < ctg_len[v.ctg]; ++j) tb_set(bits[v.ctg], j);
    }
    for (int64_t i = 0; i < n_tel && rc == 0; ++i) {                      /* :95-128 */
        const orc_telrow_t t = tel[i];
        if (t.matched < 24) continue;                                     /* MIN_TEL, :10,:98 */
        if (t.ctg < 0 || t.ctg >= n_ctg) continue;
        const int length = ctg_len[t.ctg];
        if (t.start < 0 || t.end > length || t.start >= t.end) { rc = -1; break; }
        const unsigned char *b = bits[t.ctg];
        int rStart = t.start - 100 < 0 ? 0 : t.start - 100;               /* :102 */
        int rEnd = t.end + 100 > length ? length : t.end + 100;          /* :103 */
        int all_set = 1;
        for (int j = rStart; j < rEnd; ++j)
            if (!tb_get(b, j)) { all_set = 0; break; }
        if (!all_set) continue;
        rStart = t.start;                                                 /* :114-121 */
        while (rStart > 0 && tb_get(b, rStart - 1)) rStart--;
        rEnd = t.end;
        while (rEnd < length && tb_get(b, rEnd)) rEnd++;
        for (int j = rStart; j < rEnd; ++j) tb_set(fin[t.ctg], j);
    }
    for (int32_t c = 0; c < n_ctg && rc == 0; ++c) {                      /* :133-148, one contig */
        const int length = ctg_len[c];
        for (int i = 0; i < length; ++i) {
            if (tb_get(fin[c], i)) {
                int end = i;
                while (end < length && tb_get(fin[c], end)) ++end;
                i = i - 1 < 0 ? 0 : i - 1;
                if (no == cap) {
                    cap = cap ? cap * 2 : 64;
                    o = (orc_span_t *)realloc(o, (size_t)cap * sizeof(*o));
                }
                o[no].ctg = c; o[no].start = i; o[no].end = end - 1;
                ++no;
                i = end;
            }
        }
    }
    for (int32_t c = 0; c < n_ctg; ++c) { free(bits[c]); free(fin[c]); }
    free(bits); free(fin);
    if (rc != 0) { free(o); o = NULL; no = 0; }
    if (!o) o = (orc_span_t *)malloc(sizeof(*o));
    *out = o;
    *n_out = no;
    return rc;
}

/* ---- khash v0.2.8 string map, insertion only: where the keys end up ------------------------------------- */
static uint32_t kh_x31(const char *s)                                     /* src/khash.h:395-400 */
{
    uint32_t h = (uint32_t)(int)*s;                                       /* char is signed on x86-64 */
    if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)(int)*s;
    return h;
}

int32_t orc_khash_order(const char *const *names, int32_t n, int32_t *slot, int32_t *order)
{
    /* keys[b] = id stored in bucket b, or -1 (empty).  No deletions happen in telobreaks. */
    uint32_t n_buckets = 0, size = 0, upper = 0;
    int32_t *keys = NULL, n_ids = 0;
    const char **id_name = (const char **)malloc(((size_t)n + 1) * sizeof(*id_name));
    for (int32_t it = 0; it < n; ++it) {
        if (size >= upper) {                                              /* kh_put, :308-318 (n_occupied == size) */
            uint32_t want = n_buckets > (size << 1) ? n_buckets - 1 : n_buckets + 1, nb = want;
            /* kh_resize, :244-305 */
            --nb; nb |= nb >> 1; nb |= nb >> 2; nb |= nb >> 4; nb |= nb >> 8; nb |= nb >> 16; ++nb;   /* kroundup32 */
            if (nb < 4) nb = 4;
            if (!(size >= (uint32_t)(nb * 0.77 + 0.5))) {
                int32_t *nk = (int32_t *)malloc((size_t)(nb > n_buckets ? nb : n_buckets) * sizeof(*nk));
                unsigned char *newocc = (unsigned char *)calloc(nb, 1);    /* new_flags */
                unsigned char *oldocc = (unsigned char *)calloc(n_buckets + 1, 1);
                for (uint32_t j = 0; j < (nb > n_buckets ? nb : n_buckets); ++j) nk[j] = j < n_buckets ? keys[j] : -1;
                for (uint32_t j = 0; j < n_buckets; ++j) oldocc[j] = keys[j] >= 0;
                for (uint32_t j = 0; j != n_buckets; ++j) {
                    if (!oldocc[j]) continue;
                    int32_t key = nk[j];
                    const uint32_t mask = nb - 1;
                    oldocc[j] = 0;                                        /* __ac_set_isdel_true */
                    for (;;) {                                            /* kick-out process */
                        uint32_t i = kh_x31(id_name[key]) & mask, step = 0;
                        while (newocc[i]) i = (i + (++step)) & mask;
                        newocc[i] = 1;
                        if (i < n_buckets && oldocc[i]) {
                            const int32_t tmp = nk[i]; nk[i] = key; key = tmp;
                            oldocc[i] = 0;
                        } else {
                            nk[i] = key;
                            break;
                        }
                    }
                }
                for (uint32_t j = 0; j < nb; ++j) if (!newocc[j]) nk[j] = -1;
                free(keys); free(newocc); free(oldocc);
                keys = nk;
                n_buckets = nb;
                upper = (uint32_t)(n_buckets * 0.77 + 0.5);
            }
        }
        {                                                                 /* :320-347 */
            const uint32_t mask = n_buckets - 1;
            uint32_t i = kh_x31(names[it]) & mask, step = 0;
            while (keys[i] >= 0 && strcmp(id_name[keys[i]], names[it]) != 0) i = (i + (++step)) & mask;
            if (keys[i] < 0) {
                id_name[n_ids] = names[it];
                keys[i] = n_ids++;
                ++size;
            }
            slot[it] = keys[i];
        }
    }
    int32_t k = 0;
    for (uint32_t b = 0; b < n_buckets; ++b) if (keys[b] >= 0) order[k++] = keys[b];
    free(keys); free((void *)id_name);
    return n_ids;
}

/* ------------------------------------------------------------------------------------------------
 * panel interval stage — scripts/create-cornetto.sh:44-66 (PARITY UNPINNED, see oracle.h)
 * ---------------------------------------------------------------------------------------------- */
static int span_cmp(const void *a, const void *b)
{
    const orc_span_t *x = (const orc_span_t *)a, *y = (const orc_span_t *)b;
    if (x->ctg != y->ctg) return x->ctg < y->ctg ? -1 : 1;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    return x->end < y->end ? -1 : (x->end > y->end);
}

int orc_ivl_merge(const orc_span_t *in, int64_t n, int32_t dist, orc_span_t **out, int64_t *n_out)
{
    orc_span_t *v = (orc_span_t *)malloc(((size_t)n + 1) * sizeof(*v));
    int64_t m = 0;
    if (n > 0) memcpy(v, in, (size_t)n * sizeof(*v));
    qsort(v, (size_t)n, sizeof(*v), span_cmp);
    for (int64_t i = 0; i < n; ++i) {
        /* bedtools merge: a feature joins the current cluster when it starts at most `dist` behind the cluster's end */
        if (m > 0 && v[m - 1].ctg == v[i].ctg && (int64_t)v[i].start - (int64_t)v[m - 1].end <= dist) {
            if (v[i].end > v[m - 1].end) v[m - 1].end = v[i].end;
        } else {
            v[m++] = v[i];
        }
    }
    *out = v;
    *n_out = m;
    return 0;
}

int orc_panel_boring(const int32_t *ctg_len, int32_t n_ctg, const orc_span_t *fun, int64_t n_fun, const orc_span_t *lowq, int64_t n_lowq,
                     int32_t min_lowq_len, int32_t extend, int32_t edge_len, int32_t merge_dist, int32_t min_ctg_len,
                     int32_t extend_right, int32_t extend_gate, orc_span_t **out, int64_t *n_out)
{
    orc_span_t *v = (orc_span_t *)malloc(((size_t)n_fun + (size_t)n_lowq + 2 * (size_t)n_ctg + 1) * sizeof(*v));
    int64_t n = 0;
    for (int64_t i = 0; i < n_fun; ++i) v[n++] = fun[i];                                   /* 3_tmp.bed */
    for (int64_t i = 0; i < n_lowq; ++i)                                                   /* :50  awk '($3-$2)>=8000' */
        if ((int64_t)lowq[i].end - lowq[i].start >= min_lowq_len) v[n++] = lowq[i];
    for (int64_t i = 0; i < n; ++i)                                                        /* :53  if($2>40000){$2-40000, $3+40000}; recreate :38 if($2>50000){$2-40000, $3+50000} */
        if (v[i].start > extend_gate) { v[i].start -= extend; v[i].end += extend_right; }
    for (int32_t c = 0; c < n_ctg; ++c)                                                    /* :56  if(($3-$2)>200000) two edge rows */
        if (ctg_len[c] > edge_len) {
            v[n].ctg = c; v[n].start = 0; v[n].end = edge_len; ++n;
            v[n].ctg = c; v[n].start = ctg_len[c] - edge_len; v[n].end = ctg_len[c]; ++n;
        }
    orc_span_t *m = NULL;
    int64_t nm = 0;
    orc_ivl_merge(v, n, merge_dist, &m, &nm);                                             /* :59 */
    free(v);
    /* :62 bedtools subtract -a assembly -b merged; :65-66 minus every contig shorter than min_ctg_len */
    orc_span_t *o = (orc_span_t *)malloc(((size_t)nm + (size_t)n_ctg + 1) * sizeof(*o));
    int64_t no = 0;
    for (int32_t c = 0; c < n_ctg; ++c) {
        if (ctg_len[c] < min_ctg_len || ctg_len[c] <= 0) continue;
        int32_t from = 0;                                  /* first base of the contig not yet covered by a merged row */
        for (int64_t j = 0; j < nm; ++j) {
            if (m[j].ctg != c) continue;
            if (m[j].end <= from || m[j].start >= ctg_len[c]) continue;
            if (m[j].start > from) { o[no].ctg = c; o[no].start = from; o[no].end = m[j].start; ++no; }
            from = m[j].end;
        }
        if (from < ctg_len[c]) { o[no].ctg = c; o[no].start = from; o[no].end = ctg_len[c]; ++no; }
    }
    free(m);
    *out = o;
    *n_out = no;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * FASTA/FASTQ record framing — src/kseq.h (klib kseq as vendored by the reference)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { const uint8_t *t; int64_t n, i; } fxcur_t;
typedef struct { char *s; int64_t l, m; } fxstr_t;

static void fxs_put(fxstr_t *s, const uint8_t *src, int64_t k)
{
    if (s->l + k + 1 > s->m) {
        s->m = (s->l + k + 1) * 2;
        s->s = (char *)realloc(s->s, (size_t)s->m);
    }
    if (k) memcpy(s->s + s->l, src, (size_t)k);
    s->l += k;
    s->s[s->l] = 0;
}

static int fx_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }   /* isspace() of the C locale */

/* ks_getuntil2, src/kseq.h:93-141, for the two delimiters kseq_read uses: line = 0 white space (KS_SEP_SPACE),
 * line = 1 newline (KS_SEP_LINE).  Appends; returns -1 when nothing is left at entry (:98), else the length; *dret =
 * the delimiter met, 0 when the text ended first (:96,:127-130). */
static int64_t fx_until(fxcur_t *c, int line, fxstr_t *s, int *dret)
{
    if (dret) *dret = 0;
    if (c->i >= c->n) return -1;
    int64_t j = c->i;
    if (line) {
        const uint8_t *nl = (const uint8_t *)memchr(c->t + j, '\n', (size_t)(c->n - j));
        j = nl ? nl - c->t : c->n;
    } else {
        while (j < c->n && !fx_space(c->t[j])) ++j;
    }
    fxs_put(s, c->t + c->i, j - c->i);
    if (j < c->n && dret) *dret = c->t[j];
    c->i = j < c->n ? j + 1 : c->n;
    if (line && s->l > 1 && s->s[s->l - 1] == '\r') s->s[--s->l] = 0;      /* :138 */
    return s->l;
}

int orc_fastx_parse(const uint8_t *text, int64_t n, orc_fxrec_t **recs, int64_t *n_recs)
{
    fxcur_t c = {text, n, 0};
    fxstr_t name = {0, 0, 0}, com = {0, 0, 0}, seq = {0, 0, 0}, qual = {0, 0, 0};
    orc_fxrec_t *out = NULL;
    int64_t cnt = 0, cap = 0;
    int last = 0, rc = -1;
    fxs_put(&name, NULL, 0); fxs_put(&com, NULL, 0); fxs_put(&seq, NULL, 0); fxs_put(&qual, NULL, 0);
    for (;;) {
        int ch, d;
        if (last == 0) {                                                        /* :189-193 jump to the next header */
            while (c.i < c.n && c.t[c.i] != '>' && c.t[c.i] != '@') ++c.i;
            if (c.i >= c.n) { rc = -1; break; }
            last = c.t[c.i++];
        }
        name.l = com.l = seq.l = qual.l = 0;
        name.s[0] = com.s[0] = seq.s[0] = qual.s[0] = 0;
        if (fx_until(&c, 0, &name, &d) < 0) { rc = -1; break; }                 /* :195 */
        if (d != '\n') fx_until(&c, 1, &com, NULL);                             /* :196 */
        ch = -1;
        while (c.i < c.n) {                                                     /* :201-205 */
            ch = c.t[c.i++];
            if (ch == '>' || ch == '+' || ch == '@') break;
            if (ch != '\n') {
                const uint8_t b = (uint8_t)ch;
                fxs_put(&seq, &b, 1);
                fx_until(&c, 1, &seq, NULL);
            }
            ch = -1;
        }
        last = (ch == '>' || ch == '@') ? ch : last;                            /* :206 */
        int has_qual = 0;
        if (ch == '+') {                                                        /* :214-223 */
            int got_nl = 0;
            while (c.i < c.n) if (c.t[c.i++] == '\n') { got_nl = 1; break; }
            if (!got_nl) { rc = -2; break; }
            while (fx_until(&c, 1, &qual, NULL) >= 0 && qual.l < seq.l) {}
            last = 0;
            if (seq.l != qual.l) { rc = -2; break; }
            has_qual = 1;
        } else if (ch == -1) {
            last = 0;      /* end of text: the next call finds no header and returns -1 (no state to keep) */
        }
        if (cnt == cap) {
            cap = cap ? cap * 2 : 64;
            out = (orc_fxrec_t *)realloc(out, (size_t)cap * sizeof(*out));
        }
        orc_fxrec_t *r = &out[cnt++];
        char *blk = (char *)malloc((size_t)(name.l + com.l + seq.l + qual.l + 4));
        r->name = blk;                      memcpy(r->name, name.s, (size_t)name.l + 1);
        r->comment = r->name + name.l + 1;  memcpy(r->comment, com.s, (size_t)com.l + 1);
        r->seq = r->comment + com.l + 1;    memcpy(r->seq, seq.s, (size_t)seq.l + 1);
        r->qual = NULL;
        if (has_qual) { r->qual = r->seq + seq.l + 1; memcpy(r->qual, qual.s, (size_t)qual.l + 1); }
        r->name_l = name.l; r->comment_l = com.l; r->l = seq.l; r->qual_l = has_qual ? qual.l : 0;
        if (ch == -1 && c.i >= c.n && last == 0) { rc = -1; break; }
    }
    free(name.s); free(com.s); free(seq.s); free(qual.s);
    *recs = out;
    *n_recs = cnt;
    return rc;
}

void orc_fastx_free(orc_fxrec_t *recs, int64_t n_recs)
{
    for (int64_t i = 0; i < n_recs; ++i) free(recs[i].name);
    free(recs);
}

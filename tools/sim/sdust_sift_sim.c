/* tools/sim/sdust_sift_sim.c — CPU model of the round-3 sdust decomposition (sift -> filter -> resolve), checked against
 * the oracle (oracle/oracle.c: orc_sdust, the sequential restatement of src/sdust/sdust.c:66-160).
 *
 * What is modelled:
 *   In a stretch of plain A/C/G/T the reference's result is the union of the intervals find_perfect inserts, and
 *   an interval [s, i] (first base of its first word, last base of its last word) is inserted iff it is "perfect":
 *   10 r > T l and r / l >= the ratio of every sub-interval (l = words - 1 <= W - 3, r = equal-word pairs).
 *   K1 "sift"    position-parallel necessary condition: the word ending at i has more than T / 10 equal words among the
 *                W - 3 words before it (an inserted interval gains c > T / 10 pairs with its last word).
 *   K2 "filter"  per sifted position, one walk over the suffixes: is there a suffix that is a candidate, whose ratio is
 *                >= the ratio of every shorter suffix, >= the ratio without its last word, and with c > T / 10 ?
 *   K3 "resolve" per chunk, sequential over the positions K2 kept, from an EMPTY P at chunk start - (W - 2): every
 *                entry whose start is >= the first processed step - 2 is exact; entries are recorded by the time they
 *                leave the window (start + W), exactly the rule the chunk kernel of rounds 1-2 uses.
 *   Chunks that see a non-ACGT byte within [start - 2W, end) are left to the sequential model ("slow path" = what
 *   sdust_w64 computes: the reference's process, recorded by save time).
 *
 * build: gcc -O2 -o /tmp/sift_sim tools/sim/sdust_sift_sim.c oracle/oracle.c -Ioracle -lm -lz
 * usage: /tmp/sift_sim [n_bases] [seeds] [T] [W] [chunk]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static uint64_t rng_s;
static uint64_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

static int code_of(uint8_t c)
{
    switch (c) {
        case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3;
        default: return c < 4 ? c : 4;
    }
}

typedef struct { int32_t s, f, tm; } ivl_t;
typedef struct { ivl_t *a; int64_t n, m; } ivlv_t;
static void push(ivlv_t *v, int32_t s, int32_t f)
{
    if (v->n == v->m) { v->m = v->m ? v->m * 2 : 1024; v->a = realloc(v->a, v->m * sizeof(ivl_t)); }
    v->a[v->n].s = s; v->a[v->n].f = f; v->a[v->n].tm = 0; ++v->n;
}
static int cmp_ivl(const void *x, const void *y)
{
    const ivl_t *a = x, *b = y;
    return a->s < b->s ? -1 : a->s > b->s ? 1 : (a->f < b->f ? -1 : a->f > b->f);
}
/* canonical union with the reference's rule: merge when start <= previous finish (src/sdust/sdust.c:94-98) */
static int64_t canon(ivlv_t *v)
{
    qsort(v->a, v->n, sizeof(ivl_t), cmp_ivl);
    int64_t o = 0;
    for (int64_t i = 0; i < v->n; ++i) {
        if (o && v->a[i].s <= v->a[o - 1].f) { if (v->a[i].f > v->a[o - 1].f) v->a[o - 1].f = v->a[i].f; }
        else v->a[o++] = v->a[i];
    }
    v->n = o;
    return o;
}

/* ---- sequential model with save times: the reference's process with P as "newest entry per start" slots ---------- */
typedef struct { int r, l; int used; } slot_t;
static void sequential_by_time(const uint8_t *seq, int len, int T, int W, ivlv_t *out)
{
    /* plain restatement (window as array of words, find_perfect over every suffix, no gate) run over the whole contig;
     * records the intervals saved at times in [t_from, t_to) (+ time == len when last) */
    const int CAPW = W - 2;
    int *win = malloc(sizeof(int) * (len + 8));        /* all pushed words; window = last min(n, CAPW) */
    int nw = 0, l = 0;
    unsigned t = 0;
    /* P: list of (start, r, l) sorted by descending start, as the reference keeps it */
    int pcap = 1024, pn = 0;
    int (*P)[3] = malloc(sizeof(int[3]) * pcap);
#define SAVE(startv, timev)                                                                              \
    do {                                                                                                 \
        if (pn > 0 && P[pn - 1][0] < (startv)) {                                                         \
            { push(out, P[pn - 1][0], P[pn - 1][0] + P[pn - 1][2] + 3); out->a[out->n - 1].tm = (timev); }  \
            int q = pn - 1;                                                                              \
            while (q >= 0 && P[q][0] < (startv)) --q;                                                    \
            pn = q + 1;                                                                                  \
        }                                                                                                \
    } while (0)
    for (int i = 0; i <= len; ++i) {
        const int b = i < len ? code_of(seq[i]) : 4;
        if (b < 4) {
            ++l;
            t = (t << 2 | (unsigned)b) & 63u;
            if (l >= 3) {
                const int start = (l - W > 0 ? l - W : 0) + (i + 1 - l);
                SAVE(start, i);
                win[nw++] = (int)t;
                const int size = nw < CAPW ? nw : CAPW;
                int c[64] = {0}, r = 0, max_r = 0, max_l = 0;
                for (int k = size - 1; k >= 0; --k) {
                    const int tt = win[nw - size + k];
                    r += c[tt]++;
                    const int new_l = size - k - 1;
                    if (r * 10 > T * new_l) {
                        int j;
                        for (j = 0; j < pn && P[j][0] >= k + start; ++j)
                            if (max_r == 0 || P[j][1] * max_l > max_r * P[j][2]) { max_r = P[j][1]; max_l = P[j][2]; }
                        if (max_r == 0 || r * max_l >= max_r * new_l) {
                            max_r = r; max_l = new_l;
                            if (pn == pcap) { pcap *= 2; P = realloc(P, sizeof(int[3]) * pcap); }
                            memmove(&P[j + 1], &P[j], sizeof(int[3]) * (pn - j));
                            ++pn;
                            P[j][0] = k + start; P[j][1] = r; P[j][2] = new_l;
                        }
                    }
                }
            }
        } else {
            int start = (l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l);
            while (pn) { SAVE(start, i); ++start; }
            l = 0; t = 0;
        }
    }
    free(win); free(P);
}

/* ---- the new decomposition ---------------------------------------------------------------------------------- */
static unsigned long long st_pos, st_k1, st_k2a, st_k2, st_k3steps, st_k3cand, st_k3ins, st_hot;

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 2000000;
    const int seeds = argc > 2 ? atoi(argv[2]) : 3;
    const int T = argc > 3 ? atoi(argv[3]) : 20;
    const int W = argc > 4 ? atoi(argv[4]) : 64;
    const int chunk = argc > 5 ? atoi(argv[5]) : 1792;
    const int mode = argc > 6 ? atoi(argv[6]) : 0;       /* 0 mixed, 1 pure random, 2 repeat rich */
    const int CAPW = W - 2;
    const int thr = T / 10 + 1;                          /* c > T / 10 */
    int bad = 0;
    for (int sd = 1; sd <= seeds; ++sd) {
        rng_s = 0x9E3779B97F4A7C15ull * (unsigned)sd + 12345;
        uint8_t *seq = malloc(n + 64);
        for (int i = 0; i < n; ++i) seq[i] = "ACGT"[rnd() & 3];
        if (mode != 1) {
            /* plant: tandem repeats of unit 1-7, length 8-400; satellites; N runs; lower case; odd bytes */
            const int feats = mode == 2 ? n / 300 : n / 3000;
            for (int f = 0; f < feats; ++f) {
                const int p = (int)(rnd() % (unsigned)n);
                const int kind = (int)(rnd() % 10);
                if (kind < 6) {
                    const int ul = 1 + (int)(rnd() % 7), L = 8 + (int)(rnd() % (kind < 2 ? 400 : 60));
                    char u[8];
                    for (int j = 0; j < ul; ++j) u[j] = "ACGT"[rnd() & 3];
                    for (int j = 0; j < L && p + j < n; ++j) seq[p + j] = (rnd() % 50 == 0) ? "ACGT"[rnd() & 3] : u[j % ul];
                } else if (kind < 8) {
                    const int L = 1 + (int)(rnd() % (kind == 6 ? 5 : 300));
                    for (int j = 0; j < L && p + j < n; ++j) seq[p + j] = 'N';
                } else if (kind == 8) {
                    const int L = 1 + (int)(rnd() % 200);
                    for (int j = 0; j < L && p + j < n; ++j) seq[p + j] |= 0x20;
                } else {
                    seq[p] = "RYKMSWBDHVN-*\001\003"[rnd() % 15];
                }
            }
            if (mode == 2 || sd % 2 == 0) {          /* one long satellite */
                const int p = n / 3, L = n / 10;
                for (int j = 0; j < L; ++j) seq[p + j] = (rnd() % 50 == 0) ? "ACGT"[rnd() & 3] : "CATTC"[j % 5];
            }
        }
        /* oracle */
        int32_t n_ref = 0;
        uint64_t *ref = orc_sdust(seq, n, T, W, &n_ref);

        /* words */
        uint8_t *word = calloc(n + 64, 1), *isw = calloc(n + 64, 1);
        {
            int l = 0; unsigned t = 0;
            for (int i = 0; i < n; ++i) {
                const int b = code_of(seq[i]);
                if (b < 4) { ++l; t = (t << 2 | (unsigned)b) & 63u; if (l >= 3) { isw[i] = 1; word[i] = (uint8_t)t; } }
                else { l = 0; t = 0; }
            }
        }
        /* chunk table + slow flags: any non-ACGT letter in [start - 2W, end) */
        const int nch = (n + chunk - 1) / chunk;
        uint8_t *slow = calloc(nch, 1);
        int *bad_prefix = malloc(sizeof(int) * (n + 1));
        bad_prefix[0] = 0;
        for (int i = 0; i < n; ++i) {
            const uint8_t c = seq[i];
            const int plain = c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'a' || c == 'c' || c == 'g' || c == 't';
            bad_prefix[i + 1] = bad_prefix[i] + !plain;
        }
        for (int k = 0; k < nch; ++k) {
            const int a = k * chunk - 2 * W > 0 ? k * chunk - 2 * W : 0, b = (k + 1) * chunk < n ? (k + 1) * chunk : n;
            slow[k] = bad_prefix[b] - bad_prefix[a] != 0;
            if (n < 256) slow[k] = 1;
        }
        /* K1 (sift) + L1 + L2 -> S.  ct[i] = equal words among the W - 3 words before i */
        uint8_t *S = calloc(n + 64, 1);
        int *ct = calloc(n + 64, sizeof(int));
        for (int i = 2; i < n; ++i) {
            if (!isw[i]) continue;
            int c = 0;
            for (int a = 1; a <= CAPW - 1 && i - a >= 2; ++a) c += isw[i - a] && word[i - a] == word[i];
            ct[i] = c;
        }
        int lmin = 1;
        while (5 * (lmin + 1) <= T) ++lmin;                 /* shortest l with 10 l (l + 1) / 2 > T l */
        const int K1n = lmin < 16 ? lmin : 16, LS = 15;     /* L1: partial sums over K1n words; L2: suffixes with l <= LS, or ok16 */
        for (int i = 2; i < n; ++i) {
            if (!isw[i]) continue;
            ++st_pos;
            if (ct[i] < thr) continue;
            ++st_k1;
            /* L1: removing the last k words never raises the ratio: sum_{j<k} 10 ct(i-j) > T k for k = 1..K1n */
            int sum = 0, ok = 1, ok16 = 1;
            for (int k = 1; k <= 16; ++k) {
                sum += i - (k - 1) >= 2 ? ct[i - (k - 1)] : 0;
                if (!(10 * sum > T * k)) { if (k <= K1n) ok = 0; ok16 = 0; }
            }
            if (!ok) continue;
            ++st_k2a;
            /* L2: a candidate among the suffixes with l <= LS (exact), or the 16-term condition every longer one needs */
            int c[64] = {0}, r = 0, shortc = 0;
            const int amax = i - 2 < LS ? i - 2 : LS;
            for (int a = 0; a <= amax && a <= CAPW - 1; ++a) {
                const int tt = word[i - a];
                r += c[tt]++;
                if (a >= 1 && r * 10 > T * a) shortc = 1;
            }
            if (shortc || (ok16 && CAPW - 1 > LS)) { S[i] = 1; ++st_k2; }
        }
        free(ct);
        /* K3 per chunk */
        ivlv_t got = {0, 0, 0}, seqall = {0, 0, 0};
        sequential_by_time(seq, n, T, W, &seqall);
        for (int k = 0; k < nch; ++k) {
            const int cs = k * chunk, ce = (k + 1) * chunk < n ? (k + 1) * chunk : n, last = ce == n;
            if (slow[k]) {
                for (int64_t q = 0; q < seqall.n; ++q)
                    if (seqall.a[q].tm >= cs && (seqall.a[q].tm < ce || (last && seqall.a[q].tm == n))) push(&got, seqall.a[q].s, seqall.a[q].f);
                continue;
            }
            /* lane <-> age arrays */
            int w[64], r[64], sr[64], sl[64];                /* slot: r, l (l = 0: empty) */
            int have = 0, cur = -1;
            memset(sl, 0, sizeof sl); memset(sr, 0, sizeof sr);
            const int i0 = cs == 0 ? 2 : cs - W + 2;
            for (int i = i0 < 2 ? 2 : i0; i < ce; ++i) {
                if (!S[i]) continue;
                ++st_k3steps;
                const int amax = i - 2 < CAPW - 1 ? i - 2 : CAPW - 1;
                if (!have || i - cur != 1) {
                    /* jump: evict what leaves the window (increasing start), shift the slots, reload the window */
                    if (have) {
                        const int g = i - cur;
                        for (int a = CAPW - 1; a >= 0; --a) {
                            if (a + g < CAPW) break;
                            if (sl[a]) {
                                const int s = cur - 2 - a, tm = s + W;
                                if (tm >= cs) push(&got, s, s + sl[a] + 3);
                            }
                        }
                        for (int a = CAPW - 1; a >= 0; --a) {
                            if (a - g >= 0) { sl[a] = sl[a - g]; sr[a] = sr[a - g]; } else { sl[a] = 0; sr[a] = 0; }
                        }
                    }
                    int c[64] = {0}, rr = 0;
                    for (int a = 0; a <= amax; ++a) { w[a] = word[i - a]; rr += c[w[a]]++; r[a] = rr; }
                    have = 1;
                } else {
                    /* incremental step: evict lane CAPW-1, shift, add the new word */
                    if (sl[CAPW - 1]) {
                        const int s = cur - 2 - (CAPW - 1), tm = s + W;
                        if (tm >= cs) push(&got, s, s + sl[CAPW - 1] + 3);
                    }
                    const int t = word[i];
                    int cnt = 0;
                    int nr[64];
                    nr[0] = 0;
                    for (int a = 1; a <= amax; ++a) { cnt += w[a - 1] == t; nr[a] = r[a - 1] + cnt; }
                    for (int a = CAPW - 1; a >= 1; --a) { w[a] = w[a - 1]; sl[a] = sl[a - 1]; sr[a] = sr[a - 1]; }
                    w[0] = t; sl[0] = 0; sr[0] = 0;
                    for (int a = 0; a <= amax; ++a) r[a] = nr[a];
                }
                cur = i;
                /* pass */
                int max_r = 0, max_l = 0, anyc = 0;
                for (int a = 0; a <= amax; ++a) {
                    /* entries with start >= this one: slots at ages <= a (newer starts), incl. what this pass inserted */
                    if (sl[a] && (max_r == 0 || sr[a] * max_l > max_r * sl[a])) { max_r = sr[a]; max_l = sl[a]; }
                    if (a >= 1 && r[a] * 10 > T * a) {
                        anyc = 1;
                        if (max_r == 0 || r[a] * max_l >= max_r * a) { max_r = r[a]; max_l = a; sr[a] = r[a]; sl[a] = a; ++st_k3ins; }
                    }
                }
                st_k3cand += anyc;
            }
            /* the rest of the chunk: what leaves the window before its end; the contig's end flushes everything */
            if (have) {
                for (int a = CAPW - 1; a >= 0; --a) {
                    if (!sl[a]) continue;
                    const int s = cur - 2 - a, tm = s + W;
                    if (last ? 1 : tm < ce) { if ((last && tm >= n ? n : tm) >= cs) push(&got, s, s + sl[a] + 3); }
                }
            }
        }
        canon(&got);
        int ok = got.n == n_ref;
        for (int64_t i = 0; ok && i < got.n; ++i) ok = ((uint64_t)(uint32_t)got.a[i].s << 32 | (uint32_t)got.a[i].f) == ref[i];
        int nslow = 0;
        for (int k = 0; k < nch; ++k) nslow += slow[k];
        printf("seed %d: %s  intervals %d (got %lld), slow chunks %d / %d\n", sd, ok ? "ok" : "MISMATCH", n_ref, (long long)got.n, nslow, nch);
        if (!ok) {
            ++bad;
            for (int64_t i = 0; i < got.n && i < n_ref; ++i)
                if (((uint64_t)(uint32_t)got.a[i].s << 32 | (uint32_t)got.a[i].f) != ref[i]) {
                    printf("  first difference at #%lld: got %d-%d, reference %d-%d (chunk %d)\n", (long long)i, got.a[i].s, got.a[i].f, (int)(ref[i] >> 32), (int)(uint32_t)ref[i], got.a[i].s / chunk);
                    break;
                }
        }
        free(seq); free(word); free(isw); free(slow); free(bad_prefix); free(S); free(got.a); free(seqall.a); orc_free(ref);
    }
    printf("T %d W %d chunk %d: words %llu, K1 %.4f %%, L1 %.4f %%, L2 = S %.4f %%, K3 steps %llu with candidates %llu, insertions %llu\n",
           T, W, chunk, st_pos, 100.0 * st_k1 / st_pos, 100.0 * st_k2a / st_pos, 100.0 * st_k2 / st_pos, st_k3steps, st_k3cand, st_k3ins);
    (void)st_hot;
    return bad != 0;
}

/* tools/sim/sdust_dp_sim.c — CPU model of the END-parallel form of sdust's find_perfect (round 4: the "dp tiles" of sd_sift),
 * checked against the oracle (oracle/oracle.c: orc_sdust) on plain A/C/G/T sequence.
 *
 * Claim: with word(p) = the 3-mer that ends at base p, c(j, i) = #{k in (j, i] : word(k) == word(j)},
 *   r(l, i) = sum_{j = i-l..i} c(j, i)      (the pair score of the l + 1 words that end at i)
 *   rho*(l, i) = r / l if 10 r > T l, else 0
 *   B(l, i) = max(rho*(l, i), B(l-1, i), B(l-1, i-1)),  B(0, .) = 0     (the best candidate inside the interval, itself included)
 *   perfect(l, i)  <=>  10 r > T l  and  rho*(l, i) >= max(B(l-1, i), B(l-1, i-1))
 * and the reference's result is the union (touching intervals merged) of [i - l - 2, i + 1) over the perfect (l, i), l <= W - 3.
 * c(i-l, i) = c(i-l, i-1) + [word(i) == word(i-l)]: lane <-> i, one step per l, the neighbour's value of the step before.
 *
 * build: gcc -O2 -o /tmp/dp_sim tools/sim/sdust_dp_sim.c oracle/oracle.c -Ioracle -lm
 * usage: /tmp/dp_sim [n_bases] [seeds]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static uint64_t rng_s;
static uint64_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

static void make_seq(uint8_t *s, int n, int kind)
{
    for (int i = 0; i < n; ++i) s[i] = "ACGT"[rnd() & 3];
    int feats = kind == 0 ? n / 3000 : n / 300;
    for (int f = 0; f < feats; ++f) {
        int p = (int)(rnd() % (uint64_t)(n - 600)), L = 10 + (int)(rnd() % 500), u = 1 + (int)(rnd() % 7);
        char unit[8];
        for (int j = 0; j < u; ++j) unit[j] = "ACGT"[rnd() & 3];
        int div = (int)(rnd() % 4);                       /* 0: exact, else 1 in 8 / 16 / 32 substituted */
        for (int j = 0; j < L; ++j) {
            s[p + j] = (uint8_t)unit[j % u];
            if (div && rnd() % (uint64_t)(4 << div) == 0) s[p + j] = "ACGT"[rnd() & 3];
        }
    }
    if (kind == 2) for (int i = 0; i < n; ++i) if (rnd() % 7 == 0) s[i] |= 0x20;   /* lower case */
}

/* exact compare of r1 / l1 with r2 / l2 (l > 0) */
static int ratio_ge(int r1, int l1, int r2, int l2) { return (long long)r1 * l2 >= (long long)r2 * l1; }

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 200000;
    const int seeds = argc > 2 ? atoi(argv[2]) : 6;
    static const int TW[][2] = {{20, 64}, {10, 32}, {25, 64}, {5, 64}, {30, 16}, {20, 7}, {2, 3}, {12, 66}, {20, 40}, {1, 64}, {60, 64}, {0, 10}};
    uint8_t *s = malloc(n + 8);
    int *word = malloc(sizeof(int) * (n + 8));
    long long total_iv = 0, fails = 0;
    for (int tw = 0; tw < (int)(sizeof(TW) / sizeof(TW[0])); ++tw) {
        const int T = TW[tw][0], W = TW[tw][1], LMAX = W - 3;
        if (LMAX < 1) {                                   /* W = 3: one word, never a pair */
        }
        for (int seed = 1; seed <= seeds; ++seed) {
            rng_s = 0x9E3779B97F4A7C15ull * (uint64_t)(seed + 100 * tw);
            make_seq(s, n, seed % 3);
            /* word(p) = the 3-mer that ends at p (a word for p >= 2) */
            {
                unsigned t = 0;
                for (int p = 0; p < n; ++p) {
                    int b = s[p] & 0xDF;
                    int code = b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3;
                    t = (t << 2 | (unsigned)code) & 63u;
                    word[p] = (int)t;                     /* valid as a word for p >= 2 */
                }
            }
            /* the DP, iteration l outermost as on the device; arrays over i */
            int *c = calloc(n + 1, sizeof(int)), *r = calloc(n + 1, sizeof(int));
            int *Br = calloc(n + 1, sizeof(int)), *Bl = calloc(n + 1, sizeof(int));   /* B as a fraction Br / Bl (Br == 0: none) */
            int *pc = calloc(n + 1, sizeof(int)), *pBr = calloc(n + 1, sizeof(int)), *pBl = calloc(n + 1, sizeof(int));
            int *lbest = calloc(n + 1, sizeof(int));
            for (int l = 1; l <= LMAX; ++l) {
                memcpy(pc, c, sizeof(int) * n);
                memcpy(pBr, Br, sizeof(int) * n);
                memcpy(pBl, Bl, sizeof(int) * n);
                for (int i = 2; i < n; ++i) {
                    if (i - l < 2) { c[i] = 0; continue; }            /* the suffix would begin in front of the first word */
                    const int cin = (i - 1 - (l - 1) >= 2 && i - 1 >= 2) ? pc[i - 1] : 0;   /* c(i-l, i-1): lane i-1, step l-1 */
                    c[i] = cin + (word[i] == word[i - l]);
                    r[i] += c[i];
                    const int cand = 10 * r[i] > T * l;
                    /* m = max(B(l-1, i), B(l-1, i-1)) */
                    int mr = pBr[i], ml = pBl[i];
                    if (i - 1 >= 2 && pBr[i - 1] && (!mr || !ratio_ge(mr, ml, pBr[i - 1], pBl[i - 1]))) { mr = pBr[i - 1]; ml = pBl[i - 1]; }
                    const int perf = cand && (!mr || ratio_ge(r[i], l, mr, ml));
                    if (perf) lbest[i] = l;
                    if (cand && (!mr || ratio_ge(r[i], l, mr, ml))) { Br[i] = r[i]; Bl[i] = l; }
                    else { Br[i] = mr; Bl[i] = ml; }
                }
            }
            /* union of [i - lbest - 2, i + 1) */
            uint8_t *cov = calloc(n + 2, 1);
            for (int i = 2; i < n; ++i)
                if (lbest[i]) for (int p = i - lbest[i] - 2; p <= i; ++p) cov[p] = 1;
            int32_t no = 0;
            uint64_t *exp = orc_sdust(s, n, T, W, &no);
            int k = 0, bad = 0;
            for (int p = 0; p < n && !bad;) {
                if (!cov[p]) { ++p; continue; }
                int q = p;
                while (q < n && cov[q]) ++q;
                if (k >= no || (int)(exp[k] >> 32) != p || (int)(uint32_t)exp[k] != q) bad = 1;
                ++k;
                p = q;
            }
            if (k != no) bad = 1;
            total_iv += no;
            if (bad) { ++fails; fprintf(stderr, "MISMATCH T=%d W=%d seed=%d: %d oracle intervals, %d here\n", T, W, seed, no, k); }
            free(exp); free(cov); free(c); free(r); free(Br); free(Bl); free(pc); free(pBr); free(pBl); free(lbest);
        }
    }
    printf("%lld intervals over %d (T, W) x %d seeds of %d bases: %s\n", total_iv, (int)(sizeof(TW) / sizeof(TW[0])), seeds, n, fails ? "FAILED" : "all equal");
    return fails != 0;
}

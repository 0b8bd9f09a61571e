/* sdust_trigger_sim.c — development aid (CPU): checks, on random and low-complexity sequence, the two facts the
 * production kernel's find_perfect trigger rests on, and measures how often each trigger fires.
 *   gcc -O2 -o sdust_trigger_sim sdust_trigger_sim.c && ./sdust_trigger_sim [T] [W] [Mbases]
 *
 * Reference (src/sdust/sdust.c): shift_window keeps v = the longest suffix of the window in which no 3-mer occurs more
 * than m = floor(T/5) times (:79-85), L = |v|; find_perfect is called when 10 rw > T L (:149) and examines the suffixes of
 * L+1 .. n words (:107): a suffix of q words with score r is a candidate when 10 r > T (q - 1) (:112).
 *   FACT 1  a suffix inside v is never a candidate, and a candidate implies the gate 10 rw > T L: so "examine every suffix
 *           of the window, no gate" is the same function — v, L, rv, cv and rw need not be kept.
 *   FACT 2  M' = min(M + T - 10 ct, ct >= m ? m (T - 5m - 5) : m (T - 5m + 5)) is a lower bound of
 *           min over suffixes of >= m+1 words of (T (q-1) - 10 r) after pushing a word that had ct copies in the window,
 *           if M was one before (and the exact minimum may replace M at any time): no candidate exists while M >= 0.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

static uint64_t rng_s = 88172645463325252ULL;
static inline uint64_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 20, W = argc > 2 ? atoi(argv[2]) : 64;
    const long n = (argc > 3 ? atol(argv[3]) : 20) * 1000000L;
    const int CAP = W - 2, m = (T << 1) / 10;
    uint8_t *seq = malloc(n);
    for (long i = 0; i < n; ++i) seq[i] = rnd() & 3;
    /* low-complexity features: STRs, homopolymers, diverged satellites, N runs */
    const long spacing = argc > 4 ? atol(argv[4]) : 7919;      /* bases between planted features; 0: none (uniform random) */
    for (long p = 5000; spacing > 0 && p + 3000 < n; p += spacing) {
        int kind = (p / spacing) % 6, len = 20 + rnd() % 300, ul = 1 + rnd() % 6;
        uint8_t u[8];
        for (int k = 0; k < ul; ++k) u[k] = rnd() & 3;
        if (kind == 5) { for (int k = 0; k < len / 8 + 1; ++k) seq[p + k] = 4; continue; }
        if (kind == 4) len *= 8;
        for (int k = 0; k < len; ++k) seq[p + k] = (kind >= 3 && rnd() % 50 == 0) ? (rnd() & 3) : u[k % ul];
    }
    int ring[512], front = 0, size = 0, cw[64] = {0}, cv[64] = {0}, rw = 0, rv = 0, L = 0, l = 0;
    unsigned t = 0;
    long words = 0, ref_calls = 0, ref_effect = 0, any_cand = 0, trig = 0, viol1 = 0, viol2 = 0, old_skip_calls = 0;
    long M = 0;              /* tracker of FACT 2 */
    int have_M = 0;
    long M7 = 0, have_M7 = 0, t7 = 0, viol7 = 0, B1 = 1L << 40;
    for (int o = 1; o <= W; ++o) {          /* event with ct == m, not all equal: m+1 copies of t and o other words with counts <= m */
        long g = (long)(o / m) * (m * (m - 1) / 2) + (long)(o % m) * (o % m - 1) / 2;
        long v = (long)T * (m + o) - 5L * m * (m + 1) - 10 * g;
        if (v < B1) B1 = v;
    }
    const long B0 = (long)m * (T - 5 * m - 5);
    long M8 = 0, have_M8 = 0, t8 = 0, viol8 = 0, a8 = 0, t8_ev_gt = 0, t8_ev_eq = 0, t8_decay = 0;
    long M3 = 0, have_M3 = 0, t6 = 0, viol4 = 0; int tprev = -1;
    long since_event = 0, words_since_start = 0, t2 = 0, t3 = 0, t_gate_and_M = 0, viol3 = 0, M2 = 0, have_M2 = 0, t5 = 0, trims = 0;
    for (long i = 0; i < n; ++i) {
        int b = seq[i];
        if (b >= 4) { l = 0; t = 0; continue; }
        ++l;
        t = (t << 2 | b) & 63;
        if (l < 3) continue;
        ++words;
        /* shift_window */
        if (size >= CAP) {
            int s = ring[front];
            front = (front + 1) % 512; --size;
            rw -= --cw[s];
            if (L > size) { --L; rv -= --cv[s]; }
        }
        ring[(front + size) % 512] = t; ++size; ++L;
        const int ct = cw[t];
        rw += cw[t]++;
        rv += cv[t]++;
        if (ct >= m) { since_event = 0; ++trims; } else ++since_event;
        ++words_since_start;
        if (cv[t] * 10 > T << 1) {
            int s;
            do { s = ring[(front + size - L) % 512]; rv -= --cv[s]; --L; } while (s != (int)t);
        }
        /* tracker */
        {
            const long bnew = ct >= m ? (long)m * (T - 5 * m - 5) : (long)m * (T - 5 * m + 5);
            const long adv = M + T - 10L * ct;
            M = have_M ? (adv < bnew ? adv : bnew) : -1;     /* no bound yet: trigger */
        }
        /* exact: every suffix */
        int c[64] = {0}, r = 0, cand_all = 0, cand_long = 0, hot = 0;
        long minmargin = 1L << 40, minhot = 1L << 30;
        for (int q = 1; q <= size; ++q) {
            int w = ring[(front + size - q) % 512];
            r += c[w]++;
            long margin = (long)T * (q - 1) - 10L * r;
            if (margin < 0) { ++cand_all; if (q > L) ++cand_long; else ++viol1; }
            if (q >= m + 1 && margin < minmargin) minmargin = margin;
            if (c[w] >= m + 1) hot = 1;                                /* this suffix holds m+1 copies of some word */
            if (hot && margin < minhot) minhot = margin;
            if (margin < 0 && !hot) ++viol1;                            /* a candidate is always hot */
        }
        const int gate = rw * 10 > L * T;
        {
            long Llb = ct >= m ? m : m + since_event;
            if (words_since_start < Llb || since_event == words_since_start) Llb = words_since_start < size ? words_since_start : size;   /* no event yet: exact */
            if (Llb > size) Llb = size;
            if (Llb > L) ++viol3;
            const int gate2 = rw * 10 > Llb * T;
            if (gate2) ++t2;
            if (gate2 && M < 0) ++t3;
            if (gate && M < 0) ++t_gate_and_M;
            if (cand_all && !gate2) ++viol3;
            /* T5: the tracker refreshed only by passes that run under (gate2 && M2 < 0) */
            const long bnew = ct >= m ? (long)m * (T - 5 * m - 5) : (long)m * (T - 5 * m + 5);
            const long adv = M2 + T - 10L * ct;
            M2 = have_M2 ? (adv < bnew ? adv : bnew) : -1;
            if (gate2 && M2 < 0) { ++t5; M2 = size >= m + 1 ? minmargin : 0; have_M2 = 1; }
            if (cand_all && !(gate2)) ++viol3;
            /* T6: fresh-suffix bound that looks at the previous word: the suffix of m+1 words can only be a candidate if all equal */
            const long bnew6 = (ct >= m && (int)t == tprev) ? (long)m * (T - 5 * m - 5) : (long)m * (T - 5 * m + 5);
            const long adv6 = M3 + T - 10L * ct;
            M3 = have_M3 ? (adv6 < bnew6 ? adv6 : bnew6) : -1;
            if (have_M3 && M3 >= 0 && cand_all) ++viol4;
            if (have_M3 && M3 > minmargin && size >= m + 1) ++viol4;
            if (M3 < 0) { ++t6; M3 = size >= m + 1 ? minmargin : 0; have_M3 = 1; }
            /* T7: only suffixes that reach back to 4 positions before the most recent event can hold m+1 copies of a word */
            if (have_M7) {
                M7 += T - 10L * ct;
                if (ct == m) { const long bn = (int)t == tprev ? B0 : B1; if (bn < M7) M7 = bn; }
                else if (ct > m) M7 = -1L << 30;
            } else M7 = -1;
            if (have_M7 && M7 >= 0 && cand_all) ++viol7;
            if (M7 < 0) { ++t7; M7 = size >= m + 1 ? minmargin : (1L << 30); have_M7 = 1; }
            /* T8: M bounds the margins of the HOT suffixes only (those holding m+1 copies of some word); a8 = pushes since the last event */
            {
                int cause = 0;
                if (have_M8) {
                    const long adv = M8 + T - 10L * ct;
                    if (ct < m) { M8 = adv; cause = 3; }
                    else if (ct == m) { const long bn = (int)t == tprev ? B0 : B1; M8 = adv < bn ? adv : bn; cause = 2; }
                    else if (getenv("T8_GAP")) { const long k1 = ct + 1; const long bn = 5L * (a8 + m) - T - 5L * k1 * (k1 - m); M8 = adv < bn ? adv : bn; cause = 1; }
                    else if (getenv("T8_OLD")) { M8 = -1; cause = 1; }       /* first form of the kernel: a push with ct > m always runs the pass */
                    else { const long bn = (int)t == tprev ? B0 : B1; M8 = adv < bn ? adv : bn; cause = 1; }   /* the newly hot suffixes hold exactly m+1 copies of t here as well */
                } else M8 = -1;
                if (ct >= m) a8 = 0; else ++a8;
                if (have_M8 && M8 >= 0 && cand_all) ++viol8;
                if (M8 < 0) { ++t8; if (cause == 1) ++t8_ev_gt; else if (cause == 2) ++t8_ev_eq; else ++t8_decay; M8 = getenv("T8_HOT") ? minhot : (size >= m + 1 ? minmargin : (1L << 30)); have_M8 = 1; }
            }
            tprev = (int)t;
        }
        if (gate) ++ref_calls;
        if (gate && cand_long) ++ref_effect;
        if (cand_all) ++any_cand;
        if (cand_all && !gate) ++viol1;                     /* FACT 1: a candidate implies the gate */
        if (have_M && M >= 0 && cand_all) ++viol2;          /* FACT 2: the tracker never hides a candidate */
        if (have_M && M > minmargin && size >= m + 1) ++viol2;   /* ... because it is a lower bound of the exact minimum */
        if (M < 0) {                                        /* the pass runs: it leaves the exact minimum */
            ++trig;
            M = size >= m + 1 ? minmargin : 0;
            have_M = 1;
        }
    }
    printf("T=%d W=%d m=%d: %ld words; reference calls %ld (%.3f%%), of which with candidates %ld (%.3f%%); steps with a candidate suffix %ld;\n"
           "tracker passes %ld (%.3f%% of the words; %.2f per 64-lane wave-step); FACT 1 violations %ld, FACT 2 violations %ld\n"
           "events (ct >= m) %.3f%%; gate with the lower bound of L %.3f%%; that gate & tracker %.3f%%; exact gate & tracker %.3f%%; gate' & tracker refreshed only then %.3f%% ; violations %ld\ntracker with the previous-word test: %.3f%% (%.2f per wave-step), violations %ld\nevent tracker: %.3f%% (%.3f per wave-step), violations %ld (B0 %ld B1 %ld)\nhot-suffix tracker T8: %.3f%% (%.3f per wave-step; at ct>m events %.3f%%, at ct==m events %.3f%%, at other pushes %.3f%%), violations %ld\n",
           T, W, m, words, ref_calls, 100.0 * ref_calls / words, ref_effect, 100.0 * ref_effect / words, any_cand, trig, 100.0 * trig / words,
           64.0 * trig / words, viol1, viol2, 100.0 * trims / words, 100.0 * t2 / words, 100.0 * t3 / words, 100.0 * t_gate_and_M / words, 100.0 * t5 / words, viol3, 100.0 * t6 / words, 64.0 * t6 / words, viol4, 100.0 * t7 / words, 64.0 * t7 / words, viol7, B0, B1, 100.0 * t8 / words, 64.0 * t8 / words, 100.0 * t8_ev_gt / words, 100.0 * t8_ev_eq / words, 100.0 * t8_decay / words, viol8);
    (void)old_skip_calls;
    return viol1 || viol2;
}

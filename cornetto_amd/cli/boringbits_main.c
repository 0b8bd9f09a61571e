/* boringbits_main.c — `cornetto noboringbits|boringbits cov-total.bg -q cov-mq20.bg [options]`.
 * Reference: src/boringbits_main.c:558-660 (options), :180-301 (get_depths: lock-step parse + validation),
 * :483-536 (the_boring_bits), :425-445 / :463-481 (printing).
 * Host: reads the two files in large pieces into pinned buffers and streams them to the device.
 * Device: tokenising + parsing + the reference's lock-step checks (cornetto_bgin_*), block sums, totals for the
 * mean, window means, classification, ordered selection.
 *
 * Extension (not in the reference, off by default): `noboringbits ... --panel assembly.bed [--lowq lowQ.bed]` prints
 * what steps 1-9 of scripts/create-cornetto.sh:41-66 produce (the boring bits before `bigenough`) instead of the
 * window lines: the windows are merged on the device (bedtools merge -d 1000, >= 30 kb) and the rest of the bedtools /
 * awk glue is cornetto_panel_boring().  assembly.bed is the output of `cornetto fa2bed`. */
#include <getopt.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <errno.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>

#include "cli.h"

typedef struct {
    int window_size, window_inc;
    float low_cov_thresh, high_cov_thresh, low_mq_cov_thresh;
    int min_ctg_len, edge_len;
} optp_t;

static void print_help(FILE *fp, const optp_t *o)
{
    fprintf(fp, "Usage: cornetto boringbits cov-total.bg -q cov-mq20.bg\n");
    fprintf(fp, "\nbasic options:\n");
    fprintf(fp, "   -q FILE                    depth file with high mapq read coverage\n");
    fprintf(fp, "   -w INT                     window size [%d]\n", o->window_size);
    fprintf(fp, "   -i INT                     window increment [%d]\n", o->window_inc);
    fprintf(fp, "   -L FLOAT                   low coverage threshold factor [%.1f]\n", o->low_cov_thresh);
    fprintf(fp, "   -H FLOAT                   high coverage threshold factor [%.1f]\n", o->high_cov_thresh);
    fprintf(fp, "   -Q FLOAT                   mapq low coverage threshold factor [%.1f]\n", o->low_mq_cov_thresh);
    fprintf(fp, "   -m INT                     minimum contig length [%d]\n", o->min_ctg_len);
    fprintf(fp, "   -e INT                     edge length to ignore [%d]\n", o->edge_len);
    fprintf(fp, "   -h                         help\n");
    fprintf(fp, "   --verbose INT              verbosity level [%d]\n", cli_log_level);
    fprintf(fp, "   --accel=yes|no             Running on accelerator [yes]\n");
}

/* ---- --panel: steps 4-9 of scripts/create-cornetto.sh on the merged fun windows ---------------------------------- */
typedef struct {
    char **name;
    int32_t *len, n, cap;
} pn_asm_t;
static int pn_par[9] = {1000, 30000, 8000, 40000, 200000, 200000, 800000, -1, -1};   /* create-cornetto.sh:44,47,50,53,56,59,65; extend-right, extend-gate (-1: as extend) */

static int32_t pn_find(const pn_asm_t *a, const int32_t *slots, uint32_t n_slots, const char *name)
{
    uint32_t h = 2166136261u;
    for (const char *s = name; *s; ++s) h = (h ^ (unsigned char)*s) * 16777619u;
    for (uint32_t b = h & (n_slots - 1); slots[b] >= 0; b = (b + 1) & (n_slots - 1))
        if (strcmp(a->name[slots[b]], name) == 0) return slots[b];
    return -1;
}

static void panel_print(const char *asm_bed, const char *lowq_bed, char **cov_names, int32_t n_cov, const cornetto_ivl_t *fun, int64_t n_fun)
{
    char *line = NULL;
    size_t cap = 0;
    /* the assembly BED of `cornetto fa2bed`: name 0 length, in FASTA order (create-cornetto.sh:32-33) */
    pn_asm_t a = {NULL, NULL, 0, 0};
    FILE *f = fopen(asm_bed, "r");
    if (!f) {
        CLI_ERROR("Failed to open %s : No such file or directory.", asm_bed);
        exit(EXIT_FAILURE);
    }
    while (getline(&line, &cap, f) > 0) {
        char nm[2048];
        long long b, e;
        if (sscanf(line, "%2047s %lld %lld", nm, &b, &e) != 3) continue;
        if (a.n == a.cap) {
            a.cap = a.cap ? a.cap * 2 : 256;
            a.name = (char **)cli_xrealloc(a.name, (size_t)a.cap * sizeof(char *));
            a.len = (int32_t *)cli_xrealloc(a.len, (size_t)a.cap * sizeof(int32_t));
        }
        a.name[a.n] = cli_xstrdup(nm);
        a.len[a.n++] = (int32_t)(e - b);
    }
    fclose(f);
    uint32_t n_slots = 16;
    while (n_slots < (uint32_t)a.n * 2u + 2u) n_slots <<= 1;
    int32_t *slots = (int32_t *)cli_xmalloc(n_slots * sizeof(int32_t));
    for (uint32_t i = 0; i < n_slots; ++i) slots[i] = -1;
    for (int32_t i = 0; i < a.n; ++i) {
        uint32_t h = 2166136261u;
        for (const char *s = a.name[i]; *s; ++s) h = (h ^ (unsigned char)*s) * 16777619u;
        uint32_t b = h & (n_slots - 1);
        while (slots[b] >= 0) b = (b + 1) & (n_slots - 1);
        slots[b] = i;
    }
    /* fun windows: bedgraph contig index -> assembly index (rows of contigs the assembly BED does not have vanish in
     * `bedtools subtract -a assembly`, :62) */
    cornetto_ivl_t *fv = (cornetto_ivl_t *)cli_xmalloc(((size_t)n_fun + 1) * sizeof(*fv));
    int32_t *cmap = (int32_t *)cli_xmalloc(((size_t)n_cov + 1) * sizeof(int32_t));
    for (int32_t i = 0; i < n_cov; ++i) cmap[i] = pn_find(&a, slots, n_slots, cov_names[i]);
    int64_t nf = 0;
    for (int64_t i = 0; i < n_fun; ++i)
        if (cmap[fun[i].ctg] >= 0) {
            fv[nf] = fun[i];
            fv[nf++].ctg = cmap[fun[i].ctg];
        }
    /* hifiasm low-quality regions (:50): name start end ... */
    cornetto_ivl_t *lq = NULL;
    int64_t nl = 0, capl = 0;
    if (lowq_bed) {
        f = fopen(lowq_bed, "r");
        if (!f) {
            CLI_ERROR("Failed to open %s : No such file or directory.", lowq_bed);
            exit(EXIT_FAILURE);
        }
        while (getline(&line, &cap, f) > 0) {
            char nm[2048];
            long long b, e;
            if (sscanf(line, "%2047s %lld %lld", nm, &b, &e) != 3) continue;
            const int32_t id = pn_find(&a, slots, n_slots, nm);
            if (id < 0) continue;
            if (nl == capl) {
                capl = capl ? capl * 2 : 1024;
                lq = (cornetto_ivl_t *)cli_xrealloc(lq, (size_t)capl * sizeof(*lq));
            }
            lq[nl].ctg = id; lq[nl].start = (int32_t)b; lq[nl].finish = (int32_t)e;
            ++nl;
        }
        fclose(f);
    }
    cornetto_panel_opt_t po;
    cornetto_panel_defaults(&po);
    po.min_lowq_len = pn_par[2]; po.extend = pn_par[3]; po.edge_len = pn_par[4]; po.merge_dist = pn_par[5]; po.min_ctg_len = pn_par[6];
    po.extend_right = pn_par[7] >= 0 ? pn_par[7] : pn_par[3]; /* the two constants recreate-cornetto.sh:38 adds; default: symmetric */
    po.extend_gate = pn_par[8] >= 0 ? pn_par[8] : pn_par[3];
    cornetto_ivl_t *out = NULL;
    int64_t n_out = 0;
    if (cornetto_panel_boring(a.len, a.n, fv, nf, lq, nl, &po, &out, &n_out) != CORNETTO_OK) {
        CLI_ERROR("%s", "panel interval stage failed");
        exit(EXIT_FAILURE);
    }
    for (int64_t i = 0; i < n_out; ++i) printf("%s\t%d\t%d\n", a.name[out[i].ctg], out[i].start, out[i].finish);
    cornetto_free(out);
    free(line);
}

/* ---------------- one round of the threaded reader: `want` bytes of each of the two files at their offsets ---------------- */
typedef struct {
    int fd[2], n_threads, started;
    char *dst[2];
    int64_t off[2], want[2], got[2];
    pthread_t th;
} bg_round_t;

static void *bg_round_thread(void *p)
{
    bg_round_t *r = (bg_round_t *)p;
    const double t0 = cli_realtime();
    for (int f = 0; f < 2; ++f) r->got[f] = r->want[f] > 0 ? cli_pread_parallel(r->fd[f], r->dst[f], r->want[f], r->off[f], r->n_threads) : 0;
    if (getenv("CORNETTO_CLI_TRACE"))
        fprintf(stderr, "[cli trace] round read: %lld + %lld bytes in %.2f ms\n", (long long)r->got[0], (long long)r->got[1], (cli_realtime() - t0) * 1e3);
    return NULL;
}

static void bg_round_post(bg_round_t *r)
{
    r->started = pthread_create(&r->th, NULL, bg_round_thread, r) == 0;
    if (!r->started) bg_round_thread(r); /* no thread to be had: read here */
}

static void bg_round_join(bg_round_t *r)
{
    if (r->started) pthread_join(r->th, NULL);
    r->started = 0;
}

/* ---------------- get_depths over a byte range of each of two regular files (src/boringbits_main.c:180-301) ----------------
 * pread() threads fill one pair of pinned pieces while the device takes the other.  The whole files on one device, or one share of
 * them per device (bg_split): nothing here prints or exits — what went wrong comes back in the job. */
typedef struct {
    /* in */
    cornetto_accel_t *h;
    int fd_t, fd_q, n_rd;
    int not_last;                  /* a share in front of another one: cov-mq bytes left over are the next share's partners (see bg_ingest) */
    int64_t t0, t1, q0, q1, piece; /* [t0, t1) of the total-depth file, [q0, q1) of the mapq file */
    /* out */
    cornetto_cov_t *cov;
    int32_t n_ctg;
    char **names;
    int64_t n_clamped;
    int fmt_kind, fmt_a, fmt_b; /* a check of the reference failed (cornetto_bgerr_t) */
    int rc;                     /* anything else: CORNETTO_E_*; -1000 - f: reading file f failed */
    char err[600];
} bg_job_t;

static void bg_job_fail(bg_job_t *g, int rc, const char *what)
{
    g->rc = rc;
    snprintf(g->err, sizeof(g->err), "%s failed: %s (%s)", what, cornetto_accel_last_error(g->h), cornetto_accel_strerror(rc));
}

static void bg_ingest(bg_job_t *g)
{
    /* Three stages side by side, a pair of pinned pieces each (round 6): pread() threads fill the pieces of round k + 1, the pieces of round k are on
     * their way to the device on a copy queue of their own (cornetto_bgin_prefetch), the device parses round k - 1 (cornetto_bgin_feed: tokenise,
     * convert, the five checks).  Until round 5 the upload of a round stood in front of its parse: 4.6 + 3 ms per 2 x 64 MB. */
    const int64_t piece = g->piece;
    char *buf[3][2] = {{NULL, NULL}, {NULL, NULL}, {NULL, NULL}};
    cornetto_bgin_t *bg = NULL;
    bg_round_t rd[3];
    int fin[3] = {0, 0, 0};            /* the eof bits that go with a round's pieces */
    memset(rd, 0, sizeof(rd));
    g->rc = CORNETTO_OK;
    g->fmt_kind = 0;
    for (int i = 0; i < 3; ++i)
        for (int f = 0; f < 2; ++f)
            if (!(buf[i][f] = (char *)cornetto_pinned_alloc((size_t)piece))) {
                g->rc = CORNETTO_E_NOMEM;
                snprintf(g->err, sizeof(g->err), "cannot allocate pinned read buffers");
                goto out;
            }
    for (int i = 0; i < 3; ++i) {
        rd[i].fd[0] = g->fd_t;
        rd[i].fd[1] = g->fd_q;
        rd[i].dst[0] = buf[i][0];
        rd[i].dst[1] = buf[i][1];
        rd[i].n_threads = g->n_rd;
    }
    int64_t off[2] = {g->t0, g->q0};
    const int64_t end[2] = {g->t1, g->q1};
    for (int f = 0; f < 2; ++f) {
        rd[0].off[f] = off[f];
        rd[0].want[f] = end[f] - off[f] < piece ? end[f] - off[f] : piece;
    }
    bg_round_post(&rd[0]); /* (the first round is on its way while the parser's state is set up) */
    int rc = cornetto_bgin_open(g->h, &bg);
    if (rc != CORNETTO_OK) {
        bg_job_fail(g, rc, "bedgraph ingest");
        goto out;
    }
    int eof[2] = {0, 0};
    int64_t n_posted = 1, n_joined = 0, n_fed = 0;
    while (!cornetto_bgin_done(bg)) {
        if (n_joined < n_posted) {
            /* round n_joined has been read: its pieces start for the device, the next round starts to be read */
            bg_round_t *r = &rd[n_joined % 3];
            bg_round_join(r);
            for (int f = 0; f < 2; ++f) {
                if (r->got[f] < 0) {
                    g->rc = -1000 - f;
                    goto out;
                }
                off[f] += r->got[f];
                if (r->got[f] < r->want[f] || off[f] >= end[f]) eof[f] = 1;
            }
            fin[n_joined % 3] = eof[0] | (eof[1] << 1);
            ++n_joined;
            rc = cornetto_bgin_prefetch(g->h, bg, r->dst[0], r->got[0], r->dst[1], r->got[1]);
            if (rc != CORNETTO_OK) {
                bg_job_fail(g, rc, "bedgraph ingest");
                goto out;
            }
            if (!(eof[0] && eof[1])) {
                /* the next round is sized before the ones in front of it are parsed: the bytes of either file that will be pending after them, if both
                 * files spend the same number of bytes per line (they nearly do), differ by `ahead`; the file that is ahead reads less */
                int64_t pend[2] = {0, 0};
                cornetto_bgin_pending(bg, &pend[0], &pend[1]);
                int64_t unfed[2] = {0, 0};
                for (int64_t j = n_fed; j < n_joined; ++j)
                    for (int f = 0; f < 2; ++f) unfed[f] += rd[j % 3].got[f];
                const int64_t ahead = (pend[0] + unfed[0]) - (pend[1] + unfed[1]);
                const int64_t least = piece < 4096 ? piece : 4096;
                int64_t w[2] = {piece - (ahead > 0 ? ahead : 0), piece - (ahead < 0 ? -ahead : 0)};
                bg_round_t *nx = &rd[n_posted % 3];
                for (int f = 0; f < 2; ++f) {
                    if (w[f] < least) w[f] = least;
                    if (w[f] > end[f] - off[f]) w[f] = end[f] - off[f];
                    nx->want[f] = eof[f] ? 0 : w[f];
                    nx->off[f] = off[f];
                }
                bg_round_post(nx);
                ++n_posted;
            }
            if (n_joined - n_fed < 2 && n_joined < n_posted) continue;   /* (keep one round between the reader and the parser: its upload runs beside the parse) */
        }
        if (n_fed >= n_joined) { /* cannot happen: the final feed either finishes or fails */
            g->rc = CORNETTO_E_ARG;
            snprintf(g->err, sizeof(g->err), "bedgraph ingest did not finish");
            goto out;
        }
        bg_round_t *r = &rd[n_fed % 3];
        rc = cornetto_bgin_feed(g->h, bg, r->dst[0], r->got[0], r->dst[1], r->got[1], fin[n_fed % 3]);
        ++n_fed;
        if (rc == CORNETTO_E_FORMAT) {
            const cornetto_bgerr_t *e = cornetto_bgin_error(bg);
            g->fmt_kind = e->kind ? e->kind : 5;
            g->fmt_a = e->a;
            g->fmt_b = e->b;
            goto out;
        }
        if (rc != CORNETTO_OK) {
            bg_job_fail(g, rc, "bedgraph ingest");
            goto out;
        }
    }
    if (g->not_last && cornetto_bgin_unmatched_mq(bg) > 0) {
        /* the sequential loop pairs the cov-total record after this share with those tokens (:214-227): a cut is only the same line in
         * both files when every record in front of it has its partner */
        g->fmt_kind = 3;
        goto out;
    }
    rc = cornetto_bgin_finish(g->h, bg, &g->cov, &g->n_ctg, &g->names, &g->n_clamped);
    if (rc != CORNETTO_OK) bg_job_fail(g, rc, "bedgraph ingest");
out:
    for (int i = 0; i < 3; ++i)
        if (rd[i].started) bg_round_join(&rd[i]); /* never leave a reader behind */
    if (bg) cornetto_bgin_close(g->h, bg);
    for (int i = 0; i < 3; ++i)
        for (int f = 0; f < 2; ++f) cornetto_pinned_free(buf[i][f]);
}

/* the reference's messages and exit code for what bg_ingest() brought back (:209-227, :249-259) */
static void bg_job_check(const bg_job_t *g, const char *name_t, const char *name_q)
{
    if (g->fmt_kind) {
        if (g->fmt_kind == 1 || g->fmt_kind == 2) {
            CLI_ERROR("The depth files should have 4 columns. Had %d.", g->fmt_a);
        } else if (g->fmt_kind == 3) {
            CLI_ERROR("%s", "The two files are not in the same order");
        } else if (g->fmt_kind == 4) {
            CLI_ERROR("The depth files should be incremantal at one base resolution. Found %d to %d", g->fmt_a, g->fmt_b);
        } else {
            CLI_ERROR("The depth files should have end=start+1. Found %d to %d", g->fmt_a, g->fmt_b);
        }
        exit(EXIT_FAILURE);
    }
    if (g->rc <= -1000) {
        CLI_ERROR("reading %s failed", g->rc == -1000 ? name_t : name_q);
        exit(EXIT_FAILURE);
    }
    if (g->rc != CORNETTO_OK) {
        CLI_ERROR("%s", g->err);
        exit(EXIT_FAILURE);
    }
}

/* ---------------- where to cut two per-base bedgraphs into shares of whole contigs ----------------
 * Both files hold the same lines (contig, position) in the same order, at different byte offsets.  A cut is a line start where the
 * contig name changes; it is the same cut in both files when the line in front of it (name, position) and the line behind it are the
 * same in both.  Found with a few 8 KB pread() probes per cut: the lines of a contig are one block, so "still the contig of the probe"
 * is a predicate a binary search can use. */
#define BG_NAME_MAX 1024
typedef struct {
    int64_t at;          /* line start behind the cut (the file's size: no line) */
    char prev[BG_NAME_MAX], next[BG_NAME_MAX];
    int64_t prev_pos, next_pos;
} bg_cut_t;

/* the first line that starts at or behind `off`: 1 and its start / name / first number; 0: there is none; -1: not a line this code reads */
static int bg_line_at(int fd, int64_t size, int64_t off, int64_t *ls, char *name, int64_t *pos)
{
    char buf[8192];
    if (off >= size) return 0;
    const int64_t from = off > 0 ? off - 1 : 0;
    ssize_t n;
    do n = pread(fd, buf, sizeof(buf), (off_t)from);
    while (n < 0 && errno == EINTR);
    if (n <= 0) return -1;
    ssize_t i = 0;
    if (off > 0) {
        while (i < n && buf[i] != '\n') ++i;
        if (i == n) return from + n >= size ? 0 : -1;
        ++i;
    }
    if (from + i >= size) return 0;
    ssize_t j = i;
    while (j < n && buf[j] != '\t' && buf[j] != ' ' && buf[j] != '\n') ++j;
    if (j == n || buf[j] == '\n' || j == i || j - i >= BG_NAME_MAX) return -1;
    ssize_t k = j + 1;
    int64_t v = 0;
    if (k >= n || buf[k] < '0' || buf[k] > '9') return -1;
    for (; k < n && buf[k] >= '0' && buf[k] <= '9'; ++k) v = v * 10 + (buf[k] - '0');
    if (k == n) return -1;
    memcpy(name, buf + i, (size_t)(j - i));
    name[j - i] = 0;
    *ls = from + i;
    *pos = v;
    return 1;
}

/* the line that ends right in front of the line start `at` (> 0) */
static int bg_line_before(int fd, int64_t size, int64_t at, char *name, int64_t *pos)
{
    char buf[8192];
    const int64_t from = at > (int64_t)sizeof(buf) ? at - (int64_t)sizeof(buf) : 0;
    ssize_t n;
    do n = pread(fd, buf, (size_t)(at - from), (off_t)from);
    while (n < 0 && errno == EINTR);
    if (n != at - from || n < 2 || buf[n - 1] != '\n') return -1;
    ssize_t i = n - 2;
    while (i >= 0 && buf[i] != '\n') --i;
    if (i < 0 && from > 0) return -1; /* a line of more than 8 KB */
    int64_t ls;
    return bg_line_at(fd, size, from + i + 1, &ls, name, pos) == 1 && ls == from + i + 1 ? 1 : -1;
}

/* lo: a line start of contig X.  -> the start of the first line behind it that belongs to another contig (size: none); -1: give up */
static int64_t bg_block_end(int fd, int64_t size, int64_t lo, const char *X)
{
    char nm[BG_NAME_MAX];
    int64_t hi = size, ls, p;
    for (int64_t step = 1 << 20; lo + step < size; step *= 2) { /* gallop */
        const int r = bg_line_at(fd, size, lo + step, &ls, nm, &p);
        if (r < 0) return -1;
        if (r == 0) break;
        if (strcmp(nm, X)) {
            hi = ls;
            break;
        }
        lo = ls;
    }
    int64_t cap = hi; /* no line starts in [cap, hi) */
    while (cap - lo > 4096) {
        const int64_t mid = lo + (cap - lo) / 2;
        const int r = bg_line_at(fd, size, mid, &ls, nm, &p);
        if (r < 0) return -1;
        if (r == 0 || ls >= hi) cap = mid;
        else if (strcmp(nm, X)) hi = cap = ls;
        else lo = ls;
    }
    for (;;) { /* line by line */
        const int r = bg_line_at(fd, size, lo + 1, &ls, nm, &p);
        if (r < 0) return -1;
        if (r == 0 || ls >= hi) return hi;
        if (strcmp(nm, X)) return ls;
        lo = ls;
    }
}

static int bg_cut_describe(int fd, int64_t size, int64_t at, bg_cut_t *c)
{
    int64_t ls;
    c->at = at;
    c->next[0] = 0;
    c->next_pos = -1;
    if (at < size && !(bg_line_at(fd, size, at, &ls, c->next, &c->next_pos) == 1 && ls == at)) return -1;
    return at > 0 ? bg_line_before(fd, size, at, c->prev, &c->prev_pos) : -1;
}

/* up to n_want shares -> their number (1: no cut found that is provably the same in both files); cut[k] = start of share k + 1 */
static int bg_split(int fd_t, int64_t size_t_, int fd_q, int64_t size_q, int n_want, int64_t *cut_t, int64_t *cut_q)
{
    int n = 0;
    char nm[BG_NAME_MAX];
    for (int d = 1; d < n_want; ++d) {
        int64_t ls, p;
        bg_cut_t ct, cq;
        if (bg_line_at(fd_t, size_t_, size_t_ / n_want * d, &ls, nm, &p) != 1) continue;
        const int64_t at_t = bg_block_end(fd_t, size_t_, ls, nm);
        if (at_t <= 0 || at_t >= size_t_ || (n && at_t <= cut_t[n - 1])) continue;
        if (bg_cut_describe(fd_t, size_t_, at_t, &ct) != 1) continue;
        /* the same cut in the other file: near the same fraction of its bytes */
        const int64_t est = (int64_t)((double)at_t / (double)size_t_ * (double)size_q);
        if (bg_line_at(fd_q, size_q, est < size_q ? est : size_q - 1, &ls, nm, &p) != 1) continue;
        int64_t at_q = -1;
        if (!strcmp(nm, ct.prev)) {
            at_q = bg_block_end(fd_q, size_q, ls, ct.prev);
        } else if (!strcmp(nm, ct.next)) { /* behind the cut: back to a line of the contig in front of it */
            for (int64_t step = 1 << 20, at = ls;; step *= 2) {
                at = at > step ? at - step : 0;
                const int r = bg_line_at(fd_q, size_q, at, &ls, nm, &p);
                if (r != 1) break;
                if (!strcmp(nm, ct.prev)) {
                    at_q = bg_block_end(fd_q, size_q, ls, ct.prev);
                    break;
                }
                if (strcmp(nm, ct.next) || at == 0) break; /* a third contig: the estimate was off by more than a contig */
            }
        }
        if (at_q <= 0 || at_q >= size_q || bg_cut_describe(fd_q, size_q, at_q, &cq) != 1) continue;
        if (strcmp(ct.prev, cq.prev) || strcmp(ct.next, cq.next) || ct.prev_pos != cq.prev_pos || ct.next_pos != cq.next_pos) continue;
        if (n && at_q <= cut_q[n - 1]) continue;
        cut_t[n] = at_t;
        cut_q[n] = at_q;
        ++n;
    }
    return n + 1;
}

/* ---------------- several GPUs (CORNETTO_DEVICES): the window stage sharded by contig ----------------
 * get_regs() is independent per contig (src/boringbits_main.c:331); the thresholds depend on the assembly-wide mean
 * (:283-294 -> :518-519).  The text is parsed on the first device; its contigs are then dealt to the devices (longest first,
 * each to the least loaded: LPT) and copied there (cornetto_cov_shard: xGMI peer copies); every device gets one host thread
 * with its own handle: block sums + totals of its share, the host adds the 3 x u64 totals up, then the classification with
 * the common thresholds.  The selected windows come back per device and are put together in contig order: the printing
 * below is the same as with one device. */
typedef struct {
    int dev;
    cornetto_accel_t *h, *h_src;
    const cornetto_cov_t *src;
    cornetto_cov_t *part;
    int32_t *ctgs, n;
    int phase;
    uint64_t sums[3];
    const optp_t *opt;
    int32_t lo, hi;
    int8_t boring;
    cornetto_regrec_t *recs;
    int64_t n_recs;
    int rc;
    int asserted;  /* cornetto_cov_prepare() met an assert of get_regs() (CORNETTO_E_ASSERT): err holds the reference's line; reported by the
                      caller, behind the format errors of ALL shares (the reference parses everything before get_regs()) */
    char err[600];
    bg_job_t *ing; /* the device parses its own share of the text first (sharded ingest) */
} bb_dev_t;

/* get_regs() (src/boringbits_main.c:331-369) runs over EVERY contig — it knows no -m — before anything is printed: the first contig whose
 * last window is empty (:353) or stops short of the end (:368) ends the process the way the reference's assert does */
static void regs_assert_all(const int32_t *lens, int32_t n_ctg, const optp_t *opt)
{
    if (opt->window_inc < 1) return;
    for (int32_t i = 0; i < n_ctg; ++i) {
        const int line = cornetto_regs_assert(lens[i], opt->window_size, opt->window_inc);
        if (!line) continue;
        char msg[256];
        snprintf(msg, sizeof(msg), "src/boringbits_main.c:%d: get_regs: Assertion `%s' failed. (contig %d of length %d, -w %d -i %d)", line,
                 line == 353 ? "st<end" : "end == length", i, lens[i], opt->window_size, opt->window_inc);
        cli_ref_abort(msg);
    }
}

static void bb_prepare(bb_dev_t *d)
{
    d->rc = cornetto_cov_prepare(d->h, d->part, d->opt->window_size, d->opt->window_inc, d->sums);
    if (d->rc == CORNETTO_E_ASSERT) {
        snprintf(d->err, sizeof(d->err), "%s", cornetto_accel_last_error(d->h));
        d->asserted = 1;
        d->rc = CORNETTO_OK;
    }
}

static void *bb_worker(void *p)
{
    bb_dev_t *d = (bb_dev_t *)p;
    d->rc = CORNETTO_OK;
    if (d->phase == 0) {
        if (!d->h) {
            d->rc = cornetto_accel_open(&d->h, d->dev, NULL);
            if (d->rc != CORNETTO_OK) {
                snprintf(d->err, sizeof(d->err), "cannot open HIP device %d: %s", d->dev, cornetto_accel_strerror(d->rc));
                return NULL;
            }
        }
        if (d->ing) {
            d->ing->h = d->h;
            bg_ingest(d->ing);
            if (d->ing->rc != CORNETTO_OK || d->ing->fmt_kind) return NULL; /* (reported in the order of the shares by the caller) */
            d->part = d->ing->cov;
            d->n = d->ing->n_ctg;
            if (d->n > 0) bb_prepare(d);
            if (d->rc != CORNETTO_OK) snprintf(d->err, sizeof(d->err), "device %d: %s (%s)", d->dev, cornetto_accel_last_error(d->h), cornetto_accel_strerror(d->rc));
            return NULL;
        }
        if (d->n == 0) return NULL; /* more devices than contigs: nothing to do here */
        d->rc = cornetto_cov_shard(d->h_src, d->src, d->h, d->ctgs, d->n, &d->part);
        if (d->rc == CORNETTO_OK) bb_prepare(d);
    } else if (d->n > 0) {
        d->rc = cornetto_cov_select(d->h, d->part, d->lo, d->hi, d->opt->low_mq_cov_thresh, d->opt->edge_len, d->opt->min_ctg_len, d->boring, &d->recs, &d->n_recs);
    }
    if (d->rc != CORNETTO_OK && !d->err[0])
        snprintf(d->err, sizeof(d->err), "device %d: %s (%s)", d->dev, cornetto_accel_last_error(d->h), cornetto_accel_strerror(d->rc));
    return NULL;
}

static void bb_run(bb_dev_t *dv, int n_dev, int phase)
{
    pthread_t th[CLI_MAX_DEV];
    int started[CLI_MAX_DEV];
    for (int d = 0; d < n_dev; ++d) dv[d].phase = phase;
    for (int d = 0; d < n_dev; ++d) started[d] = d > 0 && pthread_create(&th[d], NULL, bb_worker, &dv[d]) == 0;
    for (int d = 0; d < n_dev; ++d)
        if (!started[d]) bb_worker(&dv[d]);
    for (int d = 0; d < n_dev; ++d)
        if (started[d]) pthread_join(th[d], NULL);
    for (int d = 0; d < n_dev; ++d)
        if (dv[d].rc != CORNETTO_OK) {
            CLI_ERROR("%s", dv[d].err);
            exit(EXIT_FAILURE);
        }
}

/* -> the selected windows of all contigs in contig order (cornetto_free), and the three totals */
static void bb_multi(cornetto_accel_t *h0, const cornetto_cov_t *cov, int32_t n_ctg, const int32_t *lens, const int *devs, int n_dev, const optp_t *opt,
                     int8_t boring, uint64_t sums[3], int32_t *mean_depth, int32_t *mean_mq, cornetto_regrec_t **recs, int64_t *n_recs)
{
    bb_dev_t dv[CLI_MAX_DEV];
    memset(dv, 0, sizeof(dv));
    int32_t *order = (int32_t *)cli_xmalloc(((size_t)n_ctg + 1) * sizeof(int32_t));
    int32_t *owner = (int32_t *)cli_xmalloc(((size_t)n_ctg + 1) * sizeof(int32_t));
    cli_order_by_length_desc(NULL, lens, n_ctg, order);   /* by descending length, ties in input order (read-level coverage sets: 10^5..10^6 contigs) */
    int64_t load[CLI_MAX_DEV];
    for (int d = 0; d < n_dev; ++d) {
        load[d] = 0;
        dv[d].dev = devs[d];
        dv[d].h = d == 0 ? h0 : NULL;
        dv[d].h_src = h0;
        dv[d].src = cov;
        dv[d].opt = opt;
        dv[d].boring = boring;
        dv[d].ctgs = (int32_t *)cli_xmalloc(((size_t)n_ctg + 1) * sizeof(int32_t));
    }
    for (int32_t k = 0; k < n_ctg; ++k) {
        int best = 0;
        for (int d = 1; d < n_dev; ++d)
            if (load[d] < load[best]) best = d;
        owner[order[k]] = best;
        load[best] += (int64_t)lens[order[k]] + 1;
    }
    for (int32_t i = 0; i < n_ctg; ++i) dv[owner[i]].ctgs[dv[owner[i]].n++] = i; /* ascending: a device's contigs keep their input order */
    bb_run(dv, n_dev, 0);
    for (int d = 0; d < n_dev; ++d)
        if (dv[d].asserted) regs_assert_all(lens, n_ctg, opt);           /* (names the first contig in INPUT order, as the reference would) */
    sums[0] = sums[1] = sums[2] = 0;
    for (int d = 0; d < n_dev; ++d)
        for (int k = 0; k < 3; ++k) sums[k] += dv[d].sums[k];            /* the one exchange: 3 x u64 per device */
    *mean_depth = (int32_t)round((double)(int64_t)sums[0] / (double)sums[2]);     /* :293 (the totals are signed: negative depth values in the text count as they are) */
    *mean_mq = (int32_t)round((double)(int64_t)sums[1] / (double)sums[2]);        /* :294 */
    for (int d = 0; d < n_dev; ++d) {
        dv[d].lo = cornetto_cov_threshold(opt->low_cov_thresh, *mean_depth);   /* :518 */
        dv[d].hi = cornetto_cov_threshold(opt->high_cov_thresh, *mean_depth);  /* :519 */
    }
    bb_run(dv, n_dev, 1);
    int64_t total = 0;
    for (int d = 0; d < n_dev; ++d) total += dv[d].n_recs;
    cornetto_regrec_t *all = (cornetto_regrec_t *)cli_xmalloc(((size_t)total + 1) * sizeof(*all));
    int64_t cur[CLI_MAX_DEV], at = 0;
    int32_t local[CLI_MAX_DEV];
    for (int d = 0; d < n_dev; ++d) { cur[d] = 0; local[d] = 0; }
    for (int32_t i = 0; i < n_ctg; ++i) {
        bb_dev_t *d = &dv[owner[i]];
        const int32_t li = local[owner[i]]++;
        int64_t *c = &cur[owner[i]];
        for (; *c < d->n_recs && d->recs[*c].ctg == li; ++*c) {
            all[at] = d->recs[*c];
            all[at++].ctg = i;
        }
    }
    for (int d = 0; d < n_dev; ++d) {
        cornetto_free(dv[d].recs);
        if (dv[d].part) cornetto_cov_free(dv[d].h, dv[d].part);
        if (d > 0 && dv[d].h) cornetto_accel_close(dv[d].h);
        free(dv[d].ctgs);
    }
    free(order);
    free(owner);
    *recs = all;
    *n_recs = at;
}

/* Sharded ingest: device d parses share d of the text (whole contigs: bg_split), sums and classifies the contigs it parsed — the text of
 * 3 Gbp is 2 x 100 GB: every device reads its part of the files over its own PCIe link, and no depth array moves between devices.
 * The host adds the three totals up for the common thresholds.  -> names / lengths / selected windows of all contigs in file order */
static void bb_sharded(cornetto_accel_t *h0, bg_job_t *jobs, const int *devs, int n_sh, const optp_t *opt, int8_t boring, const char *name_t, const char *name_q,
                       uint64_t sums[3], int32_t *mean_depth, int32_t *mean_mq, int32_t *n_ctg_out, char ***names_out, int32_t **lens_out,
                       cornetto_regrec_t **recs, int64_t *n_recs)
{
    bb_dev_t dv[CLI_MAX_DEV];
    memset(dv, 0, sizeof(dv));
    for (int d = 0; d < n_sh; ++d) {
        dv[d].dev = devs[d];
        dv[d].h = d == 0 ? h0 : NULL;
        dv[d].opt = opt;
        dv[d].boring = boring;
        dv[d].ing = &jobs[d];
    }
    bb_run(dv, n_sh, 0);
    int64_t clamped = 0;
    int32_t n_ctg = 0;
    for (int d = 0; d < n_sh; ++d) { /* what the sequential parse would have met first */
        bg_job_check(&jobs[d], name_t, name_q);
        clamped += jobs[d].n_clamped;
        n_ctg += jobs[d].n_ctg;
    }
    if (clamped) CLI_WARNING("%lld depth values were truncated to 65535", (long long)clamped);
    for (int d = 0; d < n_sh; ++d)                                       /* shares are in file order: the first one names the assert */
        if (dv[d].asserted) cli_ref_abort(dv[d].err);
    sums[0] = sums[1] = sums[2] = 0;
    for (int d = 0; d < n_sh; ++d)
        for (int k = 0; k < 3; ++k) sums[k] += dv[d].sums[k];            /* the one exchange: 3 x u64 per device */
    *mean_depth = (int32_t)round((double)(int64_t)sums[0] / (double)sums[2]);     /* :293 (the totals are signed: negative depth values in the text count as they are) */
    *mean_mq = (int32_t)round((double)(int64_t)sums[1] / (double)sums[2]);        /* :294 */
    for (int d = 0; d < n_sh; ++d) {
        dv[d].lo = cornetto_cov_threshold(opt->low_cov_thresh, *mean_depth);   /* :518 */
        dv[d].hi = cornetto_cov_threshold(opt->high_cov_thresh, *mean_depth);  /* :519 */
    }
    bb_run(dv, n_sh, 1);
    int64_t total = 0;
    for (int d = 0; d < n_sh; ++d) total += dv[d].n_recs;
    cornetto_regrec_t *all = (cornetto_regrec_t *)cli_xmalloc(((size_t)total + 1) * sizeof(*all));
    char **names = (char **)cli_xmalloc(((size_t)n_ctg + 1) * sizeof(char *));
    int32_t *lens = (int32_t *)cli_xmalloc(((size_t)n_ctg + 1) * sizeof(int32_t));
    int64_t at = 0;
    int32_t base = 0;
    for (int d = 0; d < n_sh; ++d) {
        const int32_t *l = jobs[d].n_ctg ? cornetto_cov_lens(jobs[d].cov) : NULL;
        for (int32_t i = 0; i < jobs[d].n_ctg; ++i) {
            names[base + i] = jobs[d].names[i];
            lens[base + i] = l[i];
        }
        for (int64_t k = 0; k < dv[d].n_recs; ++k) {
            all[at] = dv[d].recs[k];
            all[at++].ctg += base;
        }
        base += jobs[d].n_ctg;
        cornetto_free(dv[d].recs);
        free(jobs[d].names);
        if (jobs[d].cov) cornetto_cov_free(dv[d].h, jobs[d].cov);
        if (d > 0 && dv[d].h) cornetto_accel_close(dv[d].h);
    }
    *n_ctg_out = n_ctg;
    *names_out = names;
    *lens_out = lens;
    *recs = all;
    *n_recs = at;
}

static void print_params(int32_t n_ctg, int32_t mean_depth, int32_t mean_mq, const optp_t *opt) /* :497-506 */
{
    fprintf(stderr, "Number of contigs: %d\n", n_ctg);
    fprintf(stderr, "Average depth: %d\n", mean_depth);
    fprintf(stderr, "Average mq depth: %d\n", mean_mq);
    fprintf(stderr, "Window size: %d\n", opt->window_size);
    fprintf(stderr, "Window increment: %d\n", opt->window_inc);
    fprintf(stderr, "Low coverage threshold: %.1fx%d\n", opt->low_cov_thresh, mean_depth);
    fprintf(stderr, "High coverage threshold: %.1fx%d\n", opt->high_cov_thresh, mean_depth);
    fprintf(stderr, "Low mapq coverage threshold: %.1f\n", opt->low_mq_cov_thresh);
    fprintf(stderr, "Min contig length: %d\n", opt->min_ctg_len);
    fprintf(stderr, "Edge length: %d\n", opt->edge_len);
}

/* print_fun_bits / print_boring_bits (:425-445 / :463-481) over the selected rows, contig by contig */
static void print_bits(char **names, const int32_t *lens, int32_t n_ctg, const cornetto_regrec_t *recs, int64_t n_recs, const optp_t *opt, int8_t boring)
{
    int64_t k = 0;
    for (int32_t i = 0; i < n_ctg; ++i) {
        const char *name = names[i];
        const size_t name_l = strlen(name);
        const int len = lens[i];
        if (!boring) {
            if (len < opt->min_ctg_len) {
                cli_out_flush();
                printf("%s\t%d\t%d\t.\t.\n", name, 0, opt->min_ctg_len); /* :430 prints min_ctg_len, not len */
            } else {
                cli_out_flush();
                printf("%s\t%d\t%d\t.\t.\n", name, 0, opt->edge_len);
                printf("%s\t%d\t%d\t.\t.\n", name, len - opt->edge_len, len);
            }
        }
        for (; k < n_recs && recs[k].ctg == i; ++k)
            { /* name, st, end, depth, mq_depth: :441 / :475 */
                cli_out_bytes(name, name_l);
                cli_out_char('\t');
                cli_out_int(recs[k].st);
                cli_out_char('\t');
                cli_out_int(recs[k].end);
                cli_out_char('\t');
                cli_out_int(recs[k].depth);
                cli_out_char('\t');
                cli_out_int(recs[k].mq_depth);
                cli_out_char('\n');
            }
    }
    cli_out_flush();
}

/* --accel=no / CORNETTO_ACCEL=no: the whole sub-command on the host (cli/host_backend.c), the_boring_bits() :483-536 */
static int host_bits(FILE *ft, FILE *fq, const optp_t *opt, int8_t boring, const char *panel_bed, const char *lowq_bed)
{
    double t0 = cli_realtime();
    cli_host_cov_t cov;
    cli_host_get_depths(ft, fq, &cov);
    fclose(ft);
    fclose(fq);
    if (cov.n_clamped) CLI_WARNING("%lld depth values were truncated to 65535", (long long)cov.n_clamped);
    CLI_VERBOSE("Loaded depth files in %.2f seconds", cli_realtime() - t0);
    int32_t mean_depth = (int32_t)0x80000000, mean_mq = (int32_t)0x80000000;   /* (no record at all: the reference rounds 0 / 0) */
    if (cov.n_ctg > 0) {
        mean_depth = (int32_t)round(cov.sum_depth / cov.positions);           /* :293 */
        mean_mq = (int32_t)round(cov.sum_mq / cov.positions);                 /* :294 */
    }
    print_params(cov.n_ctg, mean_depth, mean_mq, opt);
    t0 = cli_realtime();
    cornetto_regrec_t *recs = NULL;
    int64_t n_recs = 0;
    if (cov.n_ctg > 0) {
        const int32_t lo_t = cornetto_cov_threshold(opt->low_cov_thresh, mean_depth);   /* :518 */
        const int32_t hi_t = cornetto_cov_threshold(opt->high_cov_thresh, mean_depth);  /* :519 */
        /* get_regs() (:331-369) runs over EVERY contig — it knows no -m — before anything is printed: its asserts end the process here */
        regs_assert_all(cov.lens, cov.n_ctg, opt);
        cli_host_cov_select(&cov, opt->window_size, opt->window_inc, lo_t, hi_t, opt->low_mq_cov_thresh, opt->edge_len, opt->min_ctg_len, panel_bed ? 0 : boring,
                            &recs, &n_recs);
    }
    CLI_VERBOSE("Found regions in %.2f seconds", cli_realtime() - t0);
    if (panel_bed) {
        cornetto_ivl_t *fun = NULL;
        int64_t n_fun = 0;
        cli_host_merge_windows(recs, n_recs, pn_par[0], pn_par[1], &fun, &n_fun);       /* create-cornetto.sh:41-47 */
        panel_print(panel_bed, lowq_bed, cov.names, cov.n_ctg, fun, n_fun);
        free(fun);
    } else {
        t0 = cli_realtime();
        print_bits(cov.names, cov.lens, cov.n_ctg, recs, n_recs, opt, boring);
        CLI_VERBOSE("Printed the bits in %.2f seconds", cli_realtime() - t0);
    }
    free(recs);
    cli_host_cov_free(&cov);
    return 0;
}

int boringbits_main(int argc, char *argv[], int8_t boring)
{
    static const struct option lo[] = {
        {"threads", required_argument, 0, 't'},   {"batchsize", required_argument, 0, 'K'},
        {"max-bytes", required_argument, 0, 'B'}, {"verbose", required_argument, 0, 'v'},
        {"help", no_argument, 0, 'h'},            {"version", no_argument, 0, 'V'},
        {"output", required_argument, 0, 'o'},    {"debug-break", required_argument, 0, 0},
        {"profile-cpu", required_argument, 0, 0}, {"accel", required_argument, 0, 0},
        {"qual", required_argument, 0, 'q'},      {"window-size", required_argument, 0, 'w'},
        {"window-inc", required_argument, 0, 'i'}, {"low-thresh", required_argument, 0, 'L'},
        {"high-thresh", required_argument, 0, 'H'}, {"low-mq-thresh", required_argument, 0, 'Q'},
        {"min-ctg-len", required_argument, 0, 'm'}, {"edge-len", required_argument, 0, 'e'},
        {"panel", required_argument, 0, 0},       {"lowq", required_argument, 0, 0},
        {"panel-params", required_argument, 0, 0},
        {0, 0, 0, 0}};
    optp_t opt = {2500, 50, 0.4f, 2.5f, 0.4f, 1000000, 100000}; /* :540-556 */
    const char *covmq = NULL, *panel_bed = NULL, *lowq_bed = NULL;
    FILE *fp_help = stderr;
    int c, li = 0;
    optind = 1;
    while ((c = getopt_long(argc, argv, "t:B:K:v:o:q:Q:H:L:w:i:e:m:hV", lo, &li)) >= 0) {
        if (c == 'K') {
            if (atoi(optarg) < 1) {
                CLI_ERROR("Batch size should larger than 0. You entered %d", atoi(optarg));
                exit(EXIT_FAILURE);
            }
        } else if (c == 't') {
            if (atoi(optarg) < 1) {
                CLI_ERROR("Number of threads should larger than 0. You entered %d", atoi(optarg));
                exit(EXIT_FAILURE);
            }
        } else if (c == 'v') {
            cli_log_level = atoi(optarg);
        } else if (c == 'V') {
            fprintf(stdout, "cornetto %s\n", CORNETTO_VERSION);
            exit(EXIT_SUCCESS);
        } else if (c == 'h') {
            fp_help = stdout;
        } else if (c == 'q') {
            covmq = optarg;
        } else if (c == 'w') {
            opt.window_size = atoi(optarg);
        } else if (c == 'i') {
            opt.window_inc = atoi(optarg);
        } else if (c == 'L') {
            opt.low_cov_thresh = atof(optarg);
        } else if (c == 'H') {
            opt.high_cov_thresh = atof(optarg);
        } else if (c == 'Q') {
            opt.low_mq_cov_thresh = atof(optarg);
        } else if (c == 'm') {
            opt.min_ctg_len = atoi(optarg);
        } else if (c == 'e') {
            opt.edge_len = atoi(optarg);
        } else if (c == 0 && li == 18) {
            panel_bed = optarg;
        } else if (c == 0 && li == 19) {
            lowq_bed = optarg;
        } else if (c == 0 && li == 20) { /* the seven constants of create-cornetto.sh:44-65, in the order they appear */
            const int got = sscanf(optarg, "%d,%d,%d,%d,%d,%d,%d,%d,%d", &pn_par[0], &pn_par[1], &pn_par[2], &pn_par[3], &pn_par[4], &pn_par[5], &pn_par[6],
                                   &pn_par[7], &pn_par[8]);
            if (got != 7 && got != 9) {
                CLI_ERROR("%s", "--panel-params wants seven integers: merge-d,min-fun-len,min-lowq-len,extend,edge-len,merge-d2,min-ctg-len (or nine: ...,extend-right,extend-gate)");
                exit(EXIT_FAILURE);
            }
        } else if (c == 0 && li == 9) { /* --accel: the seam the reference left (src/boringbits_main.c:627-632) */
            if (strcmp(optarg, "no") == 0 || strcmp(optarg, "n") == 0) {
                cli_host_set(1);       /* the host path of cli/host_backend.c: sequential parse, window sums, selection */
            } else if (strcmp(optarg, "yes") == 0 || strcmp(optarg, "y") == 0) {
                cli_host_set(0);
            } else {
                fprintf(stderr, "option '--accel' only accepts 'yes' or 'no'.\n");
            }
        }
    }
    if (argc - optind != 1 || fp_help == stdout) { /* :638-644 */
        print_help(fp_help, &opt);
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }
    const char *covtotal = argv[optind];
    if (!covmq) {
        print_help(fp_help, &opt);
        exit(EXIT_FAILURE);
    }
    if ((panel_bed || lowq_bed) && (boring || !panel_bed)) {
        CLI_ERROR("%s", "--panel / --lowq: only with noboringbits, and --lowq needs --panel");
        exit(EXIT_FAILURE);
    }

    /* ---------------- get_depths (:180-301): text -> uint16 arrays, on the device ---------------- */
    double t0 = cli_realtime();
    FILE *ft = fopen(covtotal, "r");
    if (!ft) {
        CLI_ERROR("Failed to open %s : No such file or directory.", covtotal);
        exit(EXIT_FAILURE);
    }
    FILE *fq = fopen(covmq, "r");
    if (!fq) {
        CLI_ERROR("Failed to open %s : No such file or directory.", covmq);
        exit(EXIT_FAILURE);
    }
    if (cli_host_mode()) return host_bits(ft, fq, &opt, boring, panel_bed, lowq_bed);
    int devs[CLI_MAX_DEV];
    const int n_dev = cli_device_list(devs);
    if (n_dev >= 1) { /* the text is parsed on the first listed device */
        char one[32];
        snprintf(one, sizeof(one), "%d", devs[0]);
        setenv("CORNETTO_DEVICE", one, 1);
    }
    /* Two regular files: pread() threads fill one pair of pinned pieces while the device takes the other (a single fread() stream copies
     * from the page cache at 5-8 GB/s: 0.85 of the 1.3 s of a 6.8 GB pair), and with several devices every device parses its own share
     * of the text (bb_sharded).  Anything else (FIFOs, process substitution): the sequential loop below on the first device.
     * $CORNETTO_BG_THREADS=0 asks for the sequential loop, $CORNETTO_BG_PIECE sets the bytes of each file per round (tests),
     * $CORNETTO_BG_SHARD_MIN the bytes a file must have per device before it is cut (64 MB; tests: 1). */
    struct stat st_t, st_q;
    int n_rd = 8;
    if (getenv("CORNETTO_BG_THREADS")) n_rd = atoi(getenv("CORNETTO_BG_THREADS"));
    const int threaded = n_rd > 0 && fstat(fileno(ft), &st_t) == 0 && fstat(fileno(fq), &st_q) == 0 && S_ISREG(st_t.st_mode) && S_ISREG(st_q.st_mode);
    int64_t piece = threaded ? 64ll << 20 : 256ll << 20; /* bytes of each file kept in flight */
    if (getenv("CORNETTO_BG_PIECE") && atoll(getenv("CORNETTO_BG_PIECE")) > 0) piece = atoll(getenv("CORNETTO_BG_PIECE"));
    int64_t cut_t[CLI_MAX_DEV], cut_q[CLI_MAX_DEV];
    int n_sh = 1;
    if (threaded) {
        int64_t shard_min = 64ll << 20;
        if (getenv("CORNETTO_BG_SHARD_MIN") && atoll(getenv("CORNETTO_BG_SHARD_MIN")) > 0) shard_min = atoll(getenv("CORNETTO_BG_SHARD_MIN"));
        if (n_dev >= 2 && !panel_bed && (int64_t)st_t.st_size >= shard_min * n_dev && (int64_t)st_q.st_size >= shard_min * n_dev)
            n_sh = bg_split(fileno(ft), (int64_t)st_t.st_size, fileno(fq), (int64_t)st_q.st_size, n_dev, cut_t, cut_q);
        if (getenv("CORNETTO_BG_SPLIT_ONLY")) { /* (tests: the cuts, without a device) */
            for (int d = 0; d + 1 < n_sh; ++d) printf("%lld\t%lld\n", (long long)cut_t[d], (long long)cut_q[d]);
            exit(EXIT_SUCCESS);
        }
    }
    cli_accel_open_begin();
    cornetto_accel_t *h = NULL;
    cornetto_cov_t *cov = NULL;
    int32_t n_ctg = 0;
    char **names = NULL;
    int64_t n_clamped = 0;
    const int32_t *lens = NULL;
    int32_t *lens_owned = NULL;
    int32_t mean_depth = 0, mean_mq = 0;
    cornetto_regrec_t *recs = NULL;
    cornetto_ivl_t *fun = NULL;
    int64_t n_recs = 0, n_fun = 0;
    int recs_are_malloced = 0, stage_done = 0;
    if (threaded) {
        bg_job_t jobs[CLI_MAX_DEV];
        memset(jobs, 0, sizeof(jobs));
        for (int d = 0; d < n_sh; ++d) {
            jobs[d].fd_t = fileno(ft);
            jobs[d].fd_q = fileno(fq);
            jobs[d].n_rd = n_sh > 1 ? (n_rd / n_sh > 2 ? n_rd / n_sh : 2) : n_rd;
            jobs[d].piece = piece;
            jobs[d].t0 = d ? cut_t[d - 1] : 0;
            jobs[d].q0 = d ? cut_q[d - 1] : 0;
            jobs[d].not_last = d + 1 < n_sh;
            jobs[d].t1 = d + 1 < n_sh ? cut_t[d] : (int64_t)st_t.st_size;
            jobs[d].q1 = d + 1 < n_sh ? cut_q[d] : (int64_t)st_q.st_size;
        }
        h = cli_accel_open_end();
        if (n_sh > 1) {
            uint64_t sums[3];
            bb_sharded(h, jobs, devs, n_sh, &opt, boring, covtotal, covmq, sums, &mean_depth, &mean_mq, &n_ctg, &names, &lens_owned, &recs, &n_recs);
            lens = lens_owned;
            recs_are_malloced = 1;
            stage_done = 1;
            CLI_VERBOSE("Loaded depth files and found regions on %d devices (sharded ingest) in %.2f seconds", n_sh, cli_realtime() - t0);
        } else {
            jobs[0].h = h;
            bg_ingest(&jobs[0]);
            bg_job_check(&jobs[0], covtotal, covmq);
            cov = jobs[0].cov;
            n_ctg = jobs[0].n_ctg;
            names = jobs[0].names;
            n_clamped = jobs[0].n_clamped;
        }
    } else {
        char *buf_t = (char *)cornetto_pinned_alloc((size_t)piece), *buf_q = (char *)cornetto_pinned_alloc((size_t)piece);
        if (!buf_t || !buf_q) {
            (void)cli_accel_open_end(); /* no usable device: its message and exit(EXIT_FAILURE) */
            CLI_ERROR("%s", "cannot allocate pinned read buffers");
            exit(EXIT_FAILURE);
        }
        h = cli_accel_open_end();
        cornetto_bgin_t *bg = NULL;
        cli_accel_check(h, cornetto_bgin_open(h, &bg), "bedgraph ingest");
        int eof_t = 0, eof_q = 0;
        bg_job_t g;
        memset(&g, 0, sizeof(g));
        while (!cornetto_bgin_done(bg)) {
            int64_t pend_t = 0, pend_q = 0;
            cornetto_bgin_pending(bg, &pend_t, &pend_q);
            /* top both files up to the same number of pending bytes, so the unmatched tail of either stays small */
            const int64_t least = piece < 4096 ? piece : 4096;
            size_t want_t = eof_t ? 0 : (size_t)(pend_t < piece - least ? piece - pend_t : least);
            size_t want_q = eof_q ? 0 : (size_t)(pend_q < piece - least ? piece - pend_q : least);
            size_t got_t = want_t ? fread(buf_t, 1, want_t, ft) : 0, got_q = want_q ? fread(buf_q, 1, want_q, fq) : 0;
            if (got_t < want_t) eof_t = 1;
            if (got_q < want_q) eof_q = 1;
            int rc = cornetto_bgin_feed(h, bg, buf_t, (int64_t)got_t, buf_q, (int64_t)got_q, eof_t | (eof_q << 1));
            if (rc == CORNETTO_E_FORMAT) {
                const cornetto_bgerr_t *e = cornetto_bgin_error(bg);
                g.fmt_kind = e->kind ? e->kind : 5;
                g.fmt_a = e->a;
                g.fmt_b = e->b;
                bg_job_check(&g, covtotal, covmq);
            }
            cli_accel_check(h, rc, "bedgraph ingest");
            if (eof_t && eof_q && !cornetto_bgin_done(bg)) { /* cannot happen: the final feed either finishes or fails */
                CLI_ERROR("%s", "bedgraph ingest did not finish");
                exit(EXIT_FAILURE);
            }
        }
        cornetto_pinned_free(buf_t);
        cornetto_pinned_free(buf_q);
        cli_accel_check(h, cornetto_bgin_finish(h, bg, &cov, &n_ctg, &names, &n_clamped), "bedgraph ingest");
        cornetto_bgin_close(h, bg);
    }
    fclose(ft);
    fclose(fq);
    if (n_clamped) CLI_WARNING("%lld depth values were truncated to 65535", (long long)n_clamped);
    if (!stage_done) {
        lens = cornetto_cov_lens(cov);
        CLI_VERBOSE("Loaded depth files in %.2f seconds", cli_realtime() - t0);
    }

    /* ---------------- device: totals, windows, selection ---------------- */
    if (stage_done) {
        if (n_ctg == 0) mean_depth = mean_mq = (int32_t)0x80000000;
    } else if (n_ctg > 0 && n_dev >= 2 && !panel_bed) {
        t0 = cli_realtime();
        uint64_t sums[3];
        bb_multi(h, cov, n_ctg, lens, devs, n_dev, &opt, boring, sums, &mean_depth, &mean_mq, &recs, &n_recs);
        recs_are_malloced = 1;
        CLI_VERBOSE("Found regions on %d devices in %.2f seconds", n_dev, cli_realtime() - t0);
    } else if (n_ctg > 0) {
        t0 = cli_realtime();
        uint64_t sums[3];
        cli_accel_check(h, cornetto_cov_prepare(h, cov, opt.window_size, opt.window_inc, sums), "window block sums");
        /* double accumulators of the reference hold these integers exactly (:283-285) */
        mean_depth = (int32_t)round((double)(int64_t)sums[0] / (double)sums[2]);   /* :293 (the totals are signed: negative depth values in the text count as they are) */
        mean_mq = (int32_t)round((double)(int64_t)sums[1] / (double)sums[2]);      /* :294 */
        const int32_t lo_t = cornetto_cov_threshold(opt.low_cov_thresh, mean_depth);   /* :518 */
        const int32_t hi_t = cornetto_cov_threshold(opt.high_cov_thresh, mean_depth);  /* :519 */
        if (panel_bed)
            cli_accel_check(h, cornetto_cov_select_merged(h, cov, lo_t, hi_t, opt.low_mq_cov_thresh, opt.edge_len, opt.min_ctg_len, 0, pn_par[0], pn_par[1],
                                                          &fun, &n_fun), "window classification + merge");   /* create-cornetto.sh:41-47 */
        else
            cli_accel_check(h, cornetto_cov_select(h, cov, lo_t, hi_t, opt.low_mq_cov_thresh, opt.edge_len, opt.min_ctg_len, boring, &recs, &n_recs),
                            "window classification");
        CLI_VERBOSE("Found regions in %.2f seconds", cli_realtime() - t0);
    } else {
        /* the reference divides 0/0 here: round(NaN) -> INT_MIN; nothing is printed either way */
        mean_depth = mean_mq = (int32_t)0x80000000;
    }
    print_params(n_ctg, mean_depth, mean_mq, &opt);

    if (panel_bed) {
        panel_print(panel_bed, lowq_bed, names, n_ctg, fun, n_fun);
        cornetto_free(fun);
        for (int32_t i = 0; i < n_ctg; ++i) free(names[i]);
        free(names);
        cornetto_cov_free(h, cov);
        cornetto_accel_close(h);
        return 0;
    }

    /* ---------------- print (:425-445 / :463-481) ---------------- */
    t0 = cli_realtime();
    print_bits(names, lens, n_ctg, recs, n_recs, &opt, boring);
    CLI_VERBOSE("Printed the bits in %.2f seconds", cli_realtime() - t0);
    if (recs_are_malloced) free(recs);
    else cornetto_free(recs);
    for (int32_t i = 0; i < n_ctg; ++i) free(names[i]);
    free(names);
    free(lens_owned);
    if (cov) cornetto_cov_free(h, cov);
    cornetto_accel_close(h);
    return 0;
}

/* boringbits_main.c — `cornetto noboringbits|boringbits cov-total.bg -q cov-mq20.bg [options]`.
 * Reference: src/boringbits_main.c:558-660 (options), :180-301 (get_depths: lock-step parse + validation),
 * :483-536 (the_boring_bits), :425-445 / :463-481 (printing).
 * Host: streaming text parse of the two per-base bedgraphs into uint16 arrays with the reference's checks.
 * Device: block sums, totals for the mean, window means, classification, ordered selection. */
#include <getopt.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cli.h"

typedef struct {
    int window_size, window_inc;
    float low_cov_thresh, high_cov_thresh, low_mq_cov_thresh;
    int min_ctg_len, edge_len;
} optp_t;

/* ---- buffered token reader equivalent to fscanf(fp, "%s\t%d\t%d\t%d\n", ...) (:205,:214) ---- */
typedef struct {
    FILE *fp;
    char *buf;
    size_t begin, end, cap;
    int eof;
} trd_t;

static void trd_open(trd_t *t, const char *path)
{
    memset(t, 0, sizeof(*t));
    t->fp = fopen(path, "r");
    if (!t->fp) {
        CLI_ERROR("Failed to open %s : No such file or directory.", path);
        exit(EXIT_FAILURE);
    }
    t->cap = 1 << 24;
    t->buf = (char *)cli_xmalloc(t->cap + 1);
}

static int trd_fill(trd_t *t)
{
    if (t->eof) return 0;
    size_t keep = t->end - t->begin;
    memmove(t->buf, t->buf + t->begin, keep);
    t->begin = 0;
    t->end = keep;
    size_t got = fread(t->buf + t->end, 1, t->cap - t->end, t->fp);
    if (got == 0) {
        t->eof = 1;
        return 0;
    }
    t->end += got;
    return 1;
}

static inline int is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

/* make sure a whole token (up to 10000 bytes, like the reference's buffers) starting at begin is in memory */
static inline void trd_need(trd_t *t)
{
    if (t->end - t->begin < 12000 && !t->eof) trd_fill(t);
}

/* returns the number of fields converted (0..4) or EOF (-1) exactly like the fscanf call */
static int trd_record(trd_t *t, char **name, size_t *name_len, int *st, int *end, int *depth)
{
    trd_need(t);
    while (t->begin < t->end && is_ws(t->buf[t->begin])) {
        ++t->begin;
        if (t->begin == t->end) trd_need(t);
    }
    if (t->begin >= t->end) return -1; /* EOF before any conversion */
    trd_need(t);
    size_t p = t->begin;
    while (p < t->end && !is_ws(t->buf[p])) ++p;
    *name = t->buf + t->begin;
    *name_len = p - t->begin;
    int got = 1;
    int *out[3] = {st, end, depth};
    for (int k = 0; k < 3; ++k) {
        while (p < t->end && is_ws(t->buf[p])) ++p;
        if (p >= t->end) break;
        size_t q = p;
        int neg = 0;
        if (t->buf[q] == '-' || t->buf[q] == '+') neg = t->buf[q++] == '-';
        if (q >= t->end || t->buf[q] < '0' || t->buf[q] > '9') break;
        long v = 0;
        while (q < t->end && t->buf[q] >= '0' && t->buf[q] <= '9') v = v * 10 + (t->buf[q++] - '0');
        *out[k] = (int)(neg ? -v : v);
        p = q;
        ++got;
    }
    t->begin = p; /* the trailing "\n" of the format eats white space before the next record */
    return got;
}

typedef struct {
    char *name;
    int32_t len, cap;
    uint16_t *depth, *mq;
} ctgd_t;

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
        {0, 0, 0, 0}};
    optp_t opt = {2500, 50, 0.4f, 2.5f, 0.4f, 1000000, 100000}; /* :540-556 */
    const char *covmq = NULL;
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
        } else if (c == 0 && li == 9) { /* --accel: the seam the reference left (src/boringbits_main.c:627-632) */
            if (strcmp(optarg, "no") == 0 || strcmp(optarg, "n") == 0) {
                CLI_ERROR("%s", "--accel=no: this build has no CPU path for the window stage; use the reference binary");
                exit(EXIT_FAILURE);
            } else if (!(strcmp(optarg, "yes") == 0 || strcmp(optarg, "y") == 0)) {
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

    /* ---------------- get_depths (:180-301) ---------------- */
    double t0 = cli_realtime();
    trd_t ft, fq;
    trd_open(&ft, covtotal);
    trd_open(&fq, covmq);
    ctgd_t *ctg = NULL;
    int32_t n_ctg = 0, cap_ctg = 0;
    int prev_pos = 0;
    for (;;) {
        char *n1, *n2;
        size_t l1, l2;
        int st1 = 0, end1 = 0, d1 = 0, st2 = 0, end2 = 0, d2 = 0;
        int ret = trd_record(&ft, &n1, &l1, &st1, &end1, &d1);
        if (ret == -1) break;
        if (ret != 4) {
            CLI_ERROR("The depth files should have 4 columns. Had %d.", ret);
            exit(EXIT_FAILURE);
        }
        ret = trd_record(&fq, &n2, &l2, &st2, &end2, &d2);
        if (ret == -1) {
            CLI_ERROR("%s", "The two files are not in the same order");
            exit(EXIT_FAILURE);
        }
        if (ret != 4) {
            CLI_ERROR("The depth files should have 4 columns. Had %d.", ret);
            exit(EXIT_FAILURE);
        }
        if (l1 != l2 || memcmp(n1, n2, l1) != 0 || st1 != st2 || end1 != end2) { /* :224 */
            CLI_ERROR("%s", "The two files are not in the same order");
            exit(EXIT_FAILURE);
        }
        ctgd_t *cur = n_ctg ? &ctg[n_ctg - 1] : NULL;
        if (!cur || strlen(cur->name) != l1 || memcmp(cur->name, n1, l1) != 0) { /* :229 new contig */
            if (n_ctg == cap_ctg) {
                cap_ctg = cap_ctg ? cap_ctg * 2 : 16;
                ctg = (ctgd_t *)cli_xrealloc(ctg, (size_t)cap_ctg * sizeof(ctgd_t));
            }
            cur = &ctg[n_ctg++];
            cur->name = (char *)cli_xmalloc(l1 + 1);
            memcpy(cur->name, n1, l1);
            cur->name[l1] = 0;
            cur->len = 0;
            cur->cap = 1 << 16;
            cur->depth = (uint16_t *)cli_xmalloc((size_t)cur->cap * 2);
            cur->mq = (uint16_t *)cli_xmalloc((size_t)cur->cap * 2);
            prev_pos = 0; /* the first record of a contig is not checked for st == 0 */
        } else {
            if (prev_pos + 1 != st1) { /* :249 */
                CLI_ERROR("The depth files should be incremantal at one base resolution. Found %d to %d", prev_pos, st1);
                exit(EXIT_FAILURE);
            }
            prev_pos++;
        }
        if (st1 + 1 != end1) { /* :256 */
            CLI_ERROR("The depth files should have end=start+1. Found %d to %d", st1, end1);
            exit(EXIT_FAILURE);
        }
        if (d1 > 65535) { /* :261-268 */
            CLI_WARNING("The depth at %s:%d-%d was truncated to 65535. Found %d", cur->name, st1, end1, d1);
            d1 = 65535;
        }
        if (d2 > 65535) {
            CLI_WARNING("The depth at %s:%d-%d was truncated to 65535. Found %d", cur->name, st2, end2, d2);
            d2 = 65535;
        }
        if (cur->len == cur->cap) {
            if (cur->cap > 0x3fffffff) {
                CLI_ERROR("contig %s has more than 2^30 positions", cur->name);
                exit(EXIT_FAILURE);
            }
            cur->cap *= 2;
            cur->depth = (uint16_t *)cli_xrealloc(cur->depth, (size_t)cur->cap * 2);
            cur->mq = (uint16_t *)cli_xrealloc(cur->mq, (size_t)cur->cap * 2);
        }
        cur->depth[cur->len] = (uint16_t)d1; /* negative values wrap through uint16_t as in the reference */
        cur->mq[cur->len] = (uint16_t)d2;
        cur->len++;
    }
    fclose(ft.fp);
    fclose(fq.fp);
    free(ft.buf);
    free(fq.buf);
    CLI_VERBOSE("Loaded depth files in %.2f seconds", cli_realtime() - t0);

    /* ---------------- device: totals, windows, selection ---------------- */
    int32_t mean_depth = 0, mean_mq = 0;
    cornetto_regrec_t *recs = NULL;
    int64_t n_recs = 0;
    if (n_ctg > 0) {
        cornetto_accel_t *h = cli_accel_open();
        const uint16_t **pd = (const uint16_t **)cli_xmalloc((size_t)n_ctg * sizeof(*pd));
        const uint16_t **pq = (const uint16_t **)cli_xmalloc((size_t)n_ctg * sizeof(*pq));
        int32_t *lens = (int32_t *)cli_xmalloc((size_t)n_ctg * sizeof(int32_t));
        for (int32_t i = 0; i < n_ctg; ++i) {
            pd[i] = ctg[i].depth;
            pq[i] = ctg[i].mq;
            lens[i] = ctg[i].len;
        }
        cornetto_cov_t *cov = NULL;
        cli_accel_check(h, cornetto_cov_upload(h, pd, pq, lens, n_ctg, &cov), "copying depth arrays to the GPU");
        t0 = cli_realtime();
        uint64_t sums[3];
        cli_accel_check(h, cornetto_cov_prepare(h, cov, opt.window_size, opt.window_inc, sums), "window block sums");
        /* double accumulators of the reference hold these integers exactly (:283-285) */
        mean_depth = (int32_t)round((double)sums[0] / (double)sums[2]);   /* :293 */
        mean_mq = (int32_t)round((double)sums[1] / (double)sums[2]);      /* :294 */
        const int32_t lo_t = cornetto_cov_threshold(opt.low_cov_thresh, mean_depth);   /* :518 */
        const int32_t hi_t = cornetto_cov_threshold(opt.high_cov_thresh, mean_depth);  /* :519 */
        cli_accel_check(h, cornetto_cov_select(h, cov, lo_t, hi_t, opt.low_mq_cov_thresh, opt.edge_len, opt.min_ctg_len, boring, &recs, &n_recs),
                        "window classification");
        CLI_VERBOSE("Found regions in %.2f seconds", cli_realtime() - t0);
        cornetto_cov_free(h, cov);
        cornetto_accel_close(h);
        free(pd);
        free(pq);
        free(lens);
    } else {
        /* the reference divides 0/0 here: round(NaN) -> INT_MIN; nothing is printed either way */
        mean_depth = mean_mq = (int32_t)0x80000000;
    }
    fprintf(stderr, "Number of contigs: %d\n", n_ctg); /* :497-506 */
    fprintf(stderr, "Average depth: %d\n", mean_depth);
    fprintf(stderr, "Average mq depth: %d\n", mean_mq);
    fprintf(stderr, "Window size: %d\n", opt.window_size);
    fprintf(stderr, "Window increment: %d\n", opt.window_inc);
    fprintf(stderr, "Low coverage threshold: %.1fx%d\n", opt.low_cov_thresh, mean_depth);
    fprintf(stderr, "High coverage threshold: %.1fx%d\n", opt.high_cov_thresh, mean_depth);
    fprintf(stderr, "Low mapq coverage threshold: %.1f\n", opt.low_mq_cov_thresh);
    fprintf(stderr, "Min contig length: %d\n", opt.min_ctg_len);
    fprintf(stderr, "Edge length: %d\n", opt.edge_len);

    /* ---------------- print (:425-445 / :463-481) ---------------- */
    t0 = cli_realtime();
    int64_t k = 0;
    for (int32_t i = 0; i < n_ctg; ++i) {
        const char *name = ctg[i].name;
        const int len = ctg[i].len;
        if (!boring) {
            if (len < opt.min_ctg_len) {
                printf("%s\t%d\t%d\t.\t.\n", name, 0, opt.min_ctg_len); /* :430 prints min_ctg_len, not len */
            } else {
                printf("%s\t%d\t%d\t.\t.\n", name, 0, opt.edge_len);
                printf("%s\t%d\t%d\t.\t.\n", name, len - opt.edge_len, len);
            }
        }
        for (; k < n_recs && recs[k].ctg == i; ++k)
            printf("%s\t%d\t%d\t%d\t%d\n", name, recs[k].st, recs[k].end, recs[k].depth, recs[k].mq_depth);
    }
    CLI_VERBOSE("Printed the bits in %.2f seconds", cli_realtime() - t0);
    cornetto_free(recs);
    for (int32_t i = 0; i < n_ctg; ++i) {
        free(ctg[i].name);
        free(ctg[i].depth);
        free(ctg[i].mq);
    }
    free(ctg);
    return 0;
}

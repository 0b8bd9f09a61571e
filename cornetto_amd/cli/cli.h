/* cli.h — shared declarations of the `cornetto` host CLI (C99) on top of the C ABI in
 * include/cornetto_accel.h.  Drop-in for the reference's CLI on the panel-creation path: same
 * sub-command names, option letters, stdout bytes and exit codes (reference: src/main.c:95-152). */
#ifndef CORNETTO_CLI_H
#define CORNETTO_CLI_H

#include <stdint.h>
#include <stdio.h>

#include "cornetto_accel.h"

#define CORNETTO_VERSION "0.2.0" /* src/cornetto.h:40 */

/* ---- logging (format of src/error.h:54-103; stderr text is not part of parity, exit codes are) ---- */
extern int cli_log_level; /* default 4 = LOG_VERB, src/error.c:33 */
#define CLI_ERROR(...)                                                                                       \
    do {                                                                                                     \
        if (cli_log_level >= 1) {                                                                            \
            fprintf(stderr, "[%s::ERROR]\033[1;31m ", __func__);                                             \
            fprintf(stderr, __VA_ARGS__);                                                                    \
            fprintf(stderr, "\033[0m\n At %s:%d\n", __FILE__, __LINE__ - 1);                                 \
        }                                                                                                    \
    } while (0)
#define CLI_WARNING(...)                                                                                     \
    do {                                                                                                     \
        if (cli_log_level >= 2) {                                                                            \
            fprintf(stderr, "[%s::WARNING]\033[1;33m ", __func__);                                           \
            fprintf(stderr, __VA_ARGS__);                                                                    \
            fprintf(stderr, "\033[0m\n At %s:%d\n", __FILE__, __LINE__ - 1);                                 \
        }                                                                                                    \
    } while (0)
#define CLI_VERBOSE(...)                                                                                     \
    do {                                                                                                     \
        if (cli_log_level >= 4) {                                                                            \
            fprintf(stderr, "[INFO] %s: ", __func__);                                                        \
            fprintf(stderr, __VA_ARGS__);                                                                    \
            fprintf(stderr, "\n");                                                                           \
        }                                                                                                    \
    } while (0)

double cli_realtime(void);
double cli_cputime(void);
long cli_peakrss(void);
void *cli_xmalloc(size_t n);
/* order[k] = index of the k-th longest record, ties in input order (one of the two length arrays is NULL) */
void cli_order_by_length_desc(const int64_t *lens64, const int32_t *lens32, int32_t n, int32_t *order);
void *cli_xrealloc(void *p, size_t n);
char *cli_xstrdup(const char *s);

/* Open the accelerator ($CORNETTO_DEVICE, default 0) or print the reason and exit(EXIT_FAILURE): the
 * host path (host_backend.c) is only ever chosen explicitly. */
cornetto_accel_t *cli_accel_open(void);
/* the same, with the HIP initialisation running on a helper thread between _begin() and _end() (while the input is read);
 * _cancel() drops it when no device work turned up */
/* `want` bytes at file offset `off` with up to n_threads pread() threads -> bytes read (short only at the end of the file), -1 on a read error */
int64_t cli_read_at(int fd, char *dst, int64_t want, int64_t off, int *failed);   /* pread(), or a copy out of a kept mapping for a file on tmpfs */
int64_t cli_pread_parallel(int fd, char *dst, int64_t want, int64_t off, int n_threads);
void cli_accel_open_begin(void);
void cli_accel_warm_hint(int what);   /* before cli_accel_open_begin(): CORNETTO_WARM_* to run behind the open, on a handle of its own */
cornetto_accel_t *cli_accel_open_end(void);
void cli_accel_open_cancel(void);
/* print the handle's last error and exit(EXIT_FAILURE) if rc != 0 */
void cli_ref_abort(const char *msg);   /* stderr line + abort(): where the reference dies on an assert */
void cli_accel_check(cornetto_accel_t *h, int rc, const char *what);

/* ---- record lines on stdout: a plain buffer + decimal formatter (printf costs ~200 ns per field; the window lists have
 * millions of lines).  Goes through stdout's FILE, so it may be mixed with printf as long as cli_out_flush() comes first. */
extern char *cli_out_buf;
extern size_t cli_out_n;
#define CLI_OUT_CAP (1u << 22)
void cli_out_flush(void);
static inline void cli_out_room(size_t need)
{
    if (cli_out_n + need > CLI_OUT_CAP) cli_out_flush();
}
static inline void cli_out_bytes(const char *s, size_t n)
{
    if (n > CLI_OUT_CAP / 2) {
        cli_out_flush();
        fwrite(s, 1, n, stdout);
        return;
    }
    cli_out_room(n);
    for (size_t i = 0; i < n; ++i) cli_out_buf[cli_out_n + i] = s[i];
    cli_out_n += n;
}
static inline void cli_out_char(char c)
{
    cli_out_room(1);
    cli_out_buf[cli_out_n++] = c;
}
static inline void cli_out_int(long long v) /* %lld */
{
    char t[24];
    int k = 0;
    unsigned long long u = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
    do {
        t[k++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    cli_out_room(24);
    if (v < 0) cli_out_buf[cli_out_n++] = '-';
    while (k) cli_out_buf[cli_out_n++] = t[--k];
}

/* ---- FASTA/FASTQ(+gz) records with the framing rules of klib kseq (src/kseq.h:184-224) ---- */
typedef struct {
    char *s;
    size_t l, m;
} cli_str_t;

typedef struct cli_fastx cli_fastx_t;
extern int cli_dash_is_stdin;                  /* set by sdust_main: "-" = stdin (the other FASTA sub-commands open a file of that name, as the reference does) */
cli_fastx_t *cli_fastx_open(const char *path); /* NULL on failure */
/* a reader over `n` bytes at `prefix` (borrowed: must outlive the reads of them) followed by the rest of the open
 * gzFile `gz` (owned: closed by cli_fastx_close) */
cli_fastx_t *cli_fastx_open_prefixed(void *gz, const void *prefix, size_t n);
void cli_fastx_close(cli_fastx_t *f);
/* >= 0: sequence length; -1 end of file; -2 truncated quality string */
int64_t cli_fastx_read(cli_fastx_t *f, cli_str_t *name, cli_str_t *comment, cli_str_t *seq, cli_str_t *qual);

/* $CORNETTO_DEVICES: ordinals of the GPUs to spread the contigs over (at most CLI_MAX_DEV); returns how many, 0 if unset */
#define CLI_MAX_DEV 64
int cli_device_list(int *devs);

/* a batch of records held in memory for one device pass */
typedef struct {
    char **names;
    uint8_t **seqs;
    int64_t *lens;
    int32_t n, cap;
    int64_t bases;
} cli_batch_t;
void cli_batch_push(cli_batch_t *b, const char *name, const char *seq, int64_t len);
void cli_batch_take(cli_batch_t *b, const char *name, cli_str_t *seq); /* takes seq's buffer over when it is long */
void cli_batch_clear(cli_batch_t *b);
int64_t cli_batch_limit(void); /* $CORNETTO_BATCH_BASES, default 4e9 */

/* ---- the host path (cli/host_backend.c): plain sequential C99 for every scan, chosen with --accel=no (noboringbits / boringbits, the
 * reference's own switch: src/boringbits_main.c:627-632) or CORNETTO_ACCEL=no (every sub-command) — never by itself ---- */
int cli_host_mode(void);
void cli_host_set(int on);
/* the hits / intervals of record `ctg` are appended to a growing array (n, cap in records) */
void cli_host_telofind(const uint8_t *seq, int64_t len, const char *motif, int32_t ctg, cornetto_hit_t **hits, int64_t *n_hits, int64_t *cap_hits);
int cli_host_sdust(const uint8_t *seq, int64_t len, int T, int W, int32_t ctg, cornetto_ivl_t **ivls, int64_t *n_ivls, int64_t *cap_ivls); /* -1: -w / -t out of range */
/* same contract as cornetto_telowin(); results are malloc memory */
void cli_host_telowin(const cornetto_hit_t *hits, int64_t n_hits, const int32_t *lens, int32_t n_ctg, double thr_adj, cornetto_win_t **wins, int64_t *n_wins);
/* get_depths(): what the two bedgraphs hold; exits with the reference's messages on malformed input */
typedef struct {
    int32_t n_ctg;
    char **names;
    int32_t *lens;
    int64_t *first;          /* [n_ctg + 1]: contig i is depth[first[i] .. first[i + 1]) */
    uint16_t *depth, *mq;
    int64_t n_pos, n_clamped;
    double sum_depth, sum_mq, positions; /* the reference's double accumulators (:283-285) */
} cli_host_cov_t;
void cli_host_get_depths(FILE *ft, FILE *fq, cli_host_cov_t *out);
void cli_host_cov_free(cli_host_cov_t *c);
/* get_regs() + the predicate of print_fun_bits (boring = 0) / print_boring_bits (1): the rows to print, malloc memory */
void cli_host_cov_select(const cli_host_cov_t *c, int w, int inc, int32_t lo, int32_t hi, float low_mq, int32_t edge_len, int32_t min_ctg_len, int boring,
                         cornetto_regrec_t **recs, int64_t *n_recs);
void cli_host_merge_windows(const cornetto_regrec_t *recs, int64_t n_recs, int32_t dist, int32_t min_len, cornetto_ivl_t **ivls, int64_t *n_ivls);
/* same contract as cornetto_telobreaks(); -1: a coordinate outside its contig */
int cli_host_telobreaks(const int32_t *ctg_len, int32_t n_ctg, const cornetto_ivl_t *sd, int64_t n_sd, const cornetto_telrow_t *tel, int64_t n_tel,
                        cornetto_ivl_t **out, int64_t *n_out);

/* ---- sub-commands: int xxx_main(int argc, char *argv[]) with argv[0] = sub-command (src/main.c:40-54) ---- */
int depth_main(int argc, char *argv[]);
int boringbits_main(int argc, char *argv[], int8_t boring);
int bigenough_main(int argc, char *argv[]);
int find_telomere_main(int argc, char *argv[]);
int telomere_windows_main(int argc, char *argv[]);
int telomere_breaks_main(int argc, char *argv[]);
int sdust_main(int argc, char *argv[]);
int assbed_main(int argc, char *argv[]);
int seq_main(int argc, char *argv[]);

#endif

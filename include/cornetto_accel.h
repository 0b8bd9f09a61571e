/*
 * cornetto_accel.h — C ABI of libcornetto_hip.so: the MI355X (gfx950) implementation of cornetto's
 * panel-creation hot path.  Plain C, plain pointers and sizes; HIP lives behind it.
 *
 * The reference (hasindu2008/cornetto v0.2.0, paths relative to its root) exposes no FFI: its boundary is
 * the process CLI `cornetto <sub> ...` -> `int xxx_main(int argc, char *argv[])` (src/main.c:40-54,103-134)
 * and, underneath, per-contig C functions.  Each entry point below names the reference unit it replaces.
 * The host CLI in cornetto_amd/cli/ (C99) keeps the sub-command names, option letters, stdout bytes and
 * exit codes and calls only what is declared here.  There is NO CPU fallback behind these symbols: every
 * compute entry point runs HIP kernels and fails with CORNETTO_E_NODEVICE when no GPU is usable.
 *
 * Conventions
 *   - every function returns 0 (CORNETTO_OK) or a negative CORNETTO_E_* status; nothing calls exit()
 *     (the reference's ERROR()+exit(EXIT_FAILURE), src/error.h:97-103, is done by the CLI mains);
 *   - result arrays are allocated by the library (malloc, or pinned host memory from a pool for large
 *     results) and MUST be released with cornetto_free(), never free(); the one exception is
 *     cornetto_sdust(), which keeps the reference's contract (caller free()s);
 *   - calls are synchronous; one cornetto_accel_t may be used by one host thread at a time, and so may one
 *     cornetto_asm_t / cornetto_cov_t: a resident object caches its decomposition tables and the result counts of
 *     its last scan (what the next scan's buffers are sized by), so scans of the same object from two handles must
 *     not run at the same moment (two handles over two objects that wrap the same device memory may);
 *   - coordinates are 0-based, half-open, per contig, 32-bit (kseq_read returns int: src/kseq.h:185).
 */
#ifndef CORNETTO_ACCEL_H
#define CORNETTO_ACCEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CORNETTO_ACCEL_ABI 1

enum {
    CORNETTO_OK = 0,
    CORNETTO_E_NODEVICE = -1, /* no usable HIP device / HIP runtime error at open */
    CORNETTO_E_HIP = -2,      /* a HIP call or kernel failed; see cornetto_accel_last_error() */
    CORNETTO_E_ARG = -3,      /* invalid argument (NULL, negative length, misaligned device offset, ...) */
    CORNETTO_E_NOMEM = -4,    /* host or device allocation failed */
    CORNETTO_E_UNSUPPORTED = -5, /* parameter outside the implemented range (sdust W > 1026, T > 2^20, ...) */
    CORNETTO_E_FORMAT = -6,      /* malformed input text; see cornetto_bgin_error() */
    CORNETTO_E_ASSERT = -7       /* the reference ends with SIGABRT on these inputs: an assert of get_regs() fails (src/boringbits_main.c:353,368);
                                    cornetto_accel_last_error() names it.  A drop-in caller prints that and abort()s */
};

typedef struct cornetto_accel cornetto_accel_t; /* device + stream + workspaces */
typedef struct cornetto_asm cornetto_asm_t;     /* a set of contigs (or reads) resident in HBM, 1 B/base */
typedef struct cornetto_cov cornetto_cov_t;     /* per-base depth + mq-depth (uint16) of a set of contigs in HBM */

/* one telofind run; replaces one output line of find(): src/find_telomere.c:51,56 */
typedef struct {
    int32_t ctg;    /* index of the contig in the cornetto_asm_t */
    int32_t strand; /* 0 = motif, 1 = reverse complement of the motif */
    int32_t start, end;
} cornetto_hit_t;

/* one telowin window that met the threshold; replaces one printf of process_scaffold():
 * src/telomere_windows.c:37-38.  ratio printed by the caller as (double)car/(end-start), "%.3g". */
typedef struct {
    int32_t ctg;
    int32_t start, end; /* end = start + den */
    int32_t car;        /* marked bases in the window */
} cornetto_win_t;

/* one masked interval of sdust for a set of contigs */
typedef struct {
    int32_t ctg;
    int32_t start, finish;
} cornetto_ivl_t;

/* a coverage window; = reg_t of src/boringbits_main.c:133-138 */
typedef struct {
    int32_t st, end, depth, mq_depth;
} cornetto_reg_t;

/* a selected coverage window of a set of contigs (one stdout line of print_fun_bits / print_boring_bits) */
typedef struct {
    int32_t ctg;
    int32_t st, end, depth, mq_depth;
} cornetto_regrec_t;

/* ---------------------------------------------------------------------------------------------------
 * runtime
 * ------------------------------------------------------------------------------------------------- */

/* number of visible HIP devices (0 when there is none or the runtime is unusable) */
int cornetto_accel_device_count(void);

/* Open device `device` (ordinal among the visible devices).  `stream` is a hipStream_t to launch on, or
 * NULL for a stream owned by the handle.  The seam the reference left for this is --accel=yes|no /
 * CORNETTO_ACC: src/cornetto.h:47, src/boringbits_main.c:627-632. */
int cornetto_accel_open(cornetto_accel_t **h, int device, void *stream);
void cornetto_accel_close(cornetto_accel_t *h);

/* message of the last failure on this handle ("" if none); valid until the next call on the handle */
const char *cornetto_accel_last_error(const cornetto_accel_t *h);
const char *cornetto_accel_strerror(int status);

/* release a result array returned by this library */
void cornetto_free(void *p);

/* Device time of the kernels of the most recent compute call on this handle, measured with HIP events on
 * the handle's stream.  Fills up to `cap` entries of names[]/ms[] (names are static strings) and returns
 * the number of kernels recorded.  Used by bench.py for the live roofline figure. */
int cornetto_accel_last_timing(const cornetto_accel_t *h, const char **names, float *ms, int cap);

/* Share of every compute unit (wave slots and LDS), in percent (10..100, default 100), that the long-running
 * sdust kernel of this handle may occupy.  Lower it when another handle / stream computes on the same device at
 * the same time (bench.py runs telofind and the coverage windows beside sdust): the sdust waves stay resident
 * until their work queue is empty, so what they do not leave free is not available to anybody else meanwhile.
 * Scheduling only: results do not depend on it. */
int cornetto_accel_set_share(cornetto_accel_t *h, int percent);

/* Every entry point of the named groups once on a 4 kb built-in input: what the first use of the runtime costs in a process (loading the code objects,
 * setting the copy engines up, the first pinned pools: 13 of the 25 ms by which the first pass over an assembly exceeded the second) is paid here instead
 * of in the first scan.  Optional; a caller that reads its input first runs it beside that, on the thread that opened the handle.  Nothing the reference
 * has a counterpart for: its per-contig functions start at once (src/sdust/sdust.c:196-203). */
#define CORNETTO_WARM_SDUST 1
#define CORNETTO_WARM_TELO 2
#define CORNETTO_WARM_COV 4
#define CORNETTO_WARM_ALL 7
int cornetto_accel_warm(cornetto_accel_t *h, int what);

/* "The rest of the device is free now" (on != 0) / "is in use again" (on == 0): may be called from ANOTHER host thread than the one that is
 * inside cornetto_sdust_asm() on this handle.  While it is on, a running sdust call whose share is below 100 launches the waves it had left
 * to the other stream as a second kernel that draws from the same chunk counters, and later calls start with the whole chip.  bench.py
 * switches it on when the telofind + coverage thread of a step is through and off when the next step begins.  Scheduling only. */
int cornetto_accel_boost(cornetto_accel_t *h, int on);

/* How many times a cornetto_sdust_asm() call on this handle has put its resident waves on the device so far (readable from any thread while the
 * call runs).  Where the resident waves land decides how much room the other handle's kernels find on every CU for the rest of the call: a caller
 * that starts both sides of a step at once should let the count move before it launches the other side's first kernel — measured on the 3.16 Gbp
 * bench step: 8.2 ms when the sdust waves are there first, 9.2 ms when they arrive 30 us behind the other stream's first kernel. */
unsigned long long cornetto_accel_launch_count(const cornetto_accel_t *h);

/* Lazy result copies (off by default).  While on, cornetto_telo_scan() / cornetto_telofind() and cornetto_cov_select*() return as soon as their
 * kernels are through; the device-to-host copy of the large result array (telomere runs; selected windows) is still in flight on a stream of its
 * own, beside whatever the caller launches next on this handle.  The returned pointers are valid at once, their CONTENT after
 * cornetto_accel_wait().  Wait before reading, before cornetto_free() of such an array and before calling the same entry point again on the
 * handle (its device staging is reused).  Counts and every small output are final when the call returns. */
int cornetto_accel_set_lazy(cornetto_accel_t *h, int on);
int cornetto_accel_wait(cornetto_accel_t *h);

/* Statistics of the production sdust kernel (development / bench aid).  `enable` != 0 makes the following
 * cornetto_sdust_asm() calls on this handle run the counting build of the kernel (a few percent slower); `out`, if not
 * NULL, receives up to `cap` (<= 256) counters of the most recent such call: [2] wave steps, [3] find_perfect calls,
 * [4] groups of 4 steps that took the ungated path, [5] sum and [6] maximum over the waves of their run time in 10 ns ticks, [7] chunks sampled as low-complexity,
 * [10] find_perfect calls with candidates, [11] queue fetch rounds, [12] ticks spent fetching, [16 + 4b ...] per 0.5 ms bin b
 * of wave run time: waves, find_perfect calls with candidates, jobs, find_perfect calls; [254] waves launched, [255] chunks.
 * The sift / resolve kernel (the default for W <= 66) fills instead: [200] resolve steps inside runs of consecutive positions, [201] resolve steps
 * behind a gap (window reads), [202] passes with candidates, [203] positions with more than T / 10 equal words in front, [204] of those after the
 * partial sums over the shortest candidate, [205] after the 16-term sums / the exact walk, [206] tiles of 64 bases, [207] base-by-base steps of
 * chunks with other bytes.
 * Returns the number of counters copied.  Results of sdust do not depend on it. */
int cornetto_accel_sdust_stats(cornetto_accel_t *h, int enable, uint64_t *out, int cap);

/* Which kernel launches are bracketed by HIP events for cornetto_accel_last_timing(): 2 (default) every launch, 1 only
 * the three main kernels (sdust_kernel, cov_blocks, tf_scan), 0 none.  Every event is a packet of its own on the queue;
 * beside a busy second stream the dozen small launches of a call cost about a millisecond more with level 2. */
int cornetto_accel_set_timing(cornetto_accel_t *h, int level);

/* ---------------------------------------------------------------------------------------------------
 * sequences in HBM
 * ------------------------------------------------------------------------------------------------- */

/* Copy n sequences (ASCII, any case, IUPAC allowed, need not be NUL-terminated) to the device.  Replaces
 * the kseq_t.seq buffers the reference scans in place (src/find_telomere.c:101-105, src/sdust/sdust.c:196-199). */
int cornetto_asm_upload(cornetto_accel_t *h, const uint8_t *const *seqs, const int64_t *lens, int32_t n,
                        cornetto_asm_t **out);

/* Wrap bases that already are in device memory: contig i occupies d_bases[offsets[i] .. offsets[i]+lens[i]).
 * Every offset must be a multiple of 64 and the buffer must stay readable for 128 bytes past the last
 * contig (padding content is ignored).  The buffer is borrowed, never freed.  Its CONTENT must not change while the object
 * exists: work decompositions derived from it (chunk / tile tables; with CORNETTO_SDUST_SIFT=0 also the order in which the
 * chunks are handed out, which depends on a sample of the bases) are kept with the object by the first call that needs them.
 * Results stay exact if it does change (the order is a scheduling hint only), the balance of the work does not: wrap again. */
int cornetto_asm_wrap(cornetto_accel_t *h, const void *d_bases, const int64_t *offsets, const int64_t *lens,
                      int32_t n, cornetto_asm_t **out);
void cornetto_asm_free(cornetto_accel_t *h, cornetto_asm_t *a);

/* ---------------------------------------------------------------------------------------------------
 * telofind / telowin
 * ------------------------------------------------------------------------------------------------- */

/* telofind over every contig of `a`: replaces disambiguate()+find() (src/find_telomere.c:76-81,44-74)
 * called per record at :103-104.  motif: any length from 1 byte (up to 32 bytes: the fast path; longer ones are compared
 * byte by byte), compared as the reference does (sequence
 * upper-cased, motif not).  hits come out in the reference's print order: by contig, all strand-0 runs by
 * position, then all strand-1 runs. */
int cornetto_telofind(cornetto_accel_t *h, const cornetto_asm_t *a, const char *motif,
                      cornetto_hit_t **hits, int64_t *n_hits);

/* adjusted threshold of telowin: threshold * pow(identity_percent/100, 6)  (src/telomere_windows.c:53-54) */
double cornetto_telowin_threshold(double threshold, double identity_percent);

/* telowin from explicit hits (what `cornetto telowin in.telomere ...` parses from the TSV): for each of the
 * n_ctg contigs, mark [start,end) of its hits and scan 1000-bp windows, step 200.  Replaces the marking
 * loop + process_scaffold() (src/telomere_windows.c:75-79,28-43).  hits[].ctg indexes ctg_len[]; hits with
 * the same ctg need not be adjacent.  Windows are returned by contig, then by start. */
int cornetto_telowin(cornetto_accel_t *h, const cornetto_hit_t *hits, int64_t n_hits, const int32_t *ctg_len,
                     int32_t n_ctg, double thr_adj, cornetto_win_t **wins, int64_t *n_wins);

/* fused telofind -> telowin on the device (no TSV round trip, the mark bitmap never leaves HBM).
 * Equivalent to cornetto_telofind() followed by cornetto_telowin() on its hits.  hits/n_hits may be NULL. */
int cornetto_telo_scan(cornetto_accel_t *h, const cornetto_asm_t *a, const char *motif, double thr_adj,
                       cornetto_hit_t **hits, int64_t *n_hits, cornetto_win_t **wins, int64_t *n_wins);

/* ---------------------------------------------------------------------------------------------------
 * sdust
 * ------------------------------------------------------------------------------------------------- */

/* symmetric DUST over every contig of `a`; replaces sdust() called per record at src/sdust/sdust.c:199
 * (sdust_core :130-160).  3 <= W <= 1026 (CORNETTO_E_UNSUPPORTED beyond: there the reference's own int products overflow), T >= 0.  Intervals by contig, then by start; per contig they are
 * exactly the reference's (start<<32|finish) list, including intervals that run past the contig end
 * after an N run. */
int cornetto_sdust_asm(cornetto_accel_t *h, const cornetto_asm_t *a, int32_t T, int32_t W,
                       cornetto_ivl_t **ivls, int64_t *n_ivls);

/* The same call in two parts, for a host that has other work for the device meanwhile (a second stream through a second handle) and no thread
 * to spare: _begin queues the call on the handle's stream and returns without waiting WHEN the rest of the call can be sized by the counts the
 * last call over the same assembly, T and W left behind (the scan, the kernel, the stitch and the result copy in one go; the counts are checked
 * by _end, and a call whose counts outgrew them is run again the long way: never a truncated answer); otherwise — the first call for an
 * assembly, the statistics build — it queues nothing (cornetto_accel_launch_count() tells) and _end runs the whole call.  Between the two the
 * handle must not be used for anything else; the same (a, T, W) go to both.  Results, order and errors are those of cornetto_sdust_asm(). */
int cornetto_sdust_asm_begin(cornetto_accel_t *h, const cornetto_asm_t *a, int32_t T, int32_t W);
int cornetto_sdust_asm_end(cornetto_accel_t *h, const cornetto_asm_t *a, int32_t T, int32_t W,
                           cornetto_ivl_t **ivls, int64_t *n_ivls);

/* Drop-in for `uint64_t *sdust(void *km, const uint8_t *seq, int l_seq, int T, int W, int *n)`
 * (src/sdust/sdust.h:19): same arguments, same ownership (caller free()s), l_seq < 0 means strlen.
 * km must be NULL.  Uses a process-wide handle on device $CORNETTO_DEVICE (default 0); returns NULL and
 * sets *n = -1 when the device path is unavailable. */
uint64_t *cornetto_sdust(void *km, const uint8_t *seq, int l_seq, int T, int W, int *n);

/* Drop-ins for the buffered interface of src/sdust/sdust.h:16-21 (`sdust_buf_init`, `sdust_buf_destroy`, `sdust_core`):
 * the result array belongs to the buf, is valid until the next cornetto_sdust_core() on that buf or its destruction, and
 * must not be freed by the caller.  km must be NULL.  Same process-wide handle as cornetto_sdust(); NULL and *n = -1
 * when the device path is unavailable. */
typedef struct cornetto_sdust_buf cornetto_sdust_buf_t;
cornetto_sdust_buf_t *cornetto_sdust_buf_init(void *km);
void cornetto_sdust_buf_destroy(cornetto_sdust_buf_t *buf);
const uint64_t *cornetto_sdust_core(const uint8_t *seq, int l_seq, int T, int W, int *n, cornetto_sdust_buf_t *buf);

/* ---------------------------------------------------------------------------------------------------
 * (no)boringbits window stage
 * ------------------------------------------------------------------------------------------------- */

/* Copy per-base depth arrays of n contigs to the device; replaces ctg_depth_t.depth/.mq_depth filled by
 * get_depths() (src/boringbits_main.c:116-122,279-281). */
int cornetto_cov_upload(cornetto_accel_t *h, const uint16_t *const *depth, const uint16_t *const *mq_depth,
                        const int32_t *lens, int32_t n, cornetto_cov_t **out);

/* Wrap arrays already in device memory; offsets are in ELEMENTS and must be multiples of 64; both arrays
 * must stay readable for 64 elements past the last contig. */
int cornetto_cov_wrap(cornetto_accel_t *h, const void *d_depth, const void *d_mq_depth, const int64_t *offsets,
                      const int32_t *lens, int32_t n, cornetto_cov_t **out);
void cornetto_cov_free(cornetto_accel_t *h, cornetto_cov_t *c);

/* Several GPUs: copy contigs ctgs[0..n) (indices into `src`, any order) of a coverage object that is resident on the device of
 * h_src into a new object on the device of h_dst (the same device is allowed).  get_regs() is independent per contig
 * (src/boringbits_main.c:331); the one quantity that spans contigs is the assembly-wide mean behind the thresholds
 * (:283-294 -> :518-519): every device runs cornetto_cov_prepare() on its share, the caller adds the sums[] up. */
int cornetto_cov_shard(cornetto_accel_t *h_src, const cornetto_cov_t *src, cornetto_accel_t *h_dst, const int32_t *ctgs, int32_t n,
                       cornetto_cov_t **out);

/* number of contigs and their lengths (owned by the object) */
int32_t cornetto_cov_n(const cornetto_cov_t *c);
const int32_t *cornetto_cov_lens(const cornetto_cov_t *c);

/* number of windows of a contig: src/boringbits_main.c:338-339 */
int32_t cornetto_n_reg(int32_t length, int32_t window_size, int32_t window_inc);

/* The asserts of get_regs() for one contig (src/boringbits_main.c:353 `st<end`, :368 `end == length`): 0 when the reference computes
 * the windows, else the line of the assert that aborts it.  With 1 <= window_inc <= window_size and length >= 1 always 0; with
 * window_inc > window_size the windows are sparse ([j*inc, min(j*inc + w, length))) and the last one has to reach the contig's end. */
int32_t cornetto_regs_assert(int32_t length, int32_t window_size, int32_t window_inc);

/* Stage 1: per-`window_inc` block sums on the device plus the exact totals the mean needs.
 * sums[0] = sum of depth, sums[1] = sum of mq_depth, sums[2] = number of positions; the caller forms
 * mean = (int)round(sums[0]/sums[2]) (src/boringbits_main.c:283-285,293-294) — across ranks after an
 * all-reduce of sums[].  For a coverage read from bedgraph text that holds NEGATIVE depth values (the reference takes them: %d) the
 * arrays hold their uint16 (:282-283) and the totals the values themselves (:285-286): sums[0] and sums[1] are then two's complement
 * numbers — form the mean from (int64_t)sums[.].  Needs window_inc >= 1; window_inc > window_size is computed as the reference computes it (:346-366).
 * CORNETTO_E_ASSERT when cornetto_regs_assert() is non-zero for any contig: get_regs() runs over EVERY contig before the
 * reference prints anything, so the process ends there with nothing on stdout. */
int cornetto_cov_prepare(cornetto_accel_t *h, cornetto_cov_t *c, int32_t window_size, int32_t window_inc,
                         uint64_t sums[3]);

/* get_regs() for one contig (src/boringbits_main.c:346-366): all cornetto_n_reg() windows, in order, into
 * caller-provided regs[].  cornetto_cov_prepare() must have been called with the same sizes. */
int cornetto_cov_regs(cornetto_accel_t *h, const cornetto_cov_t *c, int32_t ctg, cornetto_reg_t *regs);

/* thresholds: (int)round(factor * mean) with the product formed in float (src/boringbits_main.c:518-519) */
int32_t cornetto_cov_threshold(float factor, int32_t mean);

/* Stage 2: classify every window and return the selected ones in print order (contig, then st).
 *   boring == 0: print_fun_bits  (src/boringbits_main.c:425-445) — windows of contigs with
 *                len >= min_ctg_len that satisfy depth<lo || depth>hi || mq/(double)depth < low_mq;
 *                (the caller prints the '.' edge / short-contig lines, which need no data);
 *   boring != 0: print_boring_bits (:463-481) — contigs with len > min_ctg_len, windows with
 *                st > edge_len && end < len - edge_len that do NOT satisfy the predicate. */
int cornetto_cov_select(cornetto_accel_t *h, const cornetto_cov_t *c, int32_t lo, int32_t hi, float low_mq,
                        int32_t edge_len, int32_t min_ctg_len, int boring, cornetto_regrec_t **recs,
                        int64_t *n_recs);

/* The same selection in packed form, for callers that take millions of windows per call (the whole-assembly panel: 7.5 M
 * windows of a 3 Gbp assembly are 150 MB as cornetto_regrec_t, 60 MB packed): 8 bytes per window, the contig given by
 * position — the windows of contig i are recs[ctg_first[i] .. ctg_first[i + 1]) (ctg_first has n + 1 entries) — and
 * end = min(st + window_size, length of the contig) (src/boringbits_main.c:348-351).  depth and mq_depth are means of uint16
 * values (:354-361), so they fit.  Release recs and ctg_first with cornetto_free(). */
typedef struct {
    int32_t st;
    uint16_t depth, mq_depth;
} cornetto_regpk_t;
int cornetto_cov_select_packed(cornetto_accel_t *h, const cornetto_cov_t *c, int32_t lo, int32_t hi, float low_mq,
                               int32_t edge_len, int32_t min_ctg_len, int boring, cornetto_regpk_t **recs, int64_t *n_recs,
                               int64_t **ctg_first);

/* One pass of the panel path over a resident assembly and its resident coverage in one call: cornetto_cov_prepare() ->
 * cornetto_cov_threshold() x 2 (mean = (int)round(sums[0] / sums[2]), src/boringbits_main.c:293, :518-519) ->
 * cornetto_cov_select_packed() -> cornetto_telo_scan(), the same kernels and the same results.  What it saves is host round trips:
 * everything behind the totals is queued in one go, sized by the counts the previous step over the same objects gave and checked
 * afterwards (two synchronisations instead of five; a count that outgrew its estimate repeats that part through the exact entry
 * point).  `exchange`, if not NULL, is called once with the three sums of THIS process's contigs and replaces them by the sums
 * over all processes (one rank per GPU: a 3 x int64 all-reduce — the one exchange the path has, :283-294 -> :518-519); it returns 0
 * or an error.  sums[] and thr[] (low, high) are returned; result arrays as from the single entry points (cornetto_free / free).
 * With lazy copies (cornetto_accel_set_lazy) the contents of recs / hits are complete after cornetto_accel_wait().
 * window_size above 32768 is CORNETTO_E_UNSUPPORTED here as in cornetto_cov_select_packed() (the packed record's 16-bit means). */
typedef struct {
    const char *motif;              /* telofind motif (src/find_telomere.c:101-105) */
    double thr_adj;                 /* cornetto_telowin_threshold() */
    int32_t window_size, window_inc;
    float low_cov, high_cov, low_mq;   /* -L -H -Q */
    int32_t edge_len, min_ctg_len;     /* -e -m */
    int32_t boring;                    /* 0: noboringbits' selection, 1: the deprecated boringbits' */
} cornetto_step_opt_t;
typedef int (*cornetto_sums_fn)(uint64_t sums[3], void *ctx);
int cornetto_panel_step(cornetto_accel_t *h, const cornetto_asm_t *a, const cornetto_cov_t *c, const cornetto_step_opt_t *opt, cornetto_sums_fn exchange, void *ctx,
                        uint64_t sums[3], int32_t thr[2], cornetto_regpk_t **recs, int64_t *n_recs, int64_t **ctg_first, cornetto_hit_t **hits, int64_t *n_hits,
                        cornetto_win_t **wins, int64_t *n_wins);

/* ---------------------------------------------------------------------------------------------------
 * bedgraph ingest on the device (the text parse of get_depths(), src/boringbits_main.c:204-287)
 * ------------------------------------------------------------------------------------------------- */

typedef struct cornetto_bgin cornetto_bgin_t; /* streaming state of one pair of per-base bedgraphs */

/* which of the reference's checks failed first (smallest record index), with the numbers its message prints */
typedef struct {
    int32_t kind;   /* 1: cov-total record has != 4 columns (a = converted fields)   :209-212
                       2: cov-mq record has != 4 columns (a = converted fields)      :219-222
                       3: "The two files are not in the same order"                  :214-217,:224-227
                       4: not incremental at one base resolution (a = prev_pos, b = start)  :249-252
                       5: end != start + 1 (a = start, b = end)                      :256-259 */
    int64_t record; /* 0-based index of the record */
    int32_t a, b;
} cornetto_bgerr_t;

/* host buffers the device can read at full PCIe rate (for the file pieces handed to cornetto_bgin_feed) */
void *cornetto_pinned_alloc(size_t bytes);
void cornetto_pinned_free(void *p);

int cornetto_bgin_open(cornetto_accel_t *h, cornetto_bgin_t **out);
void cornetto_bgin_close(cornetto_accel_t *h, cornetto_bgin_t *b);

/* Feed the next bytes of cov-total (tot) and cov-mq (mq); any split points, either length may be 0.  Records
 * are four white-space separated tokens exactly as fscanf("%s\t%d\t%d\t%d\n") reads them.  `final` is a bit set:
 * bit 0 = no more cov-total bytes will follow, bit 1 = no more cov-mq bytes.  Like the reference's loop the
 * ingest is done as soon as cov-total is exhausted and every one of its records found its partner
 * (cornetto_bgin_done()).  Returns CORNETTO_OK, or CORNETTO_E_FORMAT when a check of the reference fails
 * (details: cornetto_bgin_error()). */
int cornetto_bgin_feed(cornetto_accel_t *h, cornetto_bgin_t *b, const char *tot, int64_t n_tot, const char *mq, int64_t n_mq,
                       int final);
/* Optional: the pieces the NEXT cornetto_bgin_feed() will be given (pinned memory, untouched until that feed has returned) start their way to the
 * device now, on a copy queue of their own: called in front of a feed, the upload of the next pieces runs beside that feed's kernels (a round of
 * 2 x 64 MB is 4.6 ms of PCIe and ~3 ms of kernels and small read-backs).  The feed that follows with the same pointers and lengths takes the bytes
 * from the device; any other feed waits for the copy and ignores it. */
int cornetto_bgin_prefetch(cornetto_accel_t *h, cornetto_bgin_t *b, const char *tot, int64_t n_tot, const char *mq, int64_t n_mq);
/* bytes of each file handed over but not consumed yet (records without a partner in the other file): a
 * caller that tops both up to the same amount keeps them bounded */
void cornetto_bgin_pending(const cornetto_bgin_t *b, int64_t *pend_tot, int64_t *pend_mq);
/* once the ingest is done: white-space separated tokens of cov-mq behind its last consumed record.  The reference stops reading
 * when cov-total ends (src/boringbits_main.c:204-207) and never sees them; a caller that cuts the files into shares needs the
 * count: leftover tokens in a share that is not the last would have been the partners of the next share's cov-total records. */
int64_t cornetto_bgin_unmatched_mq(const cornetto_bgin_t *b);
const cornetto_bgerr_t *cornetto_bgin_error(const cornetto_bgin_t *b);
int cornetto_bgin_done(const cornetto_bgin_t *b);

/* After a successful final feed: the resident coverage object (ready for cornetto_cov_prepare), the contig
 * names in file order (malloc'd array of malloc'd strings: free() each, then the array) and the number of
 * depth values that were clamped to 65535 (the reference prints a WARNING for each, :261-268). */
int cornetto_bgin_finish(cornetto_accel_t *h, cornetto_bgin_t *b, cornetto_cov_t **cov, int32_t *n_ctg, char ***names,
                         int64_t *n_clamped);

/* ---------------------------------------------------------------------------------------------------
 * FASTQ reads: record framing of `cornetto seq` / `cornetto sdust reads.fastq` on the device
 * ------------------------------------------------------------------------------------------------- */

/* one plain four-line FASTQ record of the text handed to cornetto_fastq_split(); offsets are into that text */
typedef struct cornetto_fqrec {
    int64_t head;        /* the record's '@'; the name follows it */
    int64_t seq;         /* first base */
    int64_t qual;        /* first quality value */
    int32_t len;         /* bases = quality values (kseq_t.seq.l; a trailing '\r' is dropped as src/kseq.h:138 does) */
    int32_t name_len;    /* kseq_t.name.l: bytes up to the first white space */
    int32_t comment_len; /* kseq_t.comment.l: the comment starts at head + name_len + 2 */
    int32_t keep;        /* 1 if len >= min_len (the test of src/seq.c:120) */
} cornetto_fqrec_t;

/* Frame the FASTQ text `text[0..n)` (host memory, any split point of the file; at most 2^32-256 bytes) the way
 * kseq_read() does (src/kseq.h:184-224 as called by src/seq.c:116 and src/sdust/sdust.c:196) — for as long as the text
 * is made of plain records: '@' line, ONE sequence line, '+' line, ONE quality line of the same length.  That is how
 * basecallers write FASTQ, and for such records the device indexes every line at once instead of walking the bytes.
 *   recs / n_recs   the leading plain records, in input order (cornetto_free)
 *   consumed        bytes of `text` those records cover; hand the rest over again in front of the following bytes
 *   plain           0: what follows at text + consumed is not a plain record (a FASTA record, wrapped sequence or quality,
 *                   blank or stray lines, a quality of another length, the cut-off end of the input) — continue there
 *                   with a sequential kseq reader, which also produces the reference's error behaviour;
 *                   1: only an incomplete record (or nothing) is left, feed more bytes
 *   final           non-zero when no bytes follow `text` (its last line then need not end in a newline)
 *   reads           if not NULL: the bases of the records with len >= min_len, resident in HBM in input order, ready for
 *                   cornetto_sdust_asm() / cornetto_telofind() (cornetto_asm_free) — read i of it is the i-th record
 *                   with keep == 1 */
int cornetto_fastq_split(cornetto_accel_t *h, const char *text, int64_t n, int final, int32_t min_len, cornetto_fqrec_t **recs,
                         int64_t *n_recs, int64_t *consumed, int32_t *plain, cornetto_asm_t **reads);

/* one FASTA record of the text handed to cornetto_fasta_split() */
typedef struct cornetto_farec {
    int64_t head;        /* the record's '>'; the name follows it */
    int64_t len;         /* bases (kseq_t.seq.l): the payload of every line up to the next '>' line */
    int32_t name_len;    /* kseq_t.name.l */
    int32_t pad;
} cornetto_farec_t;

/* The same for FASTA text (kseq_read in front of telofind / sdust / fa2bed: src/find_telomere.c:101, src/sdust/sdust.c:196):
 * `text` must begin with the '>' of a record.  Plain here means that no line begins with '@' or '+' (kseq would read
 * FASTQ there); sequence lines may be wrapped at any width, empty lines and CRLF are handled as kseq does (:138,:202).
 * Unless `final`, the last record of the piece is never returned (it may go on in the following bytes): `consumed` stops
 * at its '>'.  seqs (optional): the sequences of the returned records resident in HBM, newlines removed. */
int cornetto_fasta_split(cornetto_accel_t *h, const char *text, int64_t n, int final, cornetto_farec_t **recs, int64_t *n_recs,
                         int64_t *consumed, int32_t *plain, cornetto_asm_t **seqs);

/* The same over a text that is put on the device slab by slab — a whole uncompressed FASTA file read through a small ring of pinned
 * slabs instead of pinned pieces that must each hold whole records (`cornetto telofind` / `sdust` on an assembly: one framing and one
 * scan for the file; src/find_telomere.c:101, src/sdust/sdust.c:196 read it record by record).  cornetto_text_open(): a device buffer
 * of `capacity` bytes (at most 2^32-256).  cornetto_text_put(): bytes [at, at + n) of the text from `slab` (pinned memory:
 * cornetto_pinned_alloc), asynchronously, on copy queue `slot` (0..3: puts of different slots run side by side — one copy in
 * flight moves ~28 GB/s over PCIe, several ~45); cornetto_text_wait() returns when the last put of that slot has left its slab.  cornetto_fasta_split_text(): waits for every
 * put, then frames the first n bytes exactly as cornetto_fasta_split() frames `text` (head offsets are offsets into the text). */
typedef struct cornetto_text cornetto_text_t;
int cornetto_text_open(cornetto_accel_t *h, int64_t capacity, cornetto_text_t **out);
void cornetto_text_free(cornetto_accel_t *h, cornetto_text_t *t);
int cornetto_text_put(cornetto_accel_t *h, cornetto_text_t *t, const char *slab, int64_t n, int64_t at, int slot);
int cornetto_text_wait(cornetto_accel_t *h, cornetto_text_t *t, int slot);
int cornetto_fasta_split_text(cornetto_accel_t *h, cornetto_text_t *t, int64_t n, int final, cornetto_farec_t **recs, int64_t *n_recs,
                              int64_t *consumed, int32_t *plain, cornetto_asm_t **seqs);

/* ---------------------------------------------------------------------------------------------------
 * panel interval stage — scripts/create-cornetto.sh:41-66 without bedtools / sort / awk (parity with bedtools itself
 * is unpinned: see cornetto_amd/csrc/panel.hip)
 * ------------------------------------------------------------------------------------------------- */

/* cornetto_cov_select() followed, on the device, by `bedtools merge -d merge_dist` and `awk '($3-$2)>=min_len'`
 * (create-cornetto.sh:44-47: -d 1000, 30000): the selected windows never leave the device, only the merged intervals
 * {ctg, start, finish} come back (by contig index, by start).  Release with cornetto_free(). */
int cornetto_cov_select_merged(cornetto_accel_t *h, const cornetto_cov_t *c, int32_t lo, int32_t hi, float low_mq, int32_t edge_len,
                               int32_t min_ctg_len, int boring, int32_t merge_dist, int32_t min_len, cornetto_ivl_t **ivls,
                               int64_t *n_ivls);

/* `bedtools merge -d dist` of intervals that are in (contig, start) order (CORNETTO_E_ARG if they are not), on the
 * device: an interval that starts at most `dist` past the largest finish so far in its contig is merged into it. */
int cornetto_ivl_merge(cornetto_accel_t *h, const cornetto_ivl_t *in, int64_t n, int32_t dist, cornetto_ivl_t **out, int64_t *n_out);

typedef struct {
    int32_t min_lowq_len; /* keep hifiasm "low quality" rows of at least this length (8000, create-cornetto.sh:50; 7500, recreate-cornetto.sh:35) */
    int32_t extend;       /* rows that start beyond extend_gate grow by this many bases to the LEFT (40000, :53; 40000, recreate :38) */
    int32_t edge_len;     /* add the first / last edge_len bases of every longer contig (200000, :56) */
    int32_t merge_dist;   /* bedtools merge -d (200000, :59) */
    int32_t min_ctg_len;  /* contigs shorter than this contribute nothing (800000, :65; 1000000, recreate :47) */
    int32_t extend_right; /* ... and by this many to the RIGHT (40000, :53; 50000, recreate :38) */
    int32_t extend_gate;  /* the awk's `if ($2 > gate)` (40000, :53; 50000, recreate :38) */
} cornetto_panel_opt_t;
void cornetto_panel_defaults(cornetto_panel_opt_t *opt);          /* scripts/create-cornetto.sh */
void cornetto_panel_defaults_recreate(cornetto_panel_opt_t *opt); /* scripts/recreate-cornetto.sh:34-49 (no coverage stage: n_fun = 0) */

/* Steps 4-9 of create-cornetto.sh (:50-66) on index-based intervals: fun = output of cornetto_cov_select_merged
 * (ctg = index into ctg_len, the assembly order), lowq = the rows of the hifiasm low-quality BED (any order).
 * Result: the "boring bits" before bigenough, {ctg, start, finish}, in assembly order; cornetto_free() it.  Host only. */
int cornetto_panel_boring(const int32_t *ctg_len, int32_t n_ctg, const cornetto_ivl_t *fun, int64_t n_fun, const cornetto_ivl_t *lowq,
                          int64_t n_lowq, const cornetto_panel_opt_t *opt, cornetto_ivl_t **boring, int64_t *n_boring);

/* ---------------------------------------------------------------------------------------------------
 * telobreaks — src/telomere_breaks.c:47-172 (the consumer of the sdust BED and the telofind TSV)
 * ------------------------------------------------------------------------------------------------- */

/* one row of the telofind TSV as telobreaks reads it (:97): contig, start, end, matched length (6th column) */
typedef struct {
    int32_t ctg, start, end, matched;
} cornetto_telrow_t;

/* The bitset stage of telobreaks (:79-148) for contigs given by index (the caller resolves the names; rows of names
 * that are not in the lens file are dropped, as the reference does at :83,:100: give them ctg = -1).
 * sd: low-complexity intervals [start, finish), any order, may overlap or touch.  tel: telofind rows; those with
 * matched < 24 are ignored (MIN_TEL, :10,:98).  out[] = the numbers the reference prints (:140-142): for every maximal
 * run of the low-complexity bitset that contains a telomere row together with its 100-base flanks (clipped to the
 * contig): {ctg, first position - 1 clamped at 0, last position}, by contig index, then by position.
 * The reference prints contigs in khash bucket order: cornetto_khash_str_order().  An sd interval that ends beyond its contig
 * is cut at the contig's end: sdust itself prints such intervals for a low-complexity run at the end of a contig (up to W
 * beyond the last base), the reference sets those bits beyond its bitset (:85) and never reads them (:103,:118,:136).  Other
 * coordinates outside the contig (a negative start, a telomere row beyond the end) are unchecked heap indices in the reference;
 * here they are CORNETTO_E_FORMAT.  Release out with cornetto_free(). */
int cornetto_telobreaks(cornetto_accel_t *h, const int32_t *ctg_len, int32_t n_ctg, const cornetto_ivl_t *sd, int64_t n_sd,
                        const cornetto_telrow_t *tel, int64_t n_tel, cornetto_ivl_t **out, int64_t *n_out);

/* Host helper: the iteration order of the reference's khash string map (klib khash 0.2.8, src/khash.h) after
 * kh_put() of names[0..n) in that order — the order in which telobreaks prints its contigs (:133).  slot[i] = dense
 * id of the distinct key of names[i] (ids in order of first appearance); order[k] = id in the k-th occupied bucket.
 * Returns the number of distinct keys, -1 on a bad argument.  Needs no device. */
int32_t cornetto_khash_str_order(const char *const *names, int32_t n, int32_t *slot, int32_t *order);

#ifdef __cplusplus
}
#endif
#endif /* CORNETTO_ACCEL_H */

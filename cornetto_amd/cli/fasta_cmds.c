/* fasta_cmds.c — the FASTA/FASTQ driven sub-commands: telofind, sdust (device scans), fa2bed, seq (host
 * only).  Reference: src/find_telomere.c:83-111, src/sdust/sdust.c:179-207, src/assbed.c:50-107,
 * src/seq.c:53-138.  Records are read into a batch (whole assembly, or $CORNETTO_BATCH_BASES bases of
 * reads at a time), scanned by ONE device pass per batch, and printed in input order. */
#include <errno.h>
#include <getopt.h>
#include <stdlib.h>
#include <fcntl.h>
#include <pthread.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "cli.h"

typedef void (*batch_fn)(cornetto_accel_t *h, cli_batch_t *b, void *arg);

/* read every record of `fx` (closed here), one device pass per batch; `h` may be an open handle (kept open) or NULL */
static void batches_of(cli_fastx_t *fx, cornetto_accel_t *h_open, batch_fn fn, void *arg)
{
    cornetto_accel_t *h = h_open;
    cli_str_t name = {0, 0, 0}, comment = {0, 0, 0}, seq = {0, 0, 0}, qual = {0, 0, 0};
    cli_batch_t b;
    memset(&b, 0, sizeof(b));
    const int64_t limit = cli_batch_limit();
    int64_t l;
    if (!h) cli_accel_open_begin();     /* HIP initialises while the records are read */
    while ((l = cli_fastx_read(fx, &name, &comment, &seq, &qual)) >= 0) {
        if (l > 0x7fffffffLL) {
            CLI_ERROR("record %s has %lld bases; the reference's reader is limited to 2^31-1 (src/kseq.h:185)", name.s, (long long)l);
            exit(EXIT_FAILURE);
        }
        cli_batch_take(&b, name.s, &seq);
        if (b.bases >= limit) {
            if (!h) h = cli_accel_open_end();
            fn(h, &b, arg);
            cli_batch_clear(&b);
        }
    }
    if (b.n) {
        if (!h) h = cli_accel_open_end();
        fn(h, &b, arg);
        cli_batch_clear(&b);
    }
    if (!h) cli_accel_open_cancel();
    if (h && !h_open) cornetto_accel_close(h);
    free(b.names);
    free(b.seqs);
    free(b.lens);
    free(name.s);
    free(comment.s);
    free(seq.s);
    free(qual.s);
    cli_fastx_close(fx);
}

/* ---------------------------------------------------------------- records on the device */
/* a record as the scans need it: its name (not NUL-terminated when it points into a file piece) and length */
typedef struct {
    const char *name;
    int32_t name_len;
    int64_t len;
} cli_recname_t;

/* scan the resident sequences `a` (record i of it = r[i]) and print the sub-command's lines */
typedef void (*scan_fn)(cornetto_accel_t *h, const cli_recname_t *r, int64_t n, const cornetto_asm_t *a, void *arg);

static void print_hit(const char *name, size_t name_len, int64_t len, const cornetto_hit_t *h);
static void print_ivl(const char *name, size_t name_len, const cornetto_ivl_t *v);

/* ---------------------------------------------------------------- telofind */
static void telofind_scan(cornetto_accel_t *h, const cli_recname_t *r, int64_t n_rec, const cornetto_asm_t *a, void *arg)
{
    (void)n_rec;
    cornetto_hit_t *hits = NULL;
    int64_t n = 0;
    cli_accel_check(h, cornetto_telofind(h, a, (const char *)arg, &hits, &n), "telofind");
    for (int64_t i = 0; i < n; ++i) print_hit(r[hits[i].ctg].name, (size_t)r[hits[i].ctg].name_len, r[hits[i].ctg].len, &hits[i]);
    cli_out_flush();
    cornetto_free(hits);
}

static void sdust_scan(cornetto_accel_t *h, const cli_recname_t *r, int64_t n_rec, const cornetto_asm_t *a, void *arg);

/* ---------------------------------------------------------------- several GPUs of one node
 * CORNETTO_DEVICES=0,1,2,...  (two or more ordinals; a device may be named twice): every scan is independent per record
 * (src/find_telomere.c:101-105, src/sdust/sdust.c:196-203), so the records of a batch are dealt to the devices —
 * longest first, each to the device with the least bases so far (LPT) — and every device gets one host thread with its
 * own handle: upload, scan, results.  Printing happens after the join, record by record in INPUT order (each device's
 * results come back ordered by its local record index, and a device's records keep their input order, so one cursor per
 * device suffices): the output is byte for byte that of one device.  Nothing is exchanged between the devices. */
typedef struct {
    int kind;                    /* 0 telofind, 1 sdust */
    const char *motif;
    int T, W;
} multi_what_t;

typedef struct {
    int dev;
    cornetto_accel_t *h;         /* opened by the worker on its first batch, kept for the next ones */
    const multi_what_t *what;
    const cli_batch_t *b;
    int32_t *mine, n_mine;       /* batch indices of the records of this device, ascending */
    cornetto_hit_t *hits;
    cornetto_ivl_t *ivls;
    int64_t n_res;
    int rc;
    char err[600];
} multi_dev_t;

static void *multi_worker(void *p)
{
    multi_dev_t *d = (multi_dev_t *)p;
    d->rc = CORNETTO_OK;
    d->hits = NULL;
    d->ivls = NULL;
    d->n_res = 0;
    d->err[0] = 0;
    if (!d->h) {
        d->rc = cornetto_accel_open(&d->h, d->dev, NULL);
        if (d->rc != CORNETTO_OK) {
            snprintf(d->err, sizeof(d->err), "cannot open HIP device %d: %s", d->dev, cornetto_accel_strerror(d->rc));
            return NULL;
        }
    }
    if (d->n_mine == 0) return NULL;
    const uint8_t **seqs = (const uint8_t **)cli_xmalloc((size_t)d->n_mine * sizeof(*seqs));
    int64_t *lens = (int64_t *)cli_xmalloc((size_t)d->n_mine * sizeof(*lens));
    for (int32_t i = 0; i < d->n_mine; ++i) {
        seqs[i] = d->b->seqs[d->mine[i]];
        lens[i] = d->b->lens[d->mine[i]];
    }
    cornetto_asm_t *a = NULL;
    d->rc = cornetto_asm_upload(d->h, seqs, lens, d->n_mine, &a);
    if (d->rc == CORNETTO_OK) {
        if (d->what->kind == 0) d->rc = cornetto_telofind(d->h, a, d->what->motif, &d->hits, &d->n_res);
        else d->rc = cornetto_sdust_asm(d->h, a, d->what->T, d->what->W, &d->ivls, &d->n_res);
    }
    if (d->rc != CORNETTO_OK)
        snprintf(d->err, sizeof(d->err), "device %d: %s (%s)", d->dev, cornetto_accel_last_error(d->h), cornetto_accel_strerror(d->rc));
    if (a) cornetto_asm_free(d->h, a);
    free(seqs);
    free(lens);
    return NULL;
}

static void print_hit(const char *name, size_t name_len, int64_t len, const cornetto_hit_t *h)
{
    cli_out_bytes(name, name_len);   /* "%s\t%zu\t%d\t%zu\t%zu\t%zu\n": src/find_telomere.c:51,56 */
    cli_out_char('\t');
    cli_out_int(len);
    cli_out_char('\t');
    cli_out_int(h->strand);
    cli_out_char('\t');
    cli_out_int(h->start);
    cli_out_char('\t');
    cli_out_int(h->end);
    cli_out_char('\t');
    cli_out_int(h->end - h->start);
    cli_out_char('\n');
}

static void print_ivl(const char *name, size_t name_len, const cornetto_ivl_t *v)
{
    cli_out_bytes(name, name_len);   /* "%s\t%d\t%d\n": src/sdust/sdust.c:201 */
    cli_out_char('\t');
    cli_out_int(v->start);
    cli_out_char('\t');
    cli_out_int(v->finish);
    cli_out_char('\n');
}

static void multi_batch(multi_dev_t *dv, int n_dev, const cli_batch_t *b)
{
    /* LPT: records by descending length (ties: input order), each to the least loaded device */
    int32_t *order = (int32_t *)cli_xmalloc(((size_t)b->n + 1) * sizeof(*order));
    cli_order_by_length_desc(b->lens, NULL, b->n, order);   /* (a batch may hold hundreds of thousands of reads) */
    int64_t load[CLI_MAX_DEV];
    int32_t *owner = (int32_t *)cli_xmalloc(((size_t)b->n + 1) * sizeof(*owner));
    for (int d = 0; d < n_dev; ++d) { load[d] = 0; dv[d].n_mine = 0; dv[d].b = b; }
    for (int32_t k = 0; k < b->n; ++k) {
        int best = 0;
        for (int d = 1; d < n_dev; ++d)
            if (load[d] < load[best]) best = d;
        owner[order[k]] = best;
        load[best] += b->lens[order[k]] + 1; /* (+1: empty records are spread as well) */
    }
    for (int d = 0; d < n_dev; ++d) dv[d].mine = (int32_t *)cli_xmalloc(((size_t)b->n + 1) * sizeof(int32_t));
    for (int32_t i = 0; i < b->n; ++i) dv[owner[i]].mine[dv[owner[i]].n_mine++] = i; /* ascending = input order */
    pthread_t th[CLI_MAX_DEV];
    int started[CLI_MAX_DEV];
    for (int d = 0; d < n_dev; ++d) started[d] = d > 0 && pthread_create(&th[d], NULL, multi_worker, &dv[d]) == 0;
    for (int d = 0; d < n_dev; ++d)
        if (!started[d]) multi_worker(&dv[d]);
    for (int d = 0; d < n_dev; ++d)
        if (started[d]) pthread_join(th[d], NULL);
    for (int d = 0; d < n_dev; ++d)
        if (dv[d].rc != CORNETTO_OK) {
            CLI_ERROR("%s", dv[d].err);
            exit(EXIT_FAILURE);
        }
    /* input order: record i is local record `local[owner]` of its device; its results are the next ones of that device */
    int64_t cur[CLI_MAX_DEV];
    int32_t local[CLI_MAX_DEV];
    for (int d = 0; d < n_dev; ++d) { cur[d] = 0; local[d] = 0; }
    for (int32_t i = 0; i < b->n; ++i) {
        multi_dev_t *d = &dv[owner[i]];
        const int32_t li = local[owner[i]]++;
        int64_t *c = &cur[owner[i]];
        const size_t nl = strlen(b->names[i]);
        if (d->what->kind == 0)
            for (; *c < d->n_res && d->hits[*c].ctg == li; ++*c) print_hit(b->names[i], nl, b->lens[i], &d->hits[*c]);
        else
            for (; *c < d->n_res && d->ivls[*c].ctg == li; ++*c) print_ivl(b->names[i], nl, &d->ivls[*c]);
    }
    cli_out_flush();
    for (int d = 0; d < n_dev; ++d) {
        cornetto_free(dv[d].hits);
        cornetto_free(dv[d].ivls);
        free(dv[d].mine);
        dv[d].hits = NULL;
        dv[d].ivls = NULL;
    }
    free(order);
    free(owner);
}

/* the whole sub-command over several devices: the sequential reader fills batches, every batch is dealt out */
static void multi_stream(const char *path, int must_open, const multi_what_t *what, const int *devs, int n_dev)
{
    cli_fastx_t *fx = cli_fastx_open(path);
    if (!fx) {
        if (must_open) {
            CLI_ERROR("Could not to open file %s: %s", path, strerror(errno)); /* F_CHK, src/error.h:114-119: its words */
            exit(EXIT_FAILURE);
        }
        return;
    }
    multi_dev_t dv[CLI_MAX_DEV];
    memset(dv, 0, sizeof(dv));
    for (int d = 0; d < n_dev; ++d) { dv[d].dev = devs[d]; dv[d].what = what; }
    cli_str_t name = {0, 0, 0}, comment = {0, 0, 0}, seq = {0, 0, 0}, qual = {0, 0, 0};
    cli_batch_t b;
    memset(&b, 0, sizeof(b));
    const int64_t limit = cli_batch_limit();
    int64_t l;
    while ((l = cli_fastx_read(fx, &name, &comment, &seq, &qual)) >= 0) {
        if (l > 0x7fffffffLL) {
            CLI_ERROR("record %s has %lld bases; the reference's reader is limited to 2^31-1 (src/kseq.h:185)", name.s, (long long)l);
            exit(EXIT_FAILURE);
        }
        cli_batch_take(&b, name.s, &seq);
        if (b.bases >= limit) {
            multi_batch(dv, n_dev, &b);
            cli_batch_clear(&b);
        }
    }
    if (b.n) {
        multi_batch(dv, n_dev, &b);
        cli_batch_clear(&b);
    }
    for (int d = 0; d < n_dev; ++d)
        if (dv[d].h) cornetto_accel_close(dv[d].h);
    free(b.names);
    free(b.seqs);
    free(b.lens);
    free(name.s);
    free(comment.s);
    free(seq.s);
    free(qual.s);
    cli_fastx_close(fx);
}

/* a batch of the sequential reader: upload, then the same scan */
typedef struct {
    scan_fn scan;
    void *arg;
} batch_scan_t;

static void scan_batch(cornetto_accel_t *h, cli_batch_t *b, void *arg)
{
    const batch_scan_t *bs = (const batch_scan_t *)arg;
    cornetto_asm_t *a = NULL;
    cli_accel_check(h, cornetto_asm_upload(h, (const uint8_t *const *)b->seqs, b->lens, b->n, &a), "copying sequences to the GPU");
    cli_recname_t *r = (cli_recname_t *)cli_xmalloc(((size_t)b->n + 1) * sizeof(*r));
    for (int32_t i = 0; i < b->n; ++i) {
        r[i].name = b->names[i];
        r[i].name_len = (int32_t)strlen(b->names[i]);
        r[i].len = b->lens[i];
    }
    bs->scan(h, r, b->n, a, bs->arg);
    free(r);
    cornetto_asm_free(h, a);
}

/* FASTA / FASTQ file -> scans, with the records framed on the device wherever the text is plain (cornetto_fasta_split,
 * cornetto_fastq_split): the file goes to the device in pieces as it is, names are printed straight from the piece.
 * Anything else — wrapped FASTQ, stray lines, a FASTQ record inside a FASTA file, the reference's error cases — is read
 * by the sequential reader (cli/fastx.c) from the first byte the device was not sure about: the output is kseq's either
 * way.  CORNETTO_FASTQ_PIECE = bytes per piece; CORNETTO_FASTQ_SPLIT=host = sequential reader only. */
static int64_t piece_bytes(int fasta, const char *path, gzFile fp)
{
    const char *e = getenv("CORNETTO_FASTQ_PIECE");
    int64_t v = e ? atoll(e) : 0;
    if (v < 64) {
        v = 256LL << 20;
        if (fasta) { /* a record must fit into a piece: the whole file at once when its size is known; else grown on demand */
            struct stat st;
            v = 64LL << 20;
            if (strcmp(path, "-") && gzdirect(fp) && stat(path, &st) == 0 && S_ISREG(st.st_mode)) v = (int64_t)st.st_size + 16;
            if (v > (256LL << 20)) v = 256LL << 20; /* (pinning and unpinning 1 GiB cost 0.27 s of a 0.9 s run; a longer record grows the piece) */
        }
    }
    if (v > 0xF0000000LL) v = 0xF0000000LL;
    return v;
}

/* one thread copies from the page cache at 5-8 GB/s: the largest share of the wall time of a 3 GB assembly */
static int read_threads(void)
{
    const char *e = getenv("CORNETTO_READ_THREADS");
    const int v = e ? atoi(e) : 8;
    return v < 1 ? 1 : v;
}
#define READ_THREADS read_threads()

#define TRACE(what)                                                                                      \
    do {                                                                                                 \
        if (trace) fprintf(stderr, "[cli trace] %-28s %8.1f ms\n", (what), (cli_realtime() - t_begin) * 1e3); \
    } while (0)

/* ---- read-ahead for an uncompressed FASTA file: while piece k is on the device (upload, framing, scan, printing), a thread finds where
 * piece k + 1 begins — at the '>' of the last record of piece k, which a piece that is not the file's last leaves unconsumed
 * (cornetto_fasta_split) — and reads it into a second pinned buffer (the unconsumed bytes come from the page cache once more:
 * cheaper than carrying tens of megabytes of a contig over).  The device's `consumed` is the authority: if it differs from the
 * prediction (text that is not plain FASTA), what was read ahead is dropped and the caller goes on as without it. */
typedef struct {
    int fd, n_threads, started;
    const char *cur;      /* piece k */
    int64_t cur_n, cur_off; /* its bytes and the file offset of its first byte */
    char *dst;            /* buffer of piece k + 1 (allocated by the thread on first use, and again when it is smaller than cap) */
    int64_t dst_cap;      /* its size */
    int64_t cap;          /* bytes to read ahead */
    int64_t pred;         /* out: predicted consumed bytes of piece k (-1: no prediction: nothing was read) */
    int64_t got;          /* out: bytes of piece k + 1 (-1: read error) */
    int eof;              /* out: the file ends inside piece k + 1 */
    pthread_t th;
} fa_ahead_t;

static void *fa_ahead_thread(void *p)
{
    fa_ahead_t *a = (fa_ahead_t *)p;
    a->pred = -1;
    a->got = 0;
    a->eof = 0;
    /* the last '>' that begins a line */
    int64_t at = a->cur_n;
    while (at > 0) {
        const char *q = (const char *)memrchr(a->cur, '>', (size_t)at);
        if (!q) break;
        at = (int64_t)(q - a->cur);
        if (at > 0 && a->cur[at - 1] == '\n') {
            a->pred = at;
            break;
        }
    }
    if (a->pred <= 0) {
        a->pred = -1;
        return NULL;
    }
    if (a->dst && a->dst_cap < a->cap) {          /* (the small first piece of the file, handed back as the second buffer) */
        cornetto_pinned_free(a->dst);
        a->dst = NULL;
    }
    if (!a->dst) {
        a->dst = (char *)cornetto_pinned_alloc((size_t)a->cap);
        a->dst_cap = a->cap;
    }
    if (!a->dst) {
        a->dst_cap = 0;
        a->pred = -1;
        return NULL;
    }
    a->got = cli_pread_parallel(a->fd, a->dst, a->cap, a->cur_off + a->pred, a->n_threads);
    if (a->got >= 0 && a->got < a->cap) a->eof = 1;
    return NULL;
}


/* ---- an uncompressed FASTA FILE as ONE text on the device (round 6) ------------------------------------------------------------------
 * The piece loop below pins buffers that must each hold whole records (256 MiB and more for an assembly), reads every contig that
 * straddles a piece border twice, and frames, uploads and scans a dozen pieces one after the other, each a cold first pass over a new
 * resident object: 0.45-0.63 s for the 3.16 GB assembly of which 0.15 s were the scans' pipeline.  Here the file goes through a small ring
 * of pinned slabs (reader threads fill them from the page cache, two copy queues empty them: cornetto_text_put) into one device buffer, and
 * the whole text is framed and scanned ONCE (cornetto_fasta_split_text).  Record names are read back from the file by offset.  Texts are
 * limited to 2^32-256 bytes: a longer file takes several rounds, each ending at its last complete record.
 * -> 1: the file (or all of it up to *resume_off, from where the sequential reader must go on: text that is not plain FASTA) was handled */
#define WHOLE_SLOTS 16
typedef struct {
    int fd, n_slots, failed, pin_failed;
    int64_t off0, total, slab, n_slab;
    char *ring[WHOLE_SLOTS];
    int64_t filled[WHOLE_SLOTS];  /* slab index + 1 the slot holds (0: none) */
    int64_t allowed[WHOLE_SLOTS]; /* the slab index the slot may be filled with */
    int64_t got[WHOLE_SLOTS];     /* bytes of it */
    int64_t next;
    pthread_mutex_t mu;
    pthread_cond_t cv;
} whole_ring_t;

static void *whole_reader(void *p)
{
    whole_ring_t *w = (whole_ring_t *)p;
    for (;;) {
        pthread_mutex_lock(&w->mu);
        const int64_t i = w->next++;
        if (i >= w->n_slab || w->failed) {
            pthread_mutex_unlock(&w->mu);
            return NULL;
        }
        const int s = (int)(i % w->n_slots);
        while ((w->allowed[s] != i || !w->ring[s]) && !w->failed && !w->pin_failed) pthread_cond_wait(&w->cv, &w->mu);
        const int stop = w->failed || w->pin_failed;
        pthread_mutex_unlock(&w->mu);
        if (stop) return NULL;
        const int64_t at = i * w->slab, want = w->total - at < w->slab ? w->total - at : w->slab;
        const int64_t have = cli_read_at(w->fd, w->ring[s], want, w->off0 + at, NULL);
        pthread_mutex_lock(&w->mu);
        if (have < want) w->failed = 1;          /* (a file that shrank under us, an I/O error) */
        w->got[s] = have;
        w->filled[s] = i + 1;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
    }
}

/* the slabs are page-locked one after the other (~5-20 GB/s: 128 MB take 10-25 ms) by a thread of their own, while the first ones are already
 * being filled and copied */
static void *whole_pinner(void *p)
{
    whole_ring_t *w = (whole_ring_t *)p;
    for (int s = 0; s < w->n_slots; ++s) {
        char *m = (char *)cornetto_pinned_alloc((size_t)w->slab);
        pthread_mutex_lock(&w->mu);
        if (!m) w->pin_failed = 1;
        w->ring[s] = m;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
        if (!m) return NULL;
    }
    return NULL;
}

static int stream_whole_fasta(const char *path, int fd, int64_t size, scan_fn scan, void *arg, cornetto_accel_t **h_io, int64_t *resume_off, int trace,
                              double t_begin)
{
    const int64_t LIMIT = 0xFFFFFF00LL - 4096;
    cornetto_accel_t *h = *h_io;
    if (!h) *h_io = h = cli_accel_open_end();
    TRACE("device open");
    cornetto_text_t *t = NULL;
    const int64_t cap = size < LIMIT ? size : LIMIT;
    cli_accel_check(h, cornetto_text_open(h, cap, &t), "allocating the text on the GPU");
    TRACE("device text allocated");
    whole_ring_t w;
    memset(&w, 0, sizeof(w));
    pthread_mutex_init(&w.mu, NULL);
    pthread_cond_init(&w.cv, NULL);
    w.fd = fd;
    /* sixteen slabs of 8 MiB: a reader thread per slab in flight copies from the page cache at 4-6 GB/s, the two copy queues take ~45 GB/s */
    w.slab = 8LL << 20;
    if (cap < w.slab * WHOLE_SLOTS) w.slab = ((cap + WHOLE_SLOTS - 1) / WHOLE_SLOTS + 65535) & ~65535LL;   /* (a small file: a small ring) */
    w.n_slots = (int)((cap + w.slab - 1) / w.slab);
    if (w.n_slots > WHOLE_SLOTS) w.n_slots = WHOLE_SLOTS;
    if (w.n_slots < 1) w.n_slots = 1;
    pthread_t pin_th;
    if (pthread_create(&pin_th, NULL, whole_pinner, &w) != 0) whole_pinner(&w);
    else pthread_detach(pin_th);
    int n_thr = getenv("CORNETTO_READ_THREADS") ? READ_THREADS : 16;
    if (n_thr > w.n_slots) n_thr = w.n_slots;
    int64_t off = 0;
    int plain_all = 1;
    while (off < size && plain_all) {
        const int64_t n = size - off < cap ? size - off : cap;
        const int final = off + n == size;
        w.off0 = off;
        w.total = n;
        w.n_slab = (n + w.slab - 1) / w.slab;
        w.next = 0;
        w.failed = 0;
        for (int s = 0; s < w.n_slots; ++s) { w.filled[s] = 0; w.allowed[s] = s; }
        pthread_t th[64];
        int n_started = 0;
        for (int k = 0; k < n_thr && k < 64; ++k)
            if (pthread_create(&th[n_started], NULL, whole_reader, &w) == 0) ++n_started;
        if (n_started == 0) { CLI_ERROR("could not start a reader thread"); exit(EXIT_FAILURE); }
        for (int64_t i = 0; i < w.n_slab; ++i) {
            const int s = (int)(i % w.n_slots);
            pthread_mutex_lock(&w.mu);
            while (w.filled[s] != i + 1 && !w.failed && !w.pin_failed) pthread_cond_wait(&w.cv, &w.mu);
            const int64_t got = w.got[s];
            const int failed = w.pin_failed ? 2 : w.failed;
            pthread_mutex_unlock(&w.mu);
            if (failed == 2) { CLI_ERROR("could not allocate a %lld-byte pinned slab", (long long)w.slab); exit(EXIT_FAILURE); }
            if (failed) { CLI_ERROR("reading %s failed", path); exit(EXIT_FAILURE); }
            /* four copy queues: slab i goes out on queue i & 3 once the copy that used that queue last (slab i - 4) has left its slab, which then
             * goes back to the readers: up to four copies in flight */
            if (i >= 4) {
                cli_accel_check(h, cornetto_text_wait(h, t, (int)(i & 3)), "copying the text to the GPU");
                const int sp = (int)((i - 4) % w.n_slots);
                pthread_mutex_lock(&w.mu);
                w.allowed[sp] = i - 4 + w.n_slots;
                pthread_cond_broadcast(&w.cv);
                pthread_mutex_unlock(&w.mu);
            }
            cli_accel_check(h, cornetto_text_put(h, t, w.ring[s], got, i * w.slab, (int)(i & 3)), "copying the text to the GPU");
        }
        for (int k = 0; k < n_started; ++k) pthread_join(th[k], NULL);
        TRACE("text on the device");
        cornetto_farec_t *recs = NULL;
        cornetto_asm_t *a = NULL;
        int64_t nrec = 0, used = 0;
        int32_t plain = 1;
        cli_accel_check(h, cornetto_fasta_split_text(h, t, n, final, &recs, &nrec, &used, &plain, &a), "framing the FASTA records");
        TRACE("records framed");
        if (nrec) {
            /* the names: from the file, by offset (the slabs are gone) */
            cli_recname_t *r = (cli_recname_t *)cli_xmalloc(((size_t)nrec + 1) * sizeof(*r));
            int64_t name_bytes = 0;
            for (int64_t i = 0; i < nrec; ++i) name_bytes += recs[i].name_len + 1;
            char *names = (char *)cli_xmalloc((size_t)name_bytes + 1), *q = names;
            for (int64_t i = 0; i < nrec; ++i) {
                int64_t have = 0;
                while (have < recs[i].name_len) {
                    const ssize_t g = pread(fd, q + have, (size_t)(recs[i].name_len - have), (off_t)(off + recs[i].head + 1 + have));
                    if (g < 0 && errno == EINTR) continue;
                    if (g <= 0) { CLI_ERROR("reading %s failed", path); exit(EXIT_FAILURE); }
                    have += g;
                }
                r[i].name = q;
                r[i].name_len = recs[i].name_len;
                r[i].len = recs[i].len;
                q += recs[i].name_len + 1;
            }
            scan(h, r, nrec, a, arg);
            TRACE("scanned and printed");
            free(names);
            free(r);
        }
        cornetto_free(recs);
        cornetto_asm_free(h, a);
        off += used;
        if (!plain) plain_all = 0;                 /* what follows at `off` is for the sequential reader */
        else if (used == 0 && !final) plain_all = 0; /* one record longer than a text (2^32 bytes): the sequential reader reports it as the reference's reader would */
        else if (final) off = size;
    }
    /* (the slabs, the text and the handle are left to the end of the process: main.c leaves with _exit) */
    *resume_off = off;
    return 1;
}

static void stream_records(const char *path, int must_open, scan_fn scan, void *arg)
{
    const int trace = getenv("CORNETTO_CLI_TRACE") != NULL;
    const double t_begin = cli_realtime();
    gzFile fp = (cli_dash_is_stdin && !strcmp(path, "-")) ? gzdopen(fileno(stdin), "r") : gzopen(path, "r");
    if (!fp) {
        if (must_open) {
            CLI_ERROR("Could not to open file %s: %s", path, strerror(errno)); /* F_CHK, src/error.h:114-119: its words */
            exit(EXIT_FAILURE);
        }
        return; /* sdust: the reference has no NULL check (src/sdust/sdust.c:194) and crashes; we just stop */
    }
    gzbuffer(fp, 1 << 18);
    cornetto_accel_t *h = NULL;
    char first = 0, *buf = &first;
    int64_t have = 0, start = 0; /* unread bytes: buf[start .. have) */
    int eof = 0;
    const int r0 = gzread(fp, &first, 1);
    if (r0 == 1) have = 1;
    else eof = 1;
    const char *how = getenv("CORNETTO_FASTQ_SPLIT");
    if (have && (first == '@' || first == '>') && !(how && !strcmp(how, "host"))) {
        const int fasta = first == '>';
        int64_t piece = piece_bytes(fasta, path, fp);
        /* a piece size given by hand is kept (tests: records that do not fit go to the sequential reader) unless
         * CORNETTO_FASTQ_GROW=1 asks for the growth path as well */
        const char *grow = getenv("CORNETTO_FASTQ_GROW");
        const int64_t piece_max = getenv("CORNETTO_FASTQ_PIECE") && !(grow && atoi(grow)) ? piece : 0xF0000000LL;
        /* uncompressed regular file: its bytes are the stream's bytes, read them with pread() from here on */
        int raw_fd = -1;
        int64_t raw_off = 1; /* the first byte is in `first` */
        {
            struct stat st;
            if (strcmp(path, "-") && gzdirect(fp) && stat(path, &st) == 0 && S_ISREG(st.st_mode)) raw_fd = open(path, O_RDONLY);
        }
        {
            struct stat st;
            if (fasta && raw_fd >= 0 && fstat(raw_fd, &st) == 0 && (int64_t)st.st_size >= (256LL << 20))
                cli_accel_warm_hint(scan == sdust_scan ? CORNETTO_WARM_SDUST : CORNETTO_WARM_TELO);      /* (an assembly: its one scan should not be the runtime's first) */
        }
        cli_accel_open_begin();
        /* read-ahead (uncompressed FASTA file; CORNETTO_CLI_AHEAD=0 switches it off) */
        const char *ahead_env = getenv("CORNETTO_CLI_AHEAD");
        const int use_ahead = fasta && raw_fd >= 0 && !(ahead_env && !atoi(ahead_env));
        /* With read-ahead the FIRST piece is small (64 MiB; CORNETTO_CLI_FIRST_MB): pinning and reading it takes a quarter of the time of a
         * full piece, and that time stands in front of the first scan — beside the device's start-up, on this thread; the full-size buffers
         * are made by the read-ahead thread while the device works (round 5). */
        /* the whole file as one text on the device (see stream_whole_fasta); CORNETTO_CLI_WHOLE=0 and an explicit piece size keep the piece loop */
        {
            const char *we = getenv("CORNETTO_CLI_WHOLE");
            struct stat st;
            if (use_ahead && !(we && !atoi(we)) && !getenv("CORNETTO_FASTQ_PIECE") && fstat(raw_fd, &st) == 0 && st.st_size > 0) {
                int64_t resume = 0;
                (void)stream_whole_fasta(path, raw_fd, (int64_t)st.st_size, scan, arg, &h, &resume, trace, t_begin);
                close(raw_fd);
                if (resume < (int64_t)st.st_size) {     /* not plain from there on: the sequential reader takes the rest */
                    gzseek(fp, (z_off_t)resume, SEEK_SET);
                    batch_scan_t bs = {scan, arg};
                    cli_fastx_t *fx = cli_fastx_open_prefixed(fp, NULL, 0);
                    batches_of(fx, h, scan_batch, &bs);
                } else {
                    gzclose(fp);
                }
                TRACE("done");
                return;
            }
        }
        int64_t cap_buf = piece, cap_other = 0;
        if (use_ahead && !getenv("CORNETTO_FASTQ_PIECE")) {
            const char *fm = getenv("CORNETTO_CLI_FIRST_MB");
            const int64_t first_bytes = (int64_t)(fm && atoi(fm) > 0 ? atoi(fm) : 64) << 20;
            if (first_bytes < cap_buf) cap_buf = first_bytes;
        }
        buf = (char *)cornetto_pinned_alloc((size_t)cap_buf);
        if (!buf) {
            h = cli_accel_open_end(); /* no usable device: its message and exit(EXIT_FAILURE) */
            CLI_ERROR("could not allocate a %lld-byte pinned read buffer", (long long)piece);
            exit(EXIT_FAILURE);
        }
        buf[0] = first;
        TRACE("pinned piece allocated");
        fa_ahead_t ah;
        memset(&ah, 0, sizeof(ah));
        char *other = NULL;          /* the second buffer, once the thread has made it */
        int64_t buf_off = 0;         /* file offset of buf[0] (raw files) */
        int ahead_ready = 0;         /* buf already holds the next piece (have, eof set) */
        for (;;) {
            if (ahead_ready) {
                ahead_ready = 0;
                goto piece_in_place;
            }
            if (start) {
                memmove(buf, buf + start, (size_t)(have - start));
                have -= start;
                start = 0;
            }
            if (raw_fd >= 0 && have < cap_buf && !eof) {
                const int64_t want = cap_buf - have;
                const int64_t r = cli_pread_parallel(raw_fd, buf + have, want, raw_off, READ_THREADS);
                if (r < 0) {
                    CLI_ERROR("reading %s failed", path);
                    exit(EXIT_FAILURE);
                }
                have += r;
                raw_off += r;
                if (r < want) eof = 1;
            }
            while (raw_fd < 0 && have < cap_buf && !eof) {
                const int64_t want = cap_buf - have > (1 << 30) ? (1 << 30) : cap_buf - have;
                const int r = gzread(fp, buf + have, (unsigned)want);
                if (r < 0) {
                    CLI_ERROR("reading %s failed", path);
                    exit(EXIT_FAILURE);
                }
                have += r;
                if (r < want) eof = 1;
            }
            if (have == 0) break;
        piece_in_place:
            TRACE("piece read");
            if (use_ahead && !eof && start == 0) {        /* the next piece, beside everything below */
                ah.fd = raw_fd;
                ah.n_threads = READ_THREADS;
                ah.cur = buf;
                ah.cur_n = have;
                ah.cur_off = buf_off;
                ah.dst = other;
                ah.dst_cap = cap_other;
                ah.cap = piece;
                ah.started = pthread_create(&ah.th, NULL, fa_ahead_thread, &ah) == 0;
            }
            if (!h) h = cli_accel_open_end();
            TRACE("device open");
            cornetto_asm_t *a = NULL;
            cli_recname_t *r = NULL;
            int64_t n = 0, used = 0;
            int32_t plain = 1;
            if (fasta) {
                cornetto_farec_t *recs = NULL;
                cli_accel_check(h, cornetto_fasta_split(h, buf, have, eof, &recs, &n, &used, &plain, &a), "framing the FASTA records");
                r = (cli_recname_t *)cli_xmalloc(((size_t)n + 1) * sizeof(*r));
                for (int64_t i = 0; i < n; ++i) {
                    r[i].name = buf + recs[i].head + 1;
                    r[i].name_len = recs[i].name_len;
                    r[i].len = recs[i].len;
                }
                cornetto_free(recs);
            } else {
                cornetto_fqrec_t *recs = NULL;
                cli_accel_check(h, cornetto_fastq_split(h, buf, have, eof, 0, &recs, &n, &used, &plain, &a), "framing the FASTQ records");
                r = (cli_recname_t *)cli_xmalloc(((size_t)n + 1) * sizeof(*r));
                for (int64_t i = 0; i < n; ++i) {
                    r[i].name = buf + recs[i].head + 1;
                    r[i].name_len = recs[i].name_len;
                    r[i].len = recs[i].len;
                }
                cornetto_free(recs);
            }
            TRACE("records framed");
            if (n) scan(h, r, n, a, arg);
            TRACE("scanned and printed");
            free(r);
            cornetto_asm_free(h, a);
            start = used;
            if (ah.started) {
                pthread_join(ah.th, NULL);
                ah.started = 0;
                other = ah.dst;
                cap_other = ah.dst_cap;
                if (plain && !eof && ah.pred == used && ah.got > 0) {
                    /* the piece that was read ahead begins where this one stopped: swap the buffers */
                    char *t = buf;
                    buf = other;
                    other = t;
                    const int64_t tc = cap_buf;
                    cap_buf = cap_other;
                    cap_other = tc;
                    buf_off += used;
                    raw_off = buf_off + ah.got;
                    have = ah.got;
                    eof = ah.eof;
                    start = 0;
                    ahead_ready = 1;
                    continue;
                }
                if (ah.got < 0) {
                    CLI_ERROR("reading %s failed", path);
                    exit(EXIT_FAILURE);
                }
            }
            if (!plain || eof) break;              /* not plain from buf + start on / the input is finished */
            buf_off += start;                      /* (the bytes in front of `start` are dropped by the memmove above) */
            if (used == 0 && have == cap_buf) { /* one record larger than the buffer: a larger one, as long as the index allows */
                if (cap_buf >= piece_max) break;
                int64_t bigger = cap_buf < piece ? piece : cap_buf * 2;        /* (the small first buffer: to a full piece first) */
                if (bigger > piece_max) bigger = piece_max;
                char *nb = (char *)cornetto_pinned_alloc((size_t)bigger);
                if (!nb) break;
                memcpy(nb, buf, (size_t)have);
                cornetto_pinned_free(buf);
                buf = nb;
                cap_buf = bigger;
                if (bigger > piece) piece = bigger;
                if (other) cornetto_pinned_free(other);   /* (the read-ahead buffer is made again at the new size) */
                other = NULL;
                cap_other = 0;
            }
        }
        if (!h) h = cli_accel_open_end();
        if (raw_fd >= 0) { /* the sequential reader goes on in the gz stream where the raw reads stopped */
            close(raw_fd);
            gzseek(fp, (z_off_t)raw_off, SEEK_SET);
        }
    }
    if (start < have || !eof) { /* the rest (or all of it) through the sequential reader */
        batch_scan_t bs = {scan, arg};
        cli_fastx_t *fx = cli_fastx_open_prefixed(fp, buf + start, (size_t)(have - start));
        batches_of(fx, h, scan_batch, &bs);
    } else {
        gzclose(fp);
    }
    /* the pinned piece and the device handle are left to the end of the process (main.c leaves with _exit right after
     * the sub-command): unpinning a 1 GB piece and closing the handle take about 0.1 s */
    (void)h;
    TRACE("done");
}

/* the host path (--accel=no / CORNETTO_ACCEL=no): the sequential reader, one record at a time, printed as it is scanned */
static void host_stream(const char *path, int must_open, const multi_what_t *what)
{
    cli_fastx_t *fx = cli_fastx_open(path);
    if (!fx) {
        if (must_open) {
            CLI_ERROR("Could not to open file %s: %s", path, strerror(errno)); /* F_CHK, src/error.h:114-119: its words */
            exit(EXIT_FAILURE);
        }
        return;
    }
    cli_str_t name = {0, 0, 0}, comment = {0, 0, 0}, seq = {0, 0, 0}, qual = {0, 0, 0};
    cornetto_hit_t *hits = NULL;
    cornetto_ivl_t *ivls = NULL;
    int64_t n = 0, cap = 0, l;
    while ((l = cli_fastx_read(fx, &name, &comment, &seq, &qual)) >= 0) {
        if (l > 0x7fffffffLL) {
            CLI_ERROR("record %s has %lld bases; the reference's reader is limited to 2^31-1 (src/kseq.h:185)", name.s, (long long)l);
            exit(EXIT_FAILURE);
        }
        const uint8_t *s = (const uint8_t *)(seq.s ? seq.s : "");
        n = 0;
        if (what->kind == 0) {
            cli_host_telofind(s, l, what->motif, 0, &hits, &n, &cap);
            for (int64_t i = 0; i < n; ++i) print_hit(name.s, name.l, l, &hits[i]);
        } else {
            if (cli_host_sdust(s, l, what->T, what->W, 0, &ivls, &n, &cap) != 0) {
                CLI_ERROR("sdust: -w %d / -t %d outside 3..1026 / 0..2^20", what->W, what->T);
                exit(EXIT_FAILURE);
            }
            for (int64_t i = 0; i < n; ++i) print_ivl(name.s, name.l, &ivls[i]);
        }
    }
    cli_out_flush();
    free(hits);
    free(ivls);
    free(name.s);
    free(comment.s);
    free(seq.s);
    free(qual.s);
    cli_fastx_close(fx);
}

int find_telomere_main(int argc, char *argv[])
{
    if (argc < 2) { /* src/find_telomere.c:84-88 */
        fprintf(stderr, "Error: invalid number of parameters\n");
        fprintf(stderr, "Usage: find <input fasta> [optional sequence to search for, default is vertebrate TTAGGG]\n");
        exit(EXIT_FAILURE);
    }
    const char *motif = argc >= 3 ? argv[2] : "TTAGGG";
    if (motif[0] == 0) {
        CLI_ERROR("%s", "empty search sequence");
        exit(EXIT_FAILURE);
    }
    if (cli_host_mode()) {
        const multi_what_t what = {0, motif, 0, 0};
        host_stream(argv[1], 1, &what);
        return EXIT_SUCCESS;
    }
    int devs[CLI_MAX_DEV];
    const int n_dev = cli_device_list(devs);
    if (n_dev >= 2) {
        const multi_what_t what = {0, motif, 0, 0};
        multi_stream(argv[1], 1, &what, devs, n_dev);
        return EXIT_SUCCESS;
    }
    if (n_dev == 1) {
        char one[32];
        snprintf(one, sizeof(one), "%d", devs[0]);
        setenv("CORNETTO_DEVICE", one, 1);
    }
    stream_records(argv[1], 1, telofind_scan, (void *)motif);
    return EXIT_SUCCESS;
}

/* ---------------------------------------------------------------- sdust */
typedef struct {
    int W, T;
} sdust_opt_t;

static void sdust_scan(cornetto_accel_t *h, const cli_recname_t *r, int64_t n_rec, const cornetto_asm_t *a, void *arg)
{
    (void)n_rec;
    const sdust_opt_t *o = (const sdust_opt_t *)arg;
    cornetto_ivl_t *iv = NULL;
    int64_t n = 0;
    cli_accel_check(h, cornetto_sdust_asm(h, a, o->T, o->W, &iv, &n), "sdust");
    for (int64_t i = 0; i < n; ++i) print_ivl(r[iv[i].ctg].name, (size_t)r[iv[i].ctg].name_len, &iv[i]);
    cli_out_flush();
    cornetto_free(iv);
}

int sdust_main(int argc, char *argv[])
{
    sdust_opt_t o = {64, 20}; /* src/sdust/sdust.c:183 */
    int c;
    cli_dash_is_stdin = 1;    /* src/sdust/sdust.c:194 */
    /* ketopt(..., permute=1, "w:t:") of the reference == POSIX getopt with GNU permutation */
    optind = 1;
    while ((c = getopt(argc, argv, "w:t:")) >= 0) {
        if (c == 'w') o.W = atoi(optarg);
        else if (c == 't') o.T = atoi(optarg);
    }
    if (optind == argc) {
        fprintf(stderr, "Usage: sdust [-w %d] [-t %d] <in.fa>\n", o.W, o.T);
        exit(1);
    }
    if (cli_host_mode()) {
        const multi_what_t what = {1, NULL, o.T, o.W};
        host_stream(argv[optind], 0, &what);
        return 0;
    }
    int devs[CLI_MAX_DEV];
    const int n_dev = cli_device_list(devs);
    if (n_dev >= 2) {
        const multi_what_t what = {1, NULL, o.T, o.W};
        multi_stream(argv[optind], 0, &what, devs, n_dev);
        return 0;
    }
    if (n_dev == 1) {
        char one[32];
        snprintf(one, sizeof(one), "%d", devs[0]);
        setenv("CORNETTO_DEVICE", one, 1);
    }
    stream_records(argv[optind], 0, sdust_scan, &o);
    return 0;
}

/* ---------------------------------------------------------------- fa2bed */
static const struct option help_only[] = {{"verbose", required_argument, 0, 'v'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};

int assbed_main(int argc, char *argv[])
{
    FILE *fp_help = stderr;
    int c, li = 0;
    optind = 1;
    while ((c = getopt_long(argc, argv, "h", help_only, &li)) >= 0)
        if (c == 'h') fp_help = stdout;
    if (argc - optind != 1 || fp_help == stdout) {
        fprintf(fp_help, "Usage: cornetto asmbed <assembly.fasta> \n");
        fprintf(fp_help, "   -h                         help\n");
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }
    cli_fastx_t *fx = cli_fastx_open(argv[optind]);
    if (!fx) {
        CLI_ERROR("Could not to open file %s: %s", argv[optind], strerror(errno)); /* F_CHK, src/error.h:114-119: its words */
        exit(EXIT_FAILURE);
    }
    cli_str_t name = {0, 0, 0}, comment = {0, 0, 0}, seq = {0, 0, 0}, qual = {0, 0, 0};
    int64_t l;
    while ((l = cli_fastx_read(fx, &name, &comment, &seq, &qual)) >= 0) fprintf(stdout, "%s\t%d\t%d\n", name.s, 0, (int)l);   /* src/assbed.c:99 */
    cli_fastx_close(fx);
    return 0;
}

/* ---------------------------------------------------------------- seq */
int seq_main(int argc, char *argv[])
{
    static const struct option lo[] = {{"verbose", required_argument, 0, 'v'}, {"min-len", required_argument, 0, 'm'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    FILE *fp_help = stderr;
    int min_len = 30000; /* src/seq.c:63 */
    int c, li = 0;
    optind = 1;
    while ((c = getopt_long(argc, argv, "hm:", lo, &li)) >= 0) {
        if (c == 'h') {
            fp_help = stdout;
        } else if (c == 'm') {
            min_len = atoi(optarg);
            if (min_len < 0) {
                fprintf(stderr, "Error: min-len must be a positive integer\n");
                exit(EXIT_FAILURE);
            }
        } else {
            fprintf(stderr, "Unknown option: %s\n", argv[optind - 1]);
            exit(EXIT_FAILURE);
        }
    }
    if (argc - optind != 1 || fp_help == stdout) {
        fprintf(fp_help, "Usage: cornetto seq <reads.fastq> \n");
        fprintf(fp_help, "   -m INT                     min length [%d]\n", 30000);
        fprintf(fp_help, "   -h                         help\n");
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }
    cli_fastx_t *fx = cli_fastx_open(argv[optind]);
    if (!fx) {
        CLI_ERROR("Could not to open file %s: %s", argv[optind], strerror(errno)); /* F_CHK, src/error.h:114-119: its words */
        exit(EXIT_FAILURE);
    }
    cli_str_t name = {0, 0, 0}, comment = {0, 0, 0}, seq = {0, 0, 0}, qual = {0, 0, 0};
    uint64_t before = 0, after = 0, before_n = 0, after_n = 0;
    int64_t l;
    while ((l = cli_fastx_read(fx, &name, &comment, &seq, &qual)) >= 0) {
        before += (uint64_t)l;
        before_n++;
        if (l >= min_len) { /* src/seq.c:120-129: name, TAB, comment */
            after += (uint64_t)l;
            after_n++;
            printf("@%s", name.s);
            if (comment.l) printf("\t%s", comment.s);
            printf("\n%s\n+\n%s\n", seq.s, qual.s);
        }
    }
    fprintf(stderr, "total reads: %lu\t%lu bases\t%.2f Gbases\n", (unsigned long)before_n, (unsigned long)before, before / 1e9);
    fprintf(stderr, "reads >= %d: %lu\t%lu bases\t%.2f Gbases\n", min_len, (unsigned long)after_n, (unsigned long)after, after / 1e9);
    cli_fastx_close(fx);
    return 0;
}

/* util.c — timers, allocation helpers, accelerator open for the cornetto CLI */
#include "cli.h"

#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>

int cli_log_level = 4;

double cli_realtime(void) /* src/misc.c:48-52 */
{
    struct timeval tp;
    gettimeofday(&tp, NULL);
    return tp.tv_sec + tp.tv_usec * 1e-6;
}

double cli_cputime(void) /* src/misc.c:54-59 */
{
    struct rusage r;
    getrusage(RUSAGE_SELF, &r);
    return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}

long cli_peakrss(void) /* src/misc.c:61-70 */
{
    struct rusage r;
    getrusage(RUSAGE_SELF, &r);
    return r.ru_maxrss * 1024;
}

void *cli_xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) {
        CLI_ERROR("Failed to allocate %zu bytes", n);
        exit(EXIT_FAILURE);
    }
    return p;
}

void *cli_xrealloc(void *p, size_t n)
{
    void *q = realloc(p, n ? n : 1);
    if (!q) {
        CLI_ERROR("Failed to allocate %zu bytes", n);
        exit(EXIT_FAILURE);
    }
    return q;
}

char *cli_xstrdup(const char *s)
{
    size_t n = strlen(s) + 1;
    char *d = (char *)cli_xmalloc(n);
    memcpy(d, s, n);
    return d;
}

static char out_storage[CLI_OUT_CAP + 64];
char *cli_out_buf = out_storage;
size_t cli_out_n = 0;

void cli_out_flush(void)
{
    if (cli_out_n) fwrite(cli_out_buf, 1, cli_out_n, stdout);
    cli_out_n = 0;
}

cornetto_accel_t *cli_accel_open(void)
{
    const char *d = getenv("CORNETTO_DEVICE");
    cornetto_accel_t *h = NULL;
    int rc = cornetto_accel_open(&h, d ? atoi(d) : 0, NULL);
    if (rc != CORNETTO_OK) {
        CLI_ERROR("cannot open HIP device %d: %s. The scans run on an AMD GPU; the sequential host path is a choice (--accel=no / CORNETTO_ACCEL=no), never a fallback.",
                  d ? atoi(d) : 0, cornetto_accel_strerror(rc));
        exit(EXIT_FAILURE);
    }
    return h;
}

/* The device is opened on a helper thread while the input is being read (HIP initialisation takes a few hundred
 * milliseconds); cli_accel_open_end() joins it and behaves like cli_accel_open(). */
#include <pthread.h>
static struct {
    pthread_t th;
    int started, rc, dev, ready, warm;
    cornetto_accel_t *h;
    pthread_mutex_t mu;
    pthread_cond_t cv;
} g_open = {.mu = PTHREAD_MUTEX_INITIALIZER, .cv = PTHREAD_COND_INITIALIZER};

static void *open_thread(void *arg)
{
    (void)arg;
    cornetto_accel_t *h = NULL;
    const int rc = cornetto_accel_open(&h, g_open.dev, NULL);
    pthread_mutex_lock(&g_open.mu);
    g_open.rc = rc;
    g_open.h = h;
    g_open.ready = 1;
    pthread_cond_broadcast(&g_open.cv);
    pthread_mutex_unlock(&g_open.mu);
    if (rc == CORNETTO_OK && g_open.warm) {
        /* what the first use of the runtime costs in a process (code objects, copy engines, the first pinned pools), paid here — on a handle of its
         * own, beside the caller's reading and copying — instead of in its first scan (cornetto_accel_warm) */
        cornetto_accel_t *w = NULL;
        if (cornetto_accel_open(&w, g_open.dev, NULL) == CORNETTO_OK) {
            (void)cornetto_accel_warm(w, g_open.warm);
            cornetto_accel_close(w);
        }
    }
    return NULL;
}

/* a large input is on its way: warm the named entry points up (CORNETTO_WARM_*) behind the open; before cli_accel_open_begin() */
void cli_accel_warm_hint(int what) { g_open.warm = what; }

void cli_accel_open_begin(void)
{
    if (g_open.started) return;
    const char *d = getenv("CORNETTO_DEVICE");
    g_open.dev = d ? atoi(d) : 0;
    g_open.h = NULL;
    g_open.ready = 0;
    if (pthread_create(&g_open.th, NULL, open_thread, NULL) == 0) g_open.started = 1;
}

cornetto_accel_t *cli_accel_open_end(void)
{
    if (!g_open.started) return cli_accel_open();
    pthread_mutex_lock(&g_open.mu);
    while (!g_open.ready) pthread_cond_wait(&g_open.cv, &g_open.mu);
    pthread_mutex_unlock(&g_open.mu);
    if (!g_open.warm) pthread_join(g_open.th, NULL);
    else pthread_detach(g_open.th);                 /* (the warm-up goes on beside the caller; the process leaves with _exit) */
    g_open.started = 0;
    if (g_open.rc != CORNETTO_OK) {
        CLI_ERROR("cannot open HIP device %d: %s. The scans run on an AMD GPU; the sequential host path is a choice (--accel=no / CORNETTO_ACCEL=no), never a fallback.",
                  g_open.dev, cornetto_accel_strerror(g_open.rc));
        exit(EXIT_FAILURE);
    }
    return g_open.h;
}

/* the helper thread's handle is not needed after all (an input without records) */
void cli_accel_open_cancel(void)
{
    if (!g_open.started) return;
    pthread_join(g_open.th, NULL);
    g_open.started = 0;
    if (g_open.rc == CORNETTO_OK && g_open.h) cornetto_accel_close(g_open.h);
}

/* the reference ends here with SIGABRT (an assert of get_regs(), src/boringbits_main.c:353,368): glibc's one line on stderr, nothing on
 * stdout (no output buffer is flushed), exit status 134 in a shell — the same here */
void cli_ref_abort(const char *msg)
{
    fprintf(stderr, "cornetto: %s\n", msg);
    fflush(stderr);
    abort();
}

void cli_accel_check(cornetto_accel_t *h, int rc, const char *what)
{
    if (rc == CORNETTO_OK) return;
    if (rc == CORNETTO_E_ASSERT) cli_ref_abort(cornetto_accel_last_error(h));
    CLI_ERROR("%s failed: %s (%s)", what, cornetto_accel_last_error(h), cornetto_accel_strerror(rc));
    exit(EXIT_FAILURE);
}

void cli_batch_push(cli_batch_t *b, const char *name, const char *seq, int64_t len)
{
    if (b->n == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 256;
        b->names = (char **)cli_xrealloc(b->names, (size_t)b->cap * sizeof(char *));
        b->seqs = (uint8_t **)cli_xrealloc(b->seqs, (size_t)b->cap * sizeof(uint8_t *));
        b->lens = (int64_t *)cli_xrealloc(b->lens, (size_t)b->cap * sizeof(int64_t));
    }
    b->names[b->n] = cli_xstrdup(name);
    b->seqs[b->n] = (uint8_t *)cli_xmalloc((size_t)len + 1);
    memcpy(b->seqs[b->n], seq, (size_t)len);
    b->seqs[b->n][len] = 0;
    b->lens[b->n] = len;
    b->bases += len;
    b->n++;
}

/* like cli_batch_push(), but a long sequence is taken over from the reader's buffer instead of being copied */
void cli_batch_take(cli_batch_t *b, const char *name, cli_str_t *seq)
{
    if (seq->l < (1u << 16) || !seq->s) {
        cli_batch_push(b, name, seq->s ? seq->s : "", (int64_t)seq->l);
        return;
    }
    cli_batch_push(b, name, "", 0);
    free(b->seqs[b->n - 1]);
    b->seqs[b->n - 1] = (uint8_t *)seq->s;
    b->lens[b->n - 1] = (int64_t)seq->l;
    b->bases += (int64_t)seq->l;
    seq->s = NULL;
    seq->l = seq->m = 0;
}

void cli_batch_clear(cli_batch_t *b)
{
    for (int32_t i = 0; i < b->n; ++i) {
        free(b->names[i]);
        free(b->seqs[i]);
    }
    b->n = 0;
    b->bases = 0;
}

int64_t cli_batch_limit(void)
{
    const char *s = getenv("CORNETTO_BATCH_BASES");
    int64_t v = s ? atoll(s) : 0;
    return v > 0 ? v : 4000000000LL;
}

/* the ordinals of $CORNETTO_DEVICES ("0,1,2": several GPUs of one node, a device may be named twice); 0 when it is not set
 * (one device: $CORNETTO_DEVICE) */
int cli_device_list(int *devs)
{
    const char *s = getenv("CORNETTO_DEVICES");
    int n = 0;
    if (!s || !*s) return 0;
    while (*s) {
        char *end = NULL;
        const long v = strtol(s, &end, 10);
        if (end == s || v < 0 || n == CLI_MAX_DEV) {
            CLI_ERROR("CORNETTO_DEVICES=%s: a comma-separated list of at most %d device ordinals is expected", getenv("CORNETTO_DEVICES"), CLI_MAX_DEV);
            exit(EXIT_FAILURE);
        }
        devs[n++] = (int)v;
        s = end;
        while (*s == ',' || *s == ' ') ++s;
    }
    return n;
}

/* order[k] = index of the record with the k-th largest length (ties: input order).  Lengths are below 2^31 (the reference keeps
 * them in int), so (2^31 - 1 - length) << 32 | index sorts the way LPT wants with one qsort over 64-bit keys: n log n whatever
 * the input looks like (an insertion sort here took tens of seconds on a batch of 400 000 reads of random lengths). */
static int cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

void cli_order_by_length_desc(const int64_t *lens64, const int32_t *lens32, int32_t n, int32_t *order)
{
    uint64_t *key = (uint64_t *)cli_xmalloc(((size_t)n + 1) * sizeof(*key));
    for (int32_t i = 0; i < n; ++i) {
        const int64_t len = lens64 ? lens64[i] : (int64_t)lens32[i];
        key[i] = (uint64_t)(0x7fffffffll - (len < 0 ? 0 : len > 0x7fffffffll ? 0x7fffffffll : len)) << 32 | (uint32_t)i;
    }
    qsort(key, (size_t)n, sizeof(*key), cmp_u64);
    for (int32_t i = 0; i < n; ++i) order[i] = (int32_t)(key[i] & 0xFFFFFFFFu);
    free(key);
}

/* ---------------- pread() with a few threads ----------------
 * an uncompressed regular file is read straight into a pinned piece: one thread copies from the page cache at 5-8 GB/s, which is
 * the largest share of the wall time of a 3 GB assembly or of a pair of per-base bedgraphs */
#include <errno.h>
#include <sched.h>
#include <unistd.h>
#define CLI_PREAD_MAX 64
typedef struct {
    int fd;
    char *dst;
    int64_t off, want, got;
    int failed; /* a pread() returned < 0 (EIO, ESTALE ...): not an end of file */
} pread_job_t;

/* Bytes [off, off + want) of a regular file into dst: -> bytes read (short at the end of the file), -1 on a read error.
 * A file on tmpfs (/dev/shm: where a pipeline leaves what its next step reads) is copied out of a shared mapping instead of with pread(): the first
 * read() of a page that was just written marks it accessed, one page at a time under the LRU lock, and eight or thirty-two reader threads share 12-14 GB/s
 * of such first reads — a memcpy out of the mapping does 75 GB/s on the same first pass (tools/ubench/pread_rate, profiles/r06_pread_rate.txt).  The mapping
 * of a file is made once and kept.  Files elsewhere keep pread(): an I/O error there is a return value, not a SIGBUS. */
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
static struct {
    pthread_mutex_t mu;
    int n;
    struct { dev_t dev; ino_t ino; const char *map; int64_t size; } e[8];
} g_maps = {.mu = PTHREAD_MUTEX_INITIALIZER};

static const char *tmpfs_mapping(int fd, int64_t *size)
{
    struct stat st;
    struct statfs fs;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return NULL;
    pthread_mutex_lock(&g_maps.mu);
    for (int i = 0; i < g_maps.n; ++i)
        if (g_maps.e[i].dev == st.st_dev && g_maps.e[i].ino == st.st_ino && g_maps.e[i].size == (int64_t)st.st_size) {
            const char *m = g_maps.e[i].map;
            *size = g_maps.e[i].size;
            pthread_mutex_unlock(&g_maps.mu);
            return m;
        }
    const char *m = NULL;
    const char *off_env = getenv("CORNETTO_CLI_MMAP");
    if (g_maps.n < 8 && !(off_env && !atoi(off_env)) && fstatfs(fd, &fs) == 0 && (unsigned long)fs.f_type == 0x01021994UL /* TMPFS_MAGIC */) {
        void *p = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
        if (p != MAP_FAILED) {
            m = (const char *)p;
            g_maps.e[g_maps.n].dev = st.st_dev;
            g_maps.e[g_maps.n].ino = st.st_ino;
            g_maps.e[g_maps.n].map = m;
            g_maps.e[g_maps.n].size = (int64_t)st.st_size;
            ++g_maps.n;
            *size = (int64_t)st.st_size;
        }
    }
    pthread_mutex_unlock(&g_maps.mu);
    return m;
}

int64_t cli_read_at(int fd, char *dst, int64_t want, int64_t off, int *failed)
{
    if (failed) *failed = 0;
    if (want <= 0) return 0;
    int64_t size = 0;
    const char *map = tmpfs_mapping(fd, &size);
    if (map) {
        if (off >= size) return 0;
        const int64_t n = want < size - off ? want : size - off;
        memcpy(dst, map + off, (size_t)n);
        return n;
    }
    int64_t got = 0;
    while (got < want) {
        const ssize_t r = pread(fd, dst + got, (size_t)(want - got), (off_t)(off + got));
        if (r < 0 && errno == EINTR) continue;
        if (r < 0 && failed) *failed = 1;
        if (r <= 0) break; /* r == 0: end of the file */
        got += r;
    }
    return got;
}

static void *pread_thread(void *p)
{
    pread_job_t *j = (pread_job_t *)p;
    const double t0 = getenv("CORNETTO_CLI_TRACE_READS") ? cli_realtime() : 0.0;
    j->got = cli_read_at(j->fd, j->dst, j->want, j->off, &j->failed);
    if (t0 > 0.0) {
        cpu_set_t cs;
        CPU_ZERO(&cs);
        (void)sched_getaffinity(0, sizeof(cs), &cs);
        fprintf(stderr, "[cli trace] pread job: %lld bytes at %lld in %.2f ms (from %.3f) on cpu %d of %d allowed\n", (long long)j->got, (long long)j->off,
                (cli_realtime() - t0) * 1e3, t0, sched_getcpu(), CPU_COUNT(&cs));
    }
    return NULL;
}

/* bytes read at file offset `off` (short only at the end of the file); -1 when a read failed */
int64_t cli_pread_parallel(int fd, char *dst, int64_t want, int64_t off, int n_threads)
{
    pread_job_t job[CLI_PREAD_MAX];
    pthread_t th[CLI_PREAD_MAX];
    int started[CLI_PREAD_MAX];
    if (n_threads < 1) n_threads = 1;
    if (n_threads > CLI_PREAD_MAX) n_threads = CLI_PREAD_MAX;
    int64_t part = ((want + n_threads - 1) / n_threads + 4095) & ~4095LL;
    if (part < (1 << 20)) part = 1 << 20; /* small reads: fewer threads */
    int nj = 0;
    for (int64_t o = 0; o < want; o += part, ++nj) {
        job[nj].fd = fd;
        job[nj].dst = dst + o;
        job[nj].off = off + o;
        job[nj].want = want - o < part ? want - o : part;
        started[nj] = nj > 0 && pthread_create(&th[nj], NULL, pread_thread, &job[nj]) == 0;
    }
    int64_t total = 0;
    int open_end = 1;
    for (int i = 0; i < nj; ++i) {
        if (i == 0 || !started[i]) pread_thread(&job[i]);
    }
    for (int i = 0; i < nj; ++i) {
        if (started[i]) pthread_join(th[i], NULL);
        if (open_end) total += job[i].got;
        if (open_end && job[i].failed) return -1; /* an error inside the bytes that count, not behind the end of the file */
        if (job[i].got < job[i].want) open_end = 0;
    }
    return total;
}

// common.hpp — handle, error plumbing, event timing and device-buffer helpers shared by the HIP sources
// of libcornetto_hip.so (gfx950 only).  Not part of the C ABI (include/cornetto_accel.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <string>
#include <utility>
#include <vector>

#include "../../include/cornetto_accel.h"

// wave priority of the short streaming kernels that run beside the resident sdust waves (cov_blocks, cov_windows, tf_scan, tw_scan)
#ifndef CN_STREAM_PRIO
#define CN_STREAM_PRIO 3
#endif

inline std::atomic<uint64_t> cn_uid_counter{1};    // resident objects are told apart by number, not by address (addresses come back)

struct cornetto_accel {
    int device = 0;
    // telofind: what the handle's small device blocks hold from the last call (no upload when the next call wants the same)
    std::string tf_lut_key;
    const void *tf_lut_ptr = nullptr;
    uint64_t tf_bm_uid = 0;              // telofind: the assembly whose padding words the mark bitmap block (tf_bm_ptr, tf_bm_words) holds as zeros
    const void *tf_bm_ptr = nullptr;
    size_t tf_bm_words = 0;

    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t stream2 = nullptr;     // sdust: the dense kernel runs here, beside the main kernel on `stream`
    hipEvent_t ev1 = nullptr, ev2 = nullptr;
    char err[512] = {0};
    // event timing of the kernels of the current / most recent compute call
    struct Rec {
        const char *name;
        hipEvent_t a, b;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    std::vector<std::pair<const char *, float>> last;
    // persistent workspaces (grown on demand, released at close): no hipMalloc/hipFree on the hot path
    struct Ws {
        void *p = nullptr;
        size_t bytes = 0;
    };
    Ws dev[64];
    Ws pin[32];
    int sd_slots = 0;   // sdust: waves the device holds at once (occupancy query, cached)
    int sd_cus = 0;
    int sift_per_cu = 0; // sdust sift: workgroups per CU by the occupancy query (cached)
    int sift_per_cu_default = -1;   //   ... of which build of the kernel (1: buffer size a literal, 0: run-time size)
    // cornetto_accel_set_lazy(): the large result copies of a call go out on a stream of their own and the call returns when its kernels are
    // through; cornetto_accel_wait() before the results are read
    int lazy = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_cp = nullptr;
    bool copies_pending = false;
    volatile int boost = 0;
    volatile unsigned long long launch_seq = 0;   // counts the launches of resident sdust waves (cornetto_accel_launch_count)  // cornetto_accel_boost(): the other users of the device are through (set from another host thread)
    hipEvent_t ev3 = nullptr;
    int share = 100;    // percent of every CU the resident sdust kernel may take (cornetto_accel_set_share)
    int sd_stats = 0;   // sdust: run the statistics build of the kernel (cornetto_accel_sdust_stats)
    // cornetto_sdust_asm_begin() -> _end(): 0 nothing pending, 1 the whole call is queued on the stream (one go: sized by the last call's counts,
    // checked by _end), 2 _begin had to run the call to its end (the result waits here)
    struct SdPend {
        int state = 0;
        const void *a = nullptr;
        int32_t T = 0, W = 0;
        void *of = nullptr;                   // the result array the queued copy fills
        size_t m_cap = 0, n_cap = 0, cap = 0;
        int64_t key = 0, n_done = 0;
        bool walk_pending = false;
    } sd_pend;
    unsigned long long sd_last[256] = {0};   // its counters from the most recent such run
    // scan.hpp: the single-pass scans keep their tile states in WS_SCAN; a state counts only with the epoch of its call (no clearing
    // between calls), the tiles of a call take their numbers from a ticket counter that is never reset
    uint32_t scan_epoch = 0, scan_tickets = 0;
    uint32_t st_epoch = 0, st_tickets = 0;          // the same for the single-pass interval merge (ivlmerge.hpp, WS_STITCH)
    int timing = 2;     // event pairs around: 2 every kernel launch, 1 the three streaming / scanning main kernels only, 0 none (cornetto_accel_set_timing)
};

// device workspace slot `slot` with at least `bytes` bytes (contents undefined); nullptr on failure
static inline void *cn_ws(cornetto_accel_t *h, int slot, size_t bytes)
{
    cornetto_accel::Ws &w = h->dev[slot];
    if (bytes == 0) bytes = 16;
    if (w.bytes >= bytes) return w.p;
#ifdef CN_WS_ASYNC
    // (experiment, DESIGN 8 round 6: the stream-ordered allocator instead of hipMalloc, which waits for the kernels in flight on the device)
    if (w.p) (void)hipFreeAsync(w.p, h->stream);
    w.p = nullptr;
    w.bytes = 0;
    const size_t want = bytes + bytes / 8;
    if (hipMallocAsync(&w.p, want, h->stream) == hipSuccess) w.bytes = want;
    else w.p = nullptr;
    return w.p;
#else
    if (w.p) (void)hipFree(w.p);
    w.p = nullptr;
    w.bytes = 0;
    const size_t want = bytes + bytes / 8;     // a little head room so that slowly growing inputs do not re-allocate
    if (hipMalloc(&w.p, want) == hipSuccess) w.bytes = want;
    else if (hipMalloc(&w.p, bytes) == hipSuccess) w.bytes = bytes;
    else w.p = nullptr;
    return w.p;
#endif
}

// the tables of a resident object (freed with hipFree when the object goes)
static inline hipError_t cn_obj_malloc(cornetto_accel_t *h, void **p, size_t bytes)
{
#ifdef CN_WS_ASYNC
    return hipMallocAsync(p, bytes, h->stream);
#else
    (void)h;
    return hipMalloc(p, bytes);
#endif
}

// pinned host staging slot (for asynchronous device-to-host copies at full PCIe rate)
static inline void *cn_pin(cornetto_accel_t *h, int slot, size_t bytes)
{
    cornetto_accel::Ws &w = h->pin[slot];
    if (bytes == 0) bytes = 16;
    if (w.bytes >= bytes) return w.p;
    if (w.p) (void)hipHostFree(w.p);
    w.p = nullptr;
    w.bytes = 0;
    // (at least 64 KB: hipHostMalloc / hipHostFree wait for the kernels in flight on the device — a slot that grows from the few bytes of a small
    // first input to the few hundred of an assembly stalled the first scan of the assembly behind the other stream's kernel)
    const size_t want = std::max<size_t>(bytes + bytes / 8, (size_t)64 << 10);
    if (hipHostMalloc(&w.p, want, hipHostMallocDefault) == hipSuccess) w.bytes = want;
    else w.p = nullptr;
    return w.p;
}

enum {   // device workspace slots
    WS_TF_LUT, WS_TF_CNT, WS_TF_TB, WS_TF_TC, WS_TF_L0, WS_TF_L1, WS_TF_L2, WS_TF_L3, WS_TF_BITMAP,
    WS_TF_ROFF, WS_TF_RUNS, WS_TF_NRUNS,
    WS_TW_BOFF, WS_TW_TILES, WS_TW_OUT, WS_TW_CNT, WS_TW_HITS, WS_TW_LEN, WS_TW_BITMAP,
    WS_SD_OUT, WS_SD_CNT, WS_SD_OFF, WS_SD_DST, WS_SD_STATS, WS_SD_PERM,
    WS_CB_T32, WS_CB_T64, WS_CB_GRAND,
    WS_CW_REGS, WS_CW_SEL, WS_CW_CNT, WS_CW_TRES, WS_CW_CF,
    WS_TF_HITS,
    WS_BG_TEXT_A, WS_BG_TEXT_B, WS_BG_TOK_A, WS_BG_TOK_B, WS_BG_CNT_A, WS_BG_CNT_B, WS_BG_SMALL, WS_BG_BRK,
    WS_TB, WS_TB_SMALL, WS_TB_OUT, WS_CW_MERGE, WS_IVL_MERGE,
    WS_FQ_TEXT, WS_FQ_CNT, WS_FQ_NL, WS_FQ_RECS, WS_FQ_ENDS, WS_FQ_SRC,
    WS_SCAN, WS_STITCH,
    WS_COUNT
};
static_assert(WS_COUNT <= 64, "cornetto_accel::dev has 64 slots");
enum {   // pinned host slots
    PIN_A, PIN_B, PIN_C, PIN_D, PIN_E, PIN_F, PIN_SMALL, PIN_TW, PIN_CW, PIN_STEP
};

static inline int cn_fail(cornetto_accel_t *h, int status, const char *fmt, ...)
{
    if (h) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(h->err, sizeof(h->err), fmt, ap);
        va_end(ap);
    }
    return status;
}

#define CN_HIP(h, call)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return cn_fail((h), CORNETTO_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,                \
                           hipGetErrorString(e_));                                                           \
    } while (0)

#define CN_TRY(expr)                                                                                         \
    do {                                                                                                     \
        int rc_ = (expr);                                                                                    \
        if (rc_ != CORNETTO_OK) return rc_;                                                                  \
    } while (0)

// device -> host copy of a call's RESULT array.  Default: on the handle's stream (the call's final synchronisation covers it).  Lazy handles:
// on the copy stream, behind everything the handle's stream holds so far — the caller's next kernels run beside it.
static inline hipError_t cn_result_d2h(cornetto_accel_t *h, void *dst, const void *src, size_t bytes)
{
    if (!h->lazy) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream);
    hipError_t e = hipSuccess;
    if (!h->copy_stream) {
        e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_cp, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    e = hipEventRecord(h->ev_cp, h->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->copy_stream, h->ev_cp, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->copy_stream);
    h->copies_pending = true;
    return e;
}

// before a result array with a copy in flight is given up on an error path
static inline void cn_result_quiesce(cornetto_accel_t *h)
{
    if (h->copies_pending && h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    h->copies_pending = false;
}

// ---- event timing ---------------------------------------------------------------------------------
static inline hipEvent_t cn_event(cornetto_accel_t *h)
{
    if (!h->pool.empty()) {
        hipEvent_t e = h->pool.back();
        h->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

static inline void cn_timing_begin(cornetto_accel_t *h)
{
    for (auto &r : h->recs) {
        h->pool.push_back(r.a);
        h->pool.push_back(r.b);
    }
    h->recs.clear();
    h->last.clear();
}

// resolve the recorded events into milliseconds; call after the stream has been synchronised
static inline void cn_timing_end(cornetto_accel_t *h)
{
    h->last.clear();
    for (auto &r : h->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = -1.f;
        h->last.emplace_back(r.name, ms);
        h->pool.push_back(r.a);
        h->pool.push_back(r.b);
    }
    h->recs.clear();
}

static inline bool cn_timed(const cornetto_accel_t *h, const char *name)
{
    if (h->timing >= 2) return true;
    if (h->timing <= 0) return false;
    return !strcmp(name, "sdust_kernel") || !strcmp(name, "cov_blocks") || !strcmp(name, "tf_scan");
}

// Launch `...` (a kernel<<<>>> expression on h->stream) bracketed by two events recorded under `name`.  Every event
// record is a packet of its own on the queue (tens of microseconds each beside a busy second stream): callers that
// do not read cornetto_accel_last_timing() switch them off (cornetto_accel_set_timing).
#define CN_LAUNCH(h, name, ...)                                                                              \
    do {                                                                                                     \
        if (cn_timed((h), (name))) {                                                                         \
            cornetto_accel::Rec r_{(name), cn_event(h), cn_event(h)};                                        \
            CN_HIP(h, hipEventRecord(r_.a, (h)->stream));                                                    \
            __VA_ARGS__;                                                                                     \
            CN_HIP(h, hipGetLastError());                                                                    \
            CN_HIP(h, hipEventRecord(r_.b, (h)->stream));                                                    \
            (h)->recs.push_back(r_);                                                                         \
        } else {                                                                                             \
            __VA_ARGS__;                                                                                     \
            CN_HIP(h, hipGetLastError());                                                                    \
        }                                                                                                    \
    } while (0)

// ---- RAII device buffer ----------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t n)
    {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <class T>
    T *as() const
    {
        return reinterpret_cast<T *>(p);
    }
};

// ---- resident data sets -----------------------------------------------------------------------------
// tiles.hpp: the prefix a table of pieces is filled from — first[c] = index of sequence c's first piece, n + 1 numbers — on the host and the device
struct CnPrefix {
    std::vector<int64_t> host;
    int64_t *dev = nullptr;
    size_t cap = 0;
    void release() { if (dev) (void)hipFree(dev); dev = nullptr; cap = 0; }
};

struct cornetto_asm {
    const uint64_t uid = cn_uid_counter.fetch_add(1);
    const uint8_t *d_bases = nullptr;  // 1 B/base, every contig starts at a multiple of 64
    void *owned = nullptr;             // hipMalloc'd storage when uploaded by us
    int32_t n = 0;
    std::vector<int64_t> off;          // byte offset of contig i
    std::vector<int32_t> len;
    int64_t total = 0;                 // sum of lens
    // device copies of the contig table
    int64_t *d_off = nullptr;
    int32_t *d_len = nullptr;
    // cached work decompositions (depend on the contig table only); device copies owned by the object
    std::vector<int32_t> tf_ctg_tile0;   // telofind: first tile of each contig (+ total)
    int2 *d_tf_tiles = nullptr;
    int32_t *d_tf_ct0 = nullptr;         //   tf_ctg_tile0 on the device
    int64_t tf_n_tiles = -1;
    // telofind + telowin: the list totals and the windows of the last fused scan, and for which motif / threshold (cornetto_panel_step sizes
    // the next scan's lists and copies by them and checks afterwards)
    std::string tf_est_key;
    int64_t tf_est_cnt[4] = {-1, -1, -1, -1}, tf_est_wins = -1;
    int64_t tw_n_words = -1, tw_n_tiles = 0;   // telowin on the marks of a fused scan: words of the bitmap (-1: layout not built yet), window tiles
    int64_t *d_tw_boff = nullptr;
    int2 *d_tw_tiles = nullptr;
    std::vector<int64_t> tw_boff;        //   first bit of every contig in the mark bitmap (host copy of d_tw_boff)
    int64_t sd_chunk = -1;               // sdust: chunk size the cached chunk table was built for
    std::vector<int32_t> sd_chunk_ctg;   // contig of every chunk
    void *d_sd_chunks = nullptr;
    int64_t sd_n_chunks = 0;
    CnPrefix sd_pref, tf_pref, tw_pref;  // tiles.hpp: first piece of every contig in the chunk / tile tables (host copy alive while its upload may be in flight)
    uint32_t *d_sd_plan = nullptr;       // sdust: {initial claim flags [n], queue order [n + 160], dense list [n]} of the cached plan
    int64_t sd_plan_key = -1, sd_plan_dense = 0;
    bool sd_refined = false;             // sdust: the flagged chunks of the table have been cut into shorter ones (or need not be)
    int64_t sd_tail0 = 0;                // sdust: first chunk of the part of the table that is made of short chunks
    uint32_t *d_sd_walk = nullptr;       // sdust sift: [0] = n, [1 ..] the chunks that hold other bytes than letters (found by the first call: handed out first afterwards), then a byte per chunk
    int64_t sd_walk_key = -1;            //   chunk table it belongs to; -1: none yet
    // sdust: what the last call for (chunk table, T, W) gave: rows of all chunks / merged intervals.  The next call sizes its gather,
    // merge and result copy by them and checks afterwards (no round trip for the count inside the call); -1: none yet
    int64_t sd_est_key = -1, sd_est_rows = -1, sd_est_out = -1;
    int sd_auto = -1;                    // sdust: which kernel family takes this assembly (-1 not decided, 0 the per-lane recurrence, 1 sift / resolve)
    int64_t sd_flagged = -1;             // sdust: chunks of this table that sd_prep samples as low-complexity (-1: not known yet)
    // sdust: per-256-base-block word-emission prefix table, built only if a lane needs it (N-dense input)
    uint32_t *d_wtab = nullptr;
    int64_t *d_wtab_base = nullptr;
    int64_t n_wblocks = 0;
};

struct cornetto_cov {
    const uint16_t *d_depth = nullptr, *d_mq = nullptr;
    void *owned_d = nullptr, *owned_q = nullptr;
    // a coverage that came from bedgraph text (cornetto_bgin_finish) with NEGATIVE depth values: what their stored uint16 values are above the
    // values themselves, summed — the reference's totals take the int (src/boringbits_main.c:285-286), its arrays the uint16 (:282-283)
    unsigned long long sum_corr[2] = {0, 0};
    std::vector<unsigned long long> ctg_corr;   // the same per contig ([2 c], [2 c + 1]); empty: none
    int32_t n = 0;
    std::vector<int64_t> off;          // element offset of contig i (multiple of 64)
    std::vector<int32_t> len;
    int64_t total = 0;
    int64_t *d_off = nullptr;
    int32_t *d_len = nullptr;
    // stage-1 products (cornetto_cov_prepare)
    int32_t w = 0, inc = 0;
    std::vector<int64_t> blk_off;      // first block-sum slot of contig i
    int64_t n_blk = 0;
    uint32_t *d_blk = nullptr;         // prefixes [n_blk] {depth, mq}, heads [n_blk] {depth, mq} (cov.hip: CbArgs), the tile offsets
    int64_t *d_blk_off = nullptr;
    uint64_t sums[3] = {0, 0, 0};
    // cached work decompositions
    int2 *d_cb_tiles = nullptr;          // block tiles for (w, inc)
    int4 *d_cb_tmeta = nullptr;          // the same with what cov_blocks needs of the contig in ONE 32-byte record: {ctg, first block, length, -} {offset lo, hi, -, -}
    int64_t n_cb_tiles = 0;
    std::vector<int32_t> n_reg;          // windows per contig for (w, inc)
    int32_t *d_n_reg = nullptr;
    int2 *d_cw_tiles = nullptr;          // window tiles of the last selection (mode, min_len)
    int64_t n_cw_tiles = 0;
    std::vector<int32_t> cw_first;       //   host copy of d_cw_first
    int32_t *d_cw_first = nullptr;       //   first of them of every contig (their number: the contig has none)
    int cw_mode = -1;
    int32_t cw_min_len = 0, cw_only = -2;
    // what the last packed selection gave, and for which parameters: the next one with the same parameters sizes its result copy by it
    // and checks afterwards (cornetto_panel_step: no round trip for the count); -1: none yet
    int64_t cw_est_key = -1, cw_est_cnt = -1;
    CnPrefix cb_pref, cw_pref;           // tiles.hpp: as in cornetto_asm
};

// Development switches (chunk sizes, kernel-family selection, forced estimates, ablations): read from the environment ONLY in the development
// build of the library (`make dev` -> libcornetto_hip_dev.so, -DCN_DEV; the tests of the decomposition invariance and of the fallback kernel
// families load that one).  In the product build every switch is its default at compile time and the names do not exist in the binary
// (tests/test_abi.py greps libcornetto_hip.so for them).
#ifdef CN_DEV
static inline int cn_dev_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}
#define CN_DEV_INT(name, dflt) cn_dev_int(name, dflt)
#include <time.h>
// CORNETTO_TRACE=1 (development build): wall-clock stamps of the phases of a library call on stderr
static inline void cn_trace(const char *what)
{
    static const int on = cn_dev_int("CORNETTO_TRACE", 0);
    if (!on) return;
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    static double t0 = 0.0;
    const double now = t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
    if (t0 == 0.0) t0 = now;
    fprintf(stderr, "[lib trace] %-32s %9.3f ms\n", what, now - t0);
}
#define CN_TRACE(what) cn_trace(what)
#else
#define CN_DEV_INT(name, dflt) (dflt)
#define CN_TRACE(what) ((void)0)
#endif

static inline int64_t cn_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Result arrays handed to the caller.  Small ones are malloc'd; large ones (>= 1 MiB) come from a process-wide
// pool of pinned host buffers, so the device-to-host copy runs at full PCIe rate and never page-faults on
// fresh memory.  Either kind is released with cornetto_free() (runtime.hip).
void *cn_result_alloc(size_t bytes);
void cn_result_prewarm(size_t bytes);   // (runtime.hip)

// An assembly object with room for n sequences of the given lengths in the resident layout (every sequence at a
// multiple of 64, zero-filled, contig table on the device); the caller fills a->owned (runtime.hip).
int cn_asm_alloc(cornetto_accel_t *h, const int32_t *lens, int32_t n, cornetto_asm_t **out);

// runtime.hip — handle lifetime, resident data sets (bases / coverage in HBM) and small utilities of the
// C ABI declared in include/cornetto_accel.h.
#include "common.hpp"
#include <chrono>
#include <condition_variable>
#include <deque>
#include <set>
#include <thread>

#include <map>
#include <mutex>
#include <unordered_map>

// ---- pinned result pool ------------------------------------------------------------------------------
namespace {
// (These objects are never destroyed: the background thread below may still wait on the condition variable when the process leaves main(), and the
// destructor of a condition variable with a waiter blocks until the waiter's 200 ms are over — or the waiter wakes to a dead mutex.)
std::mutex &g_pool_mu = *new std::mutex;
std::unordered_map<void *, size_t> &g_pool_live = *new std::unordered_map<void *, size_t>;      // pinned buffers currently owned by callers
std::multimap<size_t, void *> &g_pool_free = *new std::multimap<size_t, void *>;                // pinned buffers ready for reuse, by capacity
std::unordered_map<void *, uint64_t> &g_pool_age = *new std::unordered_map<void *, uint64_t>;   // when a free buffer was handed back (the oldest one makes room)
uint64_t g_pool_clock = 0;
constexpr size_t POOL_MIN = 1u << 20;                // below this, plain malloc
size_t g_pool_free_bytes = 0;                        // capacity of the free list
constexpr size_t POOL_KEEP_BYTES = 4ull << 30;       // ... and its byte budget (the newest buffer always stays: the next step wants it again)
constexpr size_t POOL_KEEP = 12;                     // free buffers kept; beyond that the one that has waited longest goes back to the driver
}  // namespace

// Blocks that are being pinned ahead of their use (cn_result_prewarm): capacities requested and not yet in the free list.  One background thread
// pins them one after the other, in the order of the requests (= the order in which a step needs them).
namespace {
std::condition_variable &g_pre_cv = *new std::condition_variable;                    // (with g_pool_mu) a request arrived / a block landed
std::deque<std::pair<size_t, int>> &g_pre_queue = *new std::deque<std::pair<size_t, int>>;      // capacity, device
std::multiset<size_t> &g_pre_pending = *new std::multiset<size_t>;                   // queued or being pinned
bool g_pre_running = false;

void pre_worker()
{
    std::unique_lock<std::mutex> lk(g_pool_mu);
    for (;;) {
        if (g_pre_queue.empty()) {
            // (the thread ends when nothing has been asked for a while: a process that scans once does not keep it)
            if (!g_pre_cv.wait_for(lk, std::chrono::milliseconds(200), [] { return !g_pre_queue.empty(); })) {
                g_pre_running = false;
                return;
            }
        }
        // (the largest first: the step wants the selected coverage windows, its largest result, before the telomere runs and the sdust intervals)
        auto big = g_pre_queue.begin();
        for (auto q = g_pre_queue.begin(); q != g_pre_queue.end(); ++q)
            if (q->first > big->first) big = q;
        const std::pair<size_t, int> job = *big;
        g_pre_queue.erase(big);
        lk.unlock();
        (void)hipSetDevice(job.second);
        void *p = nullptr;
        const bool ok = hipHostMalloc(&p, job.first, hipHostMallocDefault) == hipSuccess && p;
        lk.lock();
        auto it = g_pre_pending.find(job.first);
        if (it != g_pre_pending.end()) g_pre_pending.erase(it);
        if (ok) {
            g_pool_free.emplace(job.first, p);
            g_pool_age[p] = ++g_pool_clock;
            g_pool_free_bytes += job.first;
        }
        g_pre_cv.notify_all();
    }
}
}  // namespace

void *cn_result_alloc(size_t bytes)
{
    if (bytes < POOL_MIN) return malloc(bytes ? bytes : 1);
    size_t cap = POOL_MIN;
    while (cap < bytes) cap <<= 1;
    {
        std::unique_lock<std::mutex> lk(g_pool_mu);
        for (;;) {
            auto it = g_pool_free.lower_bound(cap);
            if (it != g_pool_free.end() && it->first <= 2 * cap) {
                void *p = it->second;
                g_pool_live[p] = it->first;
                g_pool_free_bytes -= it->first;
                g_pool_free.erase(it);
                g_pool_age.erase(p);
                return p;
            }
            // a block that would serve is being pinned ahead (cn_result_prewarm): it is ready sooner than one pinned from here
            auto pe = g_pre_pending.lower_bound(cap);
            if (pe == g_pre_pending.end() || *pe > 2 * cap) break;
            g_pre_cv.wait(lk);
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess || !p) return malloc(bytes);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_live[p] = cap;
    return p;
}

// A pinned block for a result of about `bytes` on its way into the pool ahead of its use, unless one that would serve is there or on its way: called
// when a resident object comes into being, for the results its scans will return — page-locking runs at ~20 GB/s (3 ms for the 60 MB of selected
// windows of a 3 Gbp assembly), and until round 5 every first scan paid it behind its kernels: 35 of the first pass's 45 ms were such first-time costs.
// A guess that is too small only means the exact allocation pins again, as it did before; one that is too large, a block that waits in the pool.
void cn_result_prewarm(size_t bytes)
{
    if (bytes < POOL_MIN) return;
    size_t cap = POOL_MIN;
    while (cap < bytes) cap <<= 1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_free.lower_bound(cap);
    if (it != g_pool_free.end() && it->first <= 2 * cap) return;
    auto pe = g_pre_pending.lower_bound(cap);
    if (pe != g_pre_pending.end() && *pe <= 2 * cap) return;
    g_pre_pending.insert(cap);
    g_pre_queue.emplace_back(cap, dev);
    if (!g_pre_running) {
        try {
            std::thread(pre_worker).detach();
            g_pre_running = true;
        } catch (...) {
            g_pre_queue.pop_back();
            g_pre_pending.erase(g_pre_pending.find(cap));
            return;
        }
    }
    g_pre_cv.notify_all();
}

extern "C" {

int cornetto_accel_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cornetto_accel_open(cornetto_accel_t **out, int device, void *stream)
{
    if (!out) return CORNETTO_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return CORNETTO_E_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return CORNETTO_E_NODEVICE;
#ifdef CN_WS_ASYNC
    {
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess && pool) {
            uint64_t thr = ~0ull;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
        }
    }
#endif
    cornetto_accel_t *h = new (std::nothrow) cornetto_accel;
    if (!h) return CORNETTO_E_NOMEM;
    h->device = device;
    if (stream) {
        h->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
            delete h;
            return CORNETTO_E_NODEVICE;
        }
        h->own_stream = true;
    }
    *out = h;
    return CORNETTO_OK;
}

void cornetto_accel_close(cornetto_accel_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->sd_pend.state != 0 && h->sd_pend.of) cornetto_free(h->sd_pend.of);      // (a cornetto_sdust_asm_begin() nobody finished: its copy is through now)
    h->sd_pend.state = 0;
    for (auto &r : h->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : h->pool) (void)hipEventDestroy(e);
    for (auto &w : h->dev)
        if (w.p) (void)hipFree(w.p);
    for (auto &w : h->pin)
        if (w.p) (void)hipHostFree(w.p);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->ev3) (void)hipEventDestroy(h->ev3);
    if (h->copy_stream) {
        (void)hipStreamSynchronize(h->copy_stream);
        (void)hipStreamDestroy(h->copy_stream);
    }
    if (h->ev_cp) (void)hipEventDestroy(h->ev_cp);
    if (h->stream2) {
        (void)hipStreamSynchronize(h->stream2);       // (a dense kernel of a call that failed half-way may still be writing workspaces)
        (void)hipStreamDestroy(h->stream2);
    }
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *cornetto_accel_last_error(const cornetto_accel_t *h) { return h ? h->err : "null handle"; }

const char *cornetto_accel_strerror(int status)
{
    switch (status) {
    case CORNETTO_OK: return "ok";
    case CORNETTO_E_NODEVICE: return "no usable HIP device";
    case CORNETTO_E_HIP: return "HIP runtime or kernel failure";
    case CORNETTO_E_ARG: return "invalid argument";
    case CORNETTO_E_NOMEM: return "out of memory";
    case CORNETTO_E_UNSUPPORTED: return "parameter outside the supported range";
    case CORNETTO_E_FORMAT: return "malformed input text";
    case CORNETTO_E_ASSERT: return "the reference aborts on an assert here";
    default: return "unknown status";
    }
}

void cornetto_free(void *p)
{
    if (!p) return;
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_live.find(p);
        if (it != g_pool_live.end()) {
            // keep what was used last: while the list is over its count or its byte budget the buffer that has waited longest leaves (a
            // list full of the sizes of an earlier workload made every step of the next one pin and unpin its result buffers: +7 ms per
            // step, round 4; a process that once returned multi-GB results must not keep them page-locked for good either)
            const size_t cap = it->second;
            g_pool_live.erase(it);
            g_pool_free.emplace(cap, p);
            g_pool_age[p] = ++g_pool_clock;
            g_pool_free_bytes += cap;
            while (g_pool_free.size() > 1 && (g_pool_free.size() > POOL_KEEP || g_pool_free_bytes > POOL_KEEP_BYTES)) {
                auto oldest = g_pool_free.end();
                uint64_t best = ~0ull;
                for (auto f = g_pool_free.begin(); f != g_pool_free.end(); ++f) {
                    const auto a = g_pool_age.find(f->second);
                    const uint64_t age = a == g_pool_age.end() ? 0 : a->second;
                    if (age < best) {
                        best = age;
                        oldest = f;
                    }
                }
                if (oldest == g_pool_free.end()) break;
                drop.push_back(oldest->second);
                g_pool_free_bytes -= oldest->first;
                g_pool_age.erase(oldest->second);
                g_pool_free.erase(oldest);
            }
            p = nullptr;
        }
    }
    for (void *d : drop) (void)hipHostFree(d);       // (outside the lock)
    if (p) free(p);
}

int cornetto_accel_set_share(cornetto_accel_t *h, int percent)
{
    if (!h) return CORNETTO_E_ARG;
    if (percent < 10 || percent > 100) return cn_fail(h, CORNETTO_E_ARG, "set_share: %d outside 10..100", percent);
    h->share = percent;
    return CORNETTO_OK;
}

unsigned long long cornetto_accel_launch_count(const cornetto_accel_t *h)
{
    return h ? __atomic_load_n(&h->launch_seq, __ATOMIC_ACQUIRE) : 0ull;
}

int cornetto_accel_set_lazy(cornetto_accel_t *h, int on)
{
    if (!h) return CORNETTO_E_ARG;
    if (!on && h->copies_pending) {
        CN_HIP(h, hipStreamSynchronize(h->copy_stream));
        h->copies_pending = false;
    }
    h->lazy = on ? 1 : 0;
    if (on && !h->copy_stream) {
        // (now, not at the first lazy copy: creating a queue takes milliseconds while another stream's kernel is running — the first selection of a
        // two-stream step spent 5 of its 6 ms here)
        CN_HIP(h, hipSetDevice(h->device));
        CN_HIP(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        CN_HIP(h, hipEventCreateWithFlags(&h->ev_cp, hipEventDisableTiming));
    }
    return CORNETTO_OK;
}

int cornetto_accel_wait(cornetto_accel_t *h)
{
    if (!h) return CORNETTO_E_ARG;
    if (h->copies_pending) {
        CN_HIP(h, hipSetDevice(h->device));
        CN_HIP(h, hipStreamSynchronize(h->copy_stream));
        h->copies_pending = false;
    }
    return CORNETTO_OK;
}

int cornetto_accel_boost(cornetto_accel_t *h, int on)
{
    if (!h) return CORNETTO_E_ARG;
    __atomic_store_n(&h->boost, on ? 1 : 0, __ATOMIC_RELEASE);
    return CORNETTO_OK;
}

int cornetto_accel_warm(cornetto_accel_t *h, int what)
{
    // Every entry point once on 4 kb: the first use of a code object loads it, the first copies set the copy engines up, the first results make the small
    // pools — 13 of the 25 ms by which the first pass of a process over an assembly exceeded the second (tools/perf_cold.py --warm 1).  A caller that
    // has something else to do first (read its input) runs this on the thread that opened the handle.
    if (!h) return CORNETTO_E_ARG;
    const int32_t n = 4096;
    std::vector<uint8_t> seq((size_t)n);
    std::vector<uint16_t> d((size_t)n), q((size_t)n);
    uint32_t x = 12345u;
    for (int32_t i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        seq[(size_t)i] = "ACGT"[(x >> 24) & 3];
        d[(size_t)i] = (uint16_t)(20 + ((x >> 16) & 15));
        q[(size_t)i] = (uint16_t)(d[(size_t)i] >> ((i / 700) & 1));
    }
    for (int32_t i = 100; i < 700; ++i) seq[(size_t)i] = "TTAGGG"[(i - 100) % 6];
    for (int32_t i = 1000; i < 1100; ++i) seq[(size_t)i] = 'A';
    int rc = CORNETTO_OK;
    {
        // the copy engines: the first few device-to-host copies of a quarter megabyte and more on a stream block their caller for ~7 ms each (the runtime
        // sets a copy queue up; traced in the telomere window scan's 256 KB read-back: three slow calls, then none) — on this handle's streams, now
        CN_HIP(h, hipSetDevice(h->device));
        const size_t nb = (size_t)1 << 20;
        void *dv = cn_ws(h, WS_FQ_CNT, nb);
        void *pv = cn_result_alloc(nb);
        if (dv && pv) {
            for (int i = 0; i < 4; ++i) {
                (void)hipMemcpyAsync(pv, dv, nb, hipMemcpyDeviceToHost, h->stream);
                (void)hipStreamSynchronize(h->stream);
                if (h->copy_stream) {
                    (void)hipMemcpyAsync(pv, dv, nb, hipMemcpyDeviceToHost, h->copy_stream);
                    (void)hipStreamSynchronize(h->copy_stream);
                }
            }
        }
        if (pv) cornetto_free(pv);
    }
    if (what & (CORNETTO_WARM_SDUST | CORNETTO_WARM_TELO)) {
        const uint8_t *sp = seq.data();
        const int64_t ln = n;
        cornetto_asm_t *a = nullptr;
        rc = cornetto_asm_upload(h, &sp, &ln, 1, &a);
        if (rc == CORNETTO_OK && (what & CORNETTO_WARM_SDUST)) {
            for (int rep = 0; rep < 2 && rc == CORNETTO_OK; ++rep) {      // (the second call takes the one-go path)
                cornetto_ivl_t *iv = nullptr;
                int64_t ni = 0;
                rc = cornetto_sdust_asm(h, a, 20, 64, &iv, &ni);
                cornetto_free(iv);
            }
        }
        if (rc == CORNETTO_OK && (what & CORNETTO_WARM_TELO)) {
            cornetto_hit_t *hits = nullptr;
            cornetto_win_t *wins = nullptr;
            int64_t nh = 0, nw = 0;
            rc = cornetto_telo_scan(h, a, "TTAGGG", 0.39, &hits, &nh, &wins, &nw);
            cornetto_free(hits);
            free(wins);
        }
        cornetto_asm_free(h, a);
    }
    if (rc == CORNETTO_OK && (what & CORNETTO_WARM_COV)) {
        const uint16_t *dp = d.data(), *qp = q.data();
        cornetto_cov_t *c = nullptr;
        rc = cornetto_cov_upload(h, &dp, &qp, &n, 1, &c);
        uint64_t sums[3];
        if (rc == CORNETTO_OK) rc = cornetto_cov_prepare(h, c, 500, 50, sums);
        if (rc == CORNETTO_OK) {
            cornetto_regpk_t *pk = nullptr;
            int64_t np = 0, *cf = nullptr;
            rc = cornetto_cov_select_packed(h, c, 10, 60, 0.4f, 100, 1000, 0, &pk, &np, &cf);
            cornetto_free(pk);
            free(cf);
        }
        if (c) cornetto_cov_free(h, c);
    }
    return rc;
}

int cornetto_accel_set_timing(cornetto_accel_t *h, int level)
{
    if (!h || level < 0 || level > 2) return cn_fail(h, CORNETTO_E_ARG, "set_timing: level must be 0, 1 or 2");
    h->timing = level;
    return CORNETTO_OK;
}

int cornetto_accel_last_timing(const cornetto_accel_t *h, const char **names, float *ms, int cap)
{
    if (!h) return 0;
    int n = (int)h->last.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = h->last[i].first;
        if (ms) ms[i] = h->last[i].second;
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------
// sequences
// ---------------------------------------------------------------------------------------------------
static const int64_t SLACK = 256;  // readable bytes behind the last contig

static int asm_finish_table(cornetto_accel_t *h, cornetto_asm_t *a)
{
    size_t n = (size_t)(a->n > 0 ? a->n : 1);
    CN_HIP(h, hipMalloc((void **)&a->d_off, n * sizeof(int64_t)));
    CN_HIP(h, hipMalloc((void **)&a->d_len, n * sizeof(int32_t)));
    if (a->n) {
        CN_HIP(h, hipMemcpyAsync(a->d_off, a->off.data(), a->n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
        CN_HIP(h, hipMemcpyAsync(a->d_len, a->len.data(), a->n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    }
    CN_HIP(h, hipStreamSynchronize(h->stream));
    // the results the scans of an assembly will return, pinned while the caller gets to its first scan (cn_result_prewarm): a telomere run per ~1.5 kb
    // with planted arrays, an sdust interval per ~4 kb — assemblies only: a batch of reads or a test's kilobases pin what they need when they need it
    if (a->total >= (64ll << 20)) {
        cn_result_prewarm((size_t)(a->total / 1536 + 1) * sizeof(cornetto_hit_t));
        cn_result_prewarm((size_t)(a->total / 4096 + 1) * sizeof(cornetto_ivl_t));
    }
    return CORNETTO_OK;
}

void cornetto_asm_free(cornetto_accel_t *h, cornetto_asm_t *a)
{
    if (!a) return;
    if (h) (void)hipSetDevice(h->device);
    if (h && h->sd_pend.state != 0 && h->sd_pend.a == a) {
        // a cornetto_sdust_asm_begin() over this assembly that nobody finished: its kernels and its result copy are let through, the result
        // is dropped, and the handle forgets the assembly (the next sdust call would otherwise finish it through a dangling pointer)
        if (h->sd_pend.state == 1) (void)hipStreamSynchronize(h->stream);
        if (h->sd_pend.of) cornetto_free(h->sd_pend.of);
        h->sd_pend = cornetto_accel::SdPend{};
    }
    if (a->owned) (void)hipFree(a->owned);
    if (a->d_off) (void)hipFree(a->d_off);
    if (a->d_len) (void)hipFree(a->d_len);
    if (a->d_tf_tiles) (void)hipFree(a->d_tf_tiles);
    if (a->d_tf_ct0) (void)hipFree(a->d_tf_ct0);
    if (a->d_tw_boff) (void)hipFree(a->d_tw_boff);
    if (a->d_tw_tiles) (void)hipFree(a->d_tw_tiles);
    if (a->d_sd_chunks) (void)hipFree(a->d_sd_chunks);
    a->sd_pref.release(); a->tf_pref.release(); a->tw_pref.release();
    if (a->d_sd_walk) (void)hipFree(a->d_sd_walk);
    if (a->d_sd_plan) (void)hipFree(a->d_sd_plan);
    if (a->d_wtab) (void)hipFree(a->d_wtab);
    if (a->d_wtab_base) (void)hipFree(a->d_wtab_base);
    delete a;
}

int cornetto_asm_upload(cornetto_accel_t *h, const uint8_t *const *seqs, const int64_t *lens, int32_t n,
                        cornetto_asm_t **out)
{
    if (!h || !out || n < 0 || (n > 0 && (!seqs || !lens))) return cn_fail(h, CORNETTO_E_ARG, "asm_upload: bad argument");
    *out = nullptr;
    CN_HIP(h, hipSetDevice(h->device));
    cornetto_asm_t *a = new (std::nothrow) cornetto_asm;
    if (!a) return cn_fail(h, CORNETTO_E_NOMEM, "asm_upload: host allocation failed");
    a->n = n;
    int64_t pos = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (lens[i] < 0 || lens[i] > INT32_MAX) {
            delete a;
            return cn_fail(h, CORNETTO_E_ARG, "asm_upload: contig %d has length %lld (must be 0..2^31-1)", i, (long long)lens[i]);
        }
        a->off.push_back(pos);
        a->len.push_back((int32_t)lens[i]);
        a->total += lens[i];
        pos = cn_align_up(pos + lens[i], 64);
    }
    int64_t bytes = pos + SLACK;
    if (hipMalloc(&a->owned, (size_t)bytes) != hipSuccess) {
        delete a;
        return cn_fail(h, CORNETTO_E_NOMEM, "asm_upload: hipMalloc of %lld bytes failed", (long long)bytes);
    }
    a->d_bases = (const uint8_t *)a->owned;
    int rc = CORNETTO_OK;
    if (hipMemsetAsync(a->owned, 0, (size_t)bytes, h->stream) != hipSuccess) rc = CORNETTO_E_HIP;
    for (int32_t i = 0; i < n && rc == CORNETTO_OK; ++i)
        if (lens[i] > 0 && hipMemcpyAsync((uint8_t *)a->owned + a->off[i], seqs[i], (size_t)lens[i], hipMemcpyHostToDevice, h->stream) != hipSuccess)
            rc = CORNETTO_E_HIP;
    if (rc == CORNETTO_OK) rc = asm_finish_table(h, a);
    if (rc != CORNETTO_OK) {
        cornetto_asm_free(h, a);
        return cn_fail(h, rc, "asm_upload: copy to device failed");
    }
    *out = a;
    return CORNETTO_OK;
}

int cornetto_asm_wrap(cornetto_accel_t *h, const void *d_bases, const int64_t *offsets, const int64_t *lens,
                      int32_t n, cornetto_asm_t **out)
{
    if (!h || !out || n < 0 || (n > 0 && (!d_bases || !offsets || !lens))) return cn_fail(h, CORNETTO_E_ARG, "asm_wrap: bad argument");
    *out = nullptr;
    if (((uintptr_t)d_bases & 63) != 0) return cn_fail(h, CORNETTO_E_ARG, "asm_wrap: device pointer must be 64-byte aligned");
    CN_HIP(h, hipSetDevice(h->device));
    cornetto_asm_t *a = new (std::nothrow) cornetto_asm;
    if (!a) return cn_fail(h, CORNETTO_E_NOMEM, "asm_wrap: host allocation failed");
    a->n = n;
    a->d_bases = (const uint8_t *)d_bases;
    for (int32_t i = 0; i < n; ++i) {
        if (lens[i] < 0 || lens[i] > INT32_MAX || offsets[i] < 0 || (offsets[i] & 63)) {
            delete a;
            return cn_fail(h, CORNETTO_E_ARG, "asm_wrap: contig %d: offset %lld must be a non-negative multiple of 64, length %lld within 0..2^31-1",
                           i, (long long)offsets[i], (long long)lens[i]);
        }
        a->off.push_back(offsets[i]);
        a->len.push_back((int32_t)lens[i]);
        a->total += lens[i];
    }
    int rc = asm_finish_table(h, a);
    if (rc != CORNETTO_OK) {
        cornetto_asm_free(h, a);
        return rc;
    }
    *out = a;
    return CORNETTO_OK;
}

// ---------------------------------------------------------------------------------------------------
// coverage
// ---------------------------------------------------------------------------------------------------
void cornetto_cov_free(cornetto_accel_t *h, cornetto_cov_t *c)
{
    if (!c) return;
    if (h) (void)hipSetDevice(h->device);
    if (c->owned_d) (void)hipFree(c->owned_d);
    if (c->owned_q) (void)hipFree(c->owned_q);
    if (c->d_off) (void)hipFree(c->d_off);
    if (c->d_len) (void)hipFree(c->d_len);
    if (c->d_blk) (void)hipFree(c->d_blk);
    if (c->d_blk_off) (void)hipFree(c->d_blk_off);
    if (c->d_cb_tiles) (void)hipFree(c->d_cb_tiles);
    if (c->d_cb_tmeta) (void)hipFree(c->d_cb_tmeta);
    if (c->d_n_reg) (void)hipFree(c->d_n_reg);
    if (c->d_cw_tiles) (void)hipFree(c->d_cw_tiles);
    if (c->d_cw_first) (void)hipFree(c->d_cw_first);
    c->cb_pref.release(); c->cw_pref.release();
    delete c;
}

int32_t cornetto_cov_n(const cornetto_cov_t *c) { return c ? c->n : 0; }

const int32_t *cornetto_cov_lens(const cornetto_cov_t *c) { return c ? c->len.data() : nullptr; }

static int cov_finish_table(cornetto_accel_t *h, cornetto_cov_t *c)
{
    size_t n = (size_t)(c->n > 0 ? c->n : 1);
    CN_HIP(h, hipMalloc((void **)&c->d_off, n * sizeof(int64_t)));
    CN_HIP(h, hipMalloc((void **)&c->d_len, n * sizeof(int32_t)));
    if (c->n) {
        CN_HIP(h, hipMemcpyAsync(c->d_off, c->off.data(), c->n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
        CN_HIP(h, hipMemcpyAsync(c->d_len, c->len.data(), c->n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    }
    CN_HIP(h, hipStreamSynchronize(h->stream));
    // the selected windows a scan of the coverage will return, pinned while the caller gets to it (cn_result_prewarm): an eighth of the windows at the
    // default step of 50 positions, 8 bytes each in the packed form
    if (c->total >= (64ll << 20)) cn_result_prewarm((size_t)(c->total / 50 / 8 + 1) * sizeof(cornetto_regpk_t));
    return CORNETTO_OK;
}

int cornetto_cov_upload(cornetto_accel_t *h, const uint16_t *const *depth, const uint16_t *const *mq_depth,
                        const int32_t *lens, int32_t n, cornetto_cov_t **out)
{
    if (!h || !out || n < 0 || (n > 0 && (!depth || !mq_depth || !lens))) return cn_fail(h, CORNETTO_E_ARG, "cov_upload: bad argument");
    *out = nullptr;
    CN_HIP(h, hipSetDevice(h->device));
    cornetto_cov_t *c = new (std::nothrow) cornetto_cov;
    if (!c) return cn_fail(h, CORNETTO_E_NOMEM, "cov_upload: host allocation failed");
    c->n = n;
    int64_t pos = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (lens[i] < 0) {
            delete c;
            return cn_fail(h, CORNETTO_E_ARG, "cov_upload: contig %d has negative length", i);
        }
        c->off.push_back(pos);
        c->len.push_back(lens[i]);
        c->total += lens[i];
        pos = cn_align_up(pos + lens[i], 64);
    }
    size_t bytes = (size_t)(pos + SLACK) * sizeof(uint16_t);
    if (hipMalloc(&c->owned_d, bytes) != hipSuccess || hipMalloc(&c->owned_q, bytes) != hipSuccess) {
        cornetto_cov_free(h, c);
        return cn_fail(h, CORNETTO_E_NOMEM, "cov_upload: hipMalloc of 2 x %zu bytes failed", bytes);
    }
    c->d_depth = (const uint16_t *)c->owned_d;
    c->d_mq = (const uint16_t *)c->owned_q;
    int rc = CORNETTO_OK;
    if (hipMemsetAsync(c->owned_d, 0, bytes, h->stream) != hipSuccess || hipMemsetAsync(c->owned_q, 0, bytes, h->stream) != hipSuccess)
        rc = CORNETTO_E_HIP;
    for (int32_t i = 0; i < n && rc == CORNETTO_OK; ++i) {
        if (lens[i] == 0) continue;
        size_t nb = (size_t)lens[i] * sizeof(uint16_t);
        if (hipMemcpyAsync((uint16_t *)c->owned_d + c->off[i], depth[i], nb, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync((uint16_t *)c->owned_q + c->off[i], mq_depth[i], nb, hipMemcpyHostToDevice, h->stream) != hipSuccess)
            rc = CORNETTO_E_HIP;
    }
    if (rc == CORNETTO_OK) rc = cov_finish_table(h, c);
    if (rc != CORNETTO_OK) {
        cornetto_cov_free(h, c);
        return cn_fail(h, rc, "cov_upload: copy to device failed");
    }
    *out = c;
    return CORNETTO_OK;
}

int cornetto_cov_shard(cornetto_accel_t *h_src, const cornetto_cov_t *src, cornetto_accel_t *h_dst, const int32_t *ctgs, int32_t n, cornetto_cov_t **out)
{
    if (!h_src || !h_dst || !src || !out || n < 0 || (n > 0 && !ctgs)) return cn_fail(h_dst, CORNETTO_E_ARG, "cov_shard: bad argument");
    *out = nullptr;
    for (int32_t i = 0; i < n; ++i)
        if (ctgs[i] < 0 || ctgs[i] >= src->n) return cn_fail(h_dst, CORNETTO_E_ARG, "cov_shard: contig index %d outside 0..%d", ctgs[i], src->n - 1);
    CN_HIP(h_dst, hipSetDevice(h_dst->device));
    cornetto_cov_t *c = new (std::nothrow) cornetto_cov;
    if (!c) return cn_fail(h_dst, CORNETTO_E_NOMEM, "cov_shard: host allocation failed");
    c->n = n;
    int64_t pos = 0;
    for (int32_t i = 0; i < n; ++i) {
        c->off.push_back(pos);
        c->len.push_back(src->len[ctgs[i]]);
        c->total += src->len[ctgs[i]];
        pos = cn_align_up(pos + src->len[ctgs[i]], 64);
    }
    if (!src->ctg_corr.empty()) {                      // (negative depth values of a coverage read from text go with their contigs: common.hpp sum_corr)
        c->ctg_corr.assign(2 * (size_t)n, 0ull);
        for (int32_t i = 0; i < n; ++i)
            for (int f = 0; f < 2; ++f) {
                c->ctg_corr[2 * (size_t)i + f] = src->ctg_corr[2 * (size_t)ctgs[i] + f];
                c->sum_corr[f] += src->ctg_corr[2 * (size_t)ctgs[i] + f];
            }
    }
    const size_t bytes = (size_t)(pos + SLACK) * sizeof(uint16_t);
    if (hipMalloc(&c->owned_d, bytes) != hipSuccess || hipMalloc(&c->owned_q, bytes) != hipSuccess) {
        cornetto_cov_free(h_dst, c);
        return cn_fail(h_dst, CORNETTO_E_NOMEM, "cov_shard: hipMalloc of 2 x %zu bytes failed", bytes);
    }
    c->d_depth = (const uint16_t *)c->owned_d;
    c->d_mq = (const uint16_t *)c->owned_q;
    int rc = CORNETTO_OK;
    // everything the source handle queued (the ingest) has to be done before the peer copies read it
    if (hipSetDevice(h_src->device) != hipSuccess || hipStreamSynchronize(h_src->stream) != hipSuccess) rc = CORNETTO_E_HIP;
    if (hipSetDevice(h_dst->device) != hipSuccess) rc = CORNETTO_E_HIP;
    if (rc == CORNETTO_OK && (hipMemsetAsync(c->owned_d, 0, bytes, h_dst->stream) != hipSuccess || hipMemsetAsync(c->owned_q, 0, bytes, h_dst->stream) != hipSuccess))
        rc = CORNETTO_E_HIP;
    for (int32_t i = 0; i < n && rc == CORNETTO_OK;) {
        // a run of contigs that follow each other in the source, laid out there by the same rule as here (next offset = this one + length, rounded up
        // to 64), is one copy: a read-level coverage set has 1e5 .. 1e6 contigs, a device's share of it thousands of runs, not hundreds of thousands
        int32_t j = i;
        while (j + 1 < n && ctgs[j + 1] == ctgs[j] + 1 && src->off[ctgs[j + 1]] == cn_align_up(src->off[ctgs[j]] + src->len[ctgs[j]], 64)) ++j;
        const size_t nb = (size_t)(src->off[ctgs[j]] + src->len[ctgs[j]] - src->off[ctgs[i]]) * sizeof(uint16_t);
        // xGMI peer copy (staged through the host by the runtime where peer access is not available); the same device is a plain copy
        if (nb && (hipMemcpyPeerAsync((uint16_t *)c->owned_d + c->off[i], h_dst->device, src->d_depth + src->off[ctgs[i]], h_src->device, nb, h_dst->stream) != hipSuccess ||
                   hipMemcpyPeerAsync((uint16_t *)c->owned_q + c->off[i], h_dst->device, src->d_mq + src->off[ctgs[i]], h_src->device, nb, h_dst->stream) != hipSuccess))
            rc = CORNETTO_E_HIP;
        i = j + 1;
    }
    if (rc == CORNETTO_OK) rc = cov_finish_table(h_dst, c);
    if (rc == CORNETTO_OK && hipStreamSynchronize(h_dst->stream) != hipSuccess) rc = CORNETTO_E_HIP;
    if (rc != CORNETTO_OK) {
        cornetto_cov_free(h_dst, c);
        return cn_fail(h_dst, rc, "cov_shard: copy between devices failed");
    }
    *out = c;
    return CORNETTO_OK;
}

int cornetto_cov_wrap(cornetto_accel_t *h, const void *d_depth, const void *d_mq_depth, const int64_t *offsets,
                      const int32_t *lens, int32_t n, cornetto_cov_t **out)
{
    if (!h || !out || n < 0 || (n > 0 && (!d_depth || !d_mq_depth || !offsets || !lens))) return cn_fail(h, CORNETTO_E_ARG, "cov_wrap: bad argument");
    *out = nullptr;
    if (((uintptr_t)d_depth & 127) || ((uintptr_t)d_mq_depth & 127)) return cn_fail(h, CORNETTO_E_ARG, "cov_wrap: device pointers must be 128-byte aligned");
    CN_HIP(h, hipSetDevice(h->device));
    cornetto_cov_t *c = new (std::nothrow) cornetto_cov;
    if (!c) return cn_fail(h, CORNETTO_E_NOMEM, "cov_wrap: host allocation failed");
    c->n = n;
    c->d_depth = (const uint16_t *)d_depth;
    c->d_mq = (const uint16_t *)d_mq_depth;
    for (int32_t i = 0; i < n; ++i) {
        if (lens[i] < 0 || offsets[i] < 0 || (offsets[i] & 63)) {
            delete c;
            return cn_fail(h, CORNETTO_E_ARG, "cov_wrap: contig %d: element offset %lld must be a non-negative multiple of 64", i, (long long)offsets[i]);
        }
        c->off.push_back(offsets[i]);
        c->len.push_back(lens[i]);
        c->total += lens[i];
    }
    int rc = cov_finish_table(h, c);
    if (rc != CORNETTO_OK) {
        cornetto_cov_free(h, c);
        return rc;
    }
    *out = c;
    return CORNETTO_OK;
}

}  // extern "C"

int cn_asm_alloc(cornetto_accel_t *h, const int32_t *lens, int32_t n, cornetto_asm_t **out)
{
    *out = nullptr;
    cornetto_asm_t *a = new (std::nothrow) cornetto_asm;
    if (!a) return cn_fail(h, CORNETTO_E_NOMEM, "asm_alloc: host allocation failed");
    a->n = n;
    int64_t pos = 0;
    for (int32_t i = 0; i < n; ++i) {
        a->off.push_back(pos);
        a->len.push_back(lens[i]);
        a->total += lens[i];
        pos = cn_align_up(pos + lens[i], 64);
    }
    const int64_t bytes = pos + SLACK;
    if (hipMalloc(&a->owned, (size_t)bytes) != hipSuccess) {
        delete a;
        return cn_fail(h, CORNETTO_E_NOMEM, "asm_alloc: hipMalloc of %lld bytes failed", (long long)bytes);
    }
    a->d_bases = (const uint8_t *)a->owned;
    int rc = hipMemsetAsync(a->owned, 0, (size_t)bytes, h->stream) == hipSuccess ? CORNETTO_OK : CORNETTO_E_HIP;
    if (rc == CORNETTO_OK) rc = asm_finish_table(h, a);
    if (rc != CORNETTO_OK) {
        cornetto_asm_free(h, a);
        return cn_fail(h, rc, "asm_alloc: device set-up failed");
    }
    *out = a;
    return CORNETTO_OK;
}

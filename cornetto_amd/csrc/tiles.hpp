// tiles.hpp — the decomposition tables of a resident object (sdust chunks, telomere tiles, coverage block and window tiles), built ON THE DEVICE.
// Every such table is "the sequences in order, each cut into pieces of one size": a function of the sequence lengths alone.  Rounds 1-5 built them
// on the host — a push_back per piece (1.76 M sdust chunks, 0.75 M telomere tiles, 2 x 0.25 M coverage tiles for the 3.16 Gbp assembly), a
// pageable copy of tens of megabytes and a synchronisation each: most of the 35 ms by which the FIRST pass over an assembly exceeded the later
// ones, and a panel run scans an assembly once.  Now the host only makes the prefix first[c] = index of sequence c's first piece (n + 1 numbers),
// and one launch writes the records: thread t finds its sequence by bisection (the prefix of an assembly is a few hundred bytes: L1 / scalar cache).
#pragma once
#include "common.hpp"

namespace cntiles {

// the sequence c with first[c] <= t < first[c + 1]  (first[0] = 0, first[n] = number of pieces; sequences without a piece are skipped)
__device__ __forceinline__ int seq_of(const int64_t *__restrict__ first, int n, int64_t t)
{
    int lo = 0, hi = n;                               // first[lo] <= t < first[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (first[mid] <= t) lo = mid;
        else hi = mid;
    }
    return lo;
}

// f(t, c, j): piece t of the table = piece j of sequence c
template <class F>
__global__ __launch_bounds__(256) void fill(const int64_t *__restrict__ first, int n, int64_t nt, F f)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nt) return;
    const int c = seq_of(first, n, t);
    f(t, c, t - first[c]);
}

// host: P.host[] from a per-sequence piece count, uploaded to P.dev on the handle's stream (no synchronisation: P belongs to the resident object
// and outlives the copy).  -> the number of pieces, or -1 when the device array could not be allocated.
template <class Count>
static inline int64_t prefix(cornetto_accel_t *h, CnPrefix &P, int32_t n, Count count)
{
    P.host.resize((size_t)n + 1);
    int64_t tot = 0;
    for (int32_t c = 0; c < n; ++c) {
        P.host[c] = tot;
        tot += count(c);
    }
    P.host[n] = tot;
    if (P.cap < (size_t)n + 1) {
        P.release();
        if (cn_obj_malloc(h, (void **)&P.dev, ((size_t)n + 1) * 8) != hipSuccess) return -1;
        P.cap = (size_t)n + 1;
    }
    if (hipMemcpyAsync(P.dev, P.host.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) return -1;
    return tot;
}

}  // namespace cntiles

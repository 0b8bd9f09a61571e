// panel.hip — the interval algebra between `noboringbits` and `bigenough` in the reference's panel scripts
// (scripts/create-cornetto.sh:41-66, scripts/recreate-cornetto.sh:34-49), without bedtools / sort / awk and without
// the text round trip of 7.5 M window lines:
//   step 2-3  bedtools merge -d 1000 | awk '($3-$2)>=30000'     cornetto_cov_select_merged (cov.hip, on the device)
//   step 4-9  lowQ filter, +-40 kb, 200 kb contig edges, merge -d 200000, subtract from the assembly, drop short
//             contigs                                            cornetto_panel_boring (here, host: a few thousand rows)
// and the generic device merge of an ordered interval list, cornetto_ivl_merge.
// PARITY UNPINNED: bedtools is not available where this was written; the semantics below are those of the bedtools
// documentation (merge -d: features at most d apart are merged, book-ended ones included; subtract: the parts of A
// that no B covers) and of the awk one-liners as written in the scripts.
#include <algorithm>

#include "common.hpp"
#include "ivlmerge.hpp"

extern "C" {

int cornetto_ivl_merge(cornetto_accel_t *h, const cornetto_ivl_t *in, int64_t n, int32_t dist, cornetto_ivl_t **out, int64_t *n_out)
{
    if (!h || !out || !n_out || n < 0 || (n > 0 && !in) || dist < 0) return cn_fail(h, CORNETTO_E_ARG, "ivl_merge: bad argument");
    *out = nullptr;
    *n_out = 0;
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    cornetto_ivl_t *o = nullptr;
    int64_t m = 0;
    if (n > 0) {
        for (int64_t i = 1; i < n; ++i)
            if (in[i].ctg < in[i - 1].ctg || (in[i].ctg == in[i - 1].ctg && in[i].start < in[i - 1].start))
                return cn_fail(h, CORNETTO_E_ARG, "ivl_merge: interval %lld is not in (contig, start) order", (long long)i);
        uint8_t *ws = (uint8_t *)cn_ws(h, WS_IVL_MERGE, cnivl::ws_bytes((size_t)n) + 2 * (size_t)n * sizeof(cornetto_ivl_t) + 16);
        unsigned long long *p_cnt = (unsigned long long *)cn_pin(h, PIN_SMALL, 64);
        if (!ws || !p_cnt) return cn_fail(h, CORNETTO_E_NOMEM, "ivl_merge: workspace allocation failed");
        cornetto_ivl_t *d_in = (cornetto_ivl_t *)(ws + cnivl::ws_bytes((size_t)n)), *d_out = d_in + n;
        unsigned long long *d_cnt = (unsigned long long *)(((uintptr_t)(d_out + n) + 7) & ~(uintptr_t)7);
        CN_HIP(h, hipMemcpyAsync(d_in, in, (size_t)n * sizeof(cornetto_ivl_t), hipMemcpyHostToDevice, h->stream));
        CN_TRY(cnivl::merge(h, "ivl_merge", d_in, n, dist, ws, d_out, d_cnt));
        CN_HIP(h, hipMemcpyAsync(p_cnt, d_cnt, 8, hipMemcpyDeviceToHost, h->stream));
        CN_HIP(h, hipStreamSynchronize(h->stream));
        m = (int64_t)p_cnt[0];
        o = (cornetto_ivl_t *)cn_result_alloc(((size_t)m ? (size_t)m : 1) * sizeof(cornetto_ivl_t));
        if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "ivl_merge: host allocation failed");
        if (m > 0 && (hipMemcpyAsync(o, d_out, (size_t)m * sizeof(cornetto_ivl_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                      hipStreamSynchronize(h->stream) != hipSuccess)) {
            cornetto_free(o);
            return cn_fail(h, CORNETTO_E_HIP, "ivl_merge: copy back failed");
        }
    }
    cn_timing_end(h);
    if (!o) {
        o = (cornetto_ivl_t *)malloc(sizeof(cornetto_ivl_t));
        if (!o) return cn_fail(h, CORNETTO_E_NOMEM, "ivl_merge: host allocation failed");
    }
    *out = o;
    *n_out = m;
    return CORNETTO_OK;
}

void cornetto_panel_defaults(cornetto_panel_opt_t *o)
{
    if (!o) return;
    o->min_lowq_len = 8000;     /* create-cornetto.sh:50 */
    o->extend = 40000;          /* :53 */
    o->edge_len = 200000;       /* :56 */
    o->merge_dist = 200000;     /* :59 */
    o->min_ctg_len = 800000;    /* :65 */
    o->extend_right = 40000;    /* :53 */
    o->extend_gate = 40000;     /* :53 */
}

void cornetto_panel_defaults_recreate(cornetto_panel_opt_t *o)
{
    if (!o) return;
    o->min_lowq_len = 7500;     /* recreate-cornetto.sh:35 */
    o->extend = 40000;          /* :38  print $2-40000 */
    o->extend_right = 50000;    /* :38  $3+50000 */
    o->extend_gate = 50000;     /* :38  if ($2 > 50000) */
    o->edge_len = 200000;       /* :41 */
    o->merge_dist = 200000;     /* :44 */
    o->min_ctg_len = 1000000;   /* :50 */
}

int cornetto_panel_boring(const int32_t *ctg_len, int32_t n_ctg, const cornetto_ivl_t *fun, int64_t n_fun, const cornetto_ivl_t *lowq,
                          int64_t n_lowq, const cornetto_panel_opt_t *opt, cornetto_ivl_t **boring, int64_t *n_boring)
{
    if (!boring || !n_boring || !opt || n_ctg < 0 || n_fun < 0 || n_lowq < 0 || (n_ctg > 0 && !ctg_len) || (n_fun > 0 && !fun) || (n_lowq > 0 && !lowq))
        return CORNETTO_E_ARG;
    *boring = nullptr;
    *n_boring = 0;
    std::vector<cornetto_ivl_t> v;
    // :50-53  lowQ rows of at least min_lowq_len join the merged fun windows; every row with start > extend_gate grows by
    // `extend` to the left and `extend_right` to the right (rows that start within the first extend_gate bases are left as
    // they are, as the awk does; recreate-cornetto.sh:38 uses 50000 / 40000 / 50000)
    for (int64_t i = 0; i < n_fun; ++i)
        if (fun[i].ctg >= 0 && fun[i].ctg < n_ctg) v.push_back(fun[i]);
    for (int64_t i = 0; i < n_lowq; ++i)
        if (lowq[i].ctg >= 0 && lowq[i].ctg < n_ctg && (int64_t)lowq[i].finish - lowq[i].start >= opt->min_lowq_len) v.push_back(lowq[i]);
    for (cornetto_ivl_t &x : v)
        if (x.start > opt->extend_gate) {
            x.start -= opt->extend;
            x.finish = (int32_t)std::min<int64_t>((int64_t)x.finish + opt->extend_right, INT32_MAX);
        }
    // :56  contigs longer than edge_len: their first and last edge_len bases
    for (int32_t c = 0; c < n_ctg; ++c)
        if (ctg_len[c] > opt->edge_len) {
            v.push_back(cornetto_ivl_t{c, 0, opt->edge_len});
            v.push_back(cornetto_ivl_t{c, ctg_len[c] - opt->edge_len, ctg_len[c]});
        }
    // :59  bedtools sort | bedtools merge -d merge_dist
    std::stable_sort(v.begin(), v.end(), [](const cornetto_ivl_t &a, const cornetto_ivl_t &b) { return a.ctg != b.ctg ? a.ctg < b.ctg : a.start < b.start; });
    std::vector<cornetto_ivl_t> m;
    for (const cornetto_ivl_t &x : v) {
        if (!m.empty() && m.back().ctg == x.ctg && (int64_t)x.start - m.back().finish <= opt->merge_dist) m.back().finish = std::max(m.back().finish, x.finish);
        else m.push_back(x);
    }
    // :62-66  what is left of every contig of at least min_ctg_len, in assembly order
    std::vector<cornetto_ivl_t> out;
    size_t k = 0;
    for (int32_t c = 0; c < n_ctg; ++c) {
        while (k < m.size() && m[k].ctg < c) ++k;
        size_t e = k;
        while (e < m.size() && m[e].ctg == c) ++e;
        if (ctg_len[c] >= opt->min_ctg_len && ctg_len[c] > 0) {
            int32_t pos = 0;
            for (size_t j = k; j < e; ++j) {
                const int32_t a = std::max(m[j].start, 0), b = std::min(m[j].finish, ctg_len[c]);
                if (a >= b) continue;                             // no overlap with the contig
                if (a > pos) out.push_back(cornetto_ivl_t{c, pos, a});
                pos = std::max(pos, b);
            }
            if (pos < ctg_len[c]) out.push_back(cornetto_ivl_t{c, pos, ctg_len[c]});
        }
        k = e;
    }
    cornetto_ivl_t *o = (cornetto_ivl_t *)malloc((out.size() ? out.size() : 1) * sizeof(cornetto_ivl_t));
    if (!o) return CORNETTO_E_NOMEM;
    if (!out.empty()) memcpy(o, out.data(), out.size() * sizeof(cornetto_ivl_t));
    *boring = o;
    *n_boring = (int64_t)out.size();
    return CORNETTO_OK;
}

}  // extern "C"

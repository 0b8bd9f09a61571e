// step.hip — cornetto_panel_step(): one pass of the panel path over a resident assembly and its resident coverage in ONE call on one
// handle: the totals behind the mean depth (get_depths(), src/boringbits_main.c:283-294), the thresholds (:518-519), get_regs() +
// the selection of print_fun_bits / print_boring_bits (:322-378, :425-445 / :463-481), and find() on both strands with
// process_scaffold()'s windows on the marks (src/find_telomere.c:44-74, src/telomere_windows.c:28-43,75-79).  Nothing new is computed:
// it is cornetto_cov_prepare() -> cornetto_cov_threshold() x 2 -> cornetto_cov_select_packed() -> cornetto_telo_scan() with the same
// kernels — what changes is how often the host waits.  Called one after the other those entry points synchronise five times (sums;
// selected count; telomere list totals; telomere runs; windows); here the second part is queued in one go, sized by the counts the last
// step over the same objects gave, checked afterwards: TWO synchronisations (the sums, which the thresholds — and, over several
// ranks, the all-reduce of the three sums — need on the host; everything else).  At a 1/8 share of a 3 Gbp assembly a step is about
// a millisecond and every round trip 20-30 us of it.  A count that outgrew its estimate (the first step, changed data) repeats that
// part through the exact entry point: never a truncated answer.
#include <cmath>

#include "internal.hpp"

extern "C" {

int cornetto_panel_step(cornetto_accel_t *h, const cornetto_asm_t *asm_in, const cornetto_cov_t *cov_in, const cornetto_step_opt_t *o, cornetto_sums_fn exchange, void *ctx,
                        uint64_t sums[3], int32_t thr[2], cornetto_regpk_t **recs, int64_t *n_recs, int64_t **ctg_first, cornetto_hit_t **hits, int64_t *n_hits,
                        cornetto_win_t **wins, int64_t *n_wins)
{
    if (!h || !asm_in || !cov_in || !o || !o->motif || !sums || !thr || !recs || !n_recs || !ctg_first || !hits || !n_hits || !wins || !n_wins)
        return cn_fail(h, CORNETTO_E_ARG, "panel_step: bad argument");
    cornetto_asm_t *a = const_cast<cornetto_asm_t *>(asm_in);
    cornetto_cov_t *c = const_cast<cornetto_cov_t *>(cov_in);
    *recs = nullptr; *n_recs = 0; *ctg_first = nullptr; *hits = nullptr; *n_hits = 0; *wins = nullptr; *n_wins = 0;
    CN_HIP(h, hipSetDevice(h->device));
    cn_timing_begin(h);
    // ---- the totals (one synchronisation: the thresholds need them, and across ranks the exchange)
    CN_TRY(cn_cov_prepare_impl(h, c, o->window_size, o->window_inc, sums));
    if (exchange) {
        const int xr = exchange(sums, ctx);
        if (xr != 0) return cn_fail(h, CORNETTO_E_ARG, "panel_step: the exchange of the sums failed (%d)", xr);
    }
    if (sums[2] == 0) return cn_fail(h, CORNETTO_E_ARG, "panel_step: no positions");
    const int32_t mean = (int32_t)round((double)(int64_t)sums[0] / (double)sums[2]);          // src/boringbits_main.c:293 (signed: cornetto_cov_prepare)
    thr[0] = cornetto_cov_threshold(o->low_cov, mean);                               // :518
    thr[1] = cornetto_cov_threshold(o->high_cov, mean);                              // :519
    // ---- everything else, queued in one go where the last step left its counts
    unsigned long long *p = (unsigned long long *)cn_pin(h, PIN_STEP, 32 * 8);
    if (!p) return cn_fail(h, CORNETTO_E_NOMEM, "panel_step: pinned allocation failed");
    CnCovSpec CS;
    CnTeloSpec TS;
    int rc = cn_cov_spec_queue(h, c, thr[0], thr[1], o->low_mq, o->edge_len, o->min_ctg_len, o->boring, p, &CS);
    // (queued whether or not the coverage side could be — no contig passes -m, or -w above the packed form's 32768: its exact entry point
    // answers or refuses below, the telomere scan keeps its one synchronisation)
    if (rc == CORNETTO_OK) rc = cn_telo_spec_queue(h, a, o->motif, o->thr_adj, p + 8, &TS);
    if (rc == CORNETTO_OK && (CS.queued || TS.queued)) {
        if (hipStreamSynchronize(h->stream) != hipSuccess) rc = cn_fail(h, CORNETTO_E_HIP, "panel_step: the queued scans failed");
    }
    int cov_done = 1, telo_done = 1;                  // 1: take the exact entry point
    if (rc == CORNETTO_OK) {
        cov_done = cn_cov_spec_finish(h, c, &CS, recs, n_recs, ctg_first);
        telo_done = cn_telo_spec_finish(h, a, o->motif, o->thr_adj, &TS, hits, n_hits, wins, n_wins);
        if (cov_done < 0) rc = cov_done;
        else if (telo_done < 0) rc = telo_done;
    } else {                                          // give the buffers of a half-queued attempt back
        (void)hipStreamSynchronize(h->stream);
        cn_result_quiesce(h);
        if (CS.o) cornetto_free(CS.o);
        if (TS.hits) cornetto_free(TS.hits);
    }
    if (rc == CORNETTO_OK && cov_done == 1)
        rc = cn_cov_select_packed_impl(h, c, thr[0], thr[1], o->low_mq, o->edge_len, o->min_ctg_len, o->boring, recs, n_recs, ctg_first);
    if (rc == CORNETTO_OK && telo_done == 1) rc = cn_telo_scan_impl(h, a, o->motif, o->thr_adj, hits, n_hits, wins, n_wins);
    if (rc == CORNETTO_OK) {
        cn_timing_end(h);
        return CORNETTO_OK;
    }
    cn_result_quiesce(h);
    if (*recs) cornetto_free(*recs);
    if (*ctg_first) free(*ctg_first);
    if (*hits) cornetto_free(*hits);
    if (*wins) free(*wins);
    *recs = nullptr; *ctg_first = nullptr; *hits = nullptr; *wins = nullptr;
    *n_recs = *n_hits = *n_wins = 0;
    return rc;
}

}  // extern "C"

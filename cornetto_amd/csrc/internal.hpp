// internal.hpp — pieces of cov.hip / telo.hip that step.hip (cornetto_panel_step) puts together: the same kernels and host
// logic as the public entry points, without their timing brackets and — the "spec" pairs — without their synchronisation.
// Not part of the C ABI.
#pragma once
#include "common.hpp"

// cornetto_cov_prepare() without the timing bracket (synchronises: the sums are on the host when it returns)
int cn_cov_prepare_impl(cornetto_accel_t *h, cornetto_cov_t *c, int32_t w, int32_t inc, uint64_t sums[3]);

// A packed selection (cornetto_cov_select_packed) queued WITHOUT its synchronisation: the result copy is sized by the count the last
// selection with the same parameters gave (cornetto_cov::cw_est_*), the count is checked afterwards.  cn_cov_spec_queue() leaves
// `queued` false when there is no such count (the caller takes cornetto_cov_select_packed()); cn_cov_spec_finish(), called after the
// handle's stream has been synchronised, returns 1 when the count outgrew the copy (nothing is returned: take the exact call), else
// CORNETTO_OK with the results, or an error.
struct CnCovSpec {
    bool queued = false;
    cornetto_regpk_t *o = nullptr;
    size_t n_copy = 0, cap = 0;
    unsigned long long *p_cnt = nullptr;   // pinned, 8 bytes, the caller's
    uint32_t *p_cf = nullptr;
    int64_t key = 0;
};
int cn_cov_spec_queue(cornetto_accel_t *h, cornetto_cov_t *c, int32_t lo, int32_t hi, float low_mq, int32_t edge_len, int32_t min_ctg_len, int boring,
                      unsigned long long *p_cnt, CnCovSpec *S);
int cn_cov_spec_finish(cornetto_accel_t *h, cornetto_cov_t *c, CnCovSpec *S, cornetto_regpk_t **recs, int64_t *n_recs, int64_t **ctg_first);
int cn_cov_select_packed_impl(cornetto_accel_t *h, cornetto_cov_t *c, int32_t lo, int32_t hi, float low_mq, int32_t edge_len, int32_t min_ctg_len, int boring,
                              cornetto_regpk_t **recs, int64_t *n_recs, int64_t **ctg_first);

// The fused telomere scan (cornetto_telo_scan with hits) for a motif without a border, queued without its three synchronisations: the dense
// lists, the pairing and the two result copies are sized by the counts of the last scan of the same assembly with the same motif and
// threshold (cornetto_asm::tf_est_*), the counts are checked afterwards.  Same protocol as the pair above.
struct CnTeloSpec {
    bool queued = false;
    cornetto_hit_t *hits = nullptr;        // result buffer, hit_cap records being copied
    size_t hit_cap = 0;
    size_t seg_cap[4] = {0, 0, 0, 0};
    int4 *p_wins = nullptr;                // pinned staging of the windows, win_cap of them
    size_t win_cap = 0;
    unsigned long long *p_cnt = nullptr;   // pinned, 16 x u64, the caller's: [0..3] list totals, [4] overflow | error, [8] windows
};
int cn_telo_spec_queue(cornetto_accel_t *h, cornetto_asm_t *a, const char *motif, double thr_adj, unsigned long long *p_cnt, CnTeloSpec *S);
int cn_telo_spec_finish(cornetto_accel_t *h, cornetto_asm_t *a, const char *motif, double thr_adj, CnTeloSpec *S, cornetto_hit_t **hits, int64_t *n_hits,
                        cornetto_win_t **wins, int64_t *n_wins);
int cn_telo_scan_impl(cornetto_accel_t *h, const cornetto_asm_t *a, const char *motif, double thr_adj, cornetto_hit_t **hits, int64_t *n_hits, cornetto_win_t **wins,
                      int64_t *n_wins);

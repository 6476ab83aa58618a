// qk_scan_plan.hip -- which form of the partition scan serves a batch, and with what geometry: the HOST rule of qk_scan_device
// (qk_scan.hip), moved into its own translation unit in round 4.  Everything here is host code: a table of measured
// crossovers (every constant carries the measurement it came from; tests/test_scan_form_selection_gpu.py pins the table) and
// the form feedback that corrects it per batch shape from measured call times.  The forms (same bits, all of them):
//   0  k_scan          16 x 16 MFMA tiles, one wave per workgroup -- or QUERY-SHARING workgroups (2 / 4 waves walk the same row
//                      tiles, each with its own 16-query tile)                                       qk_scan.hip
//   1  k_scan_rl       row-per-lane per-wave walk (v_mfma_f32_4x4x1, passes of 32 queries)            qk_scan_rl.hip
//   2  k_scan_rl mixed the walk for cold lists + dense workgroup items behind a bf16 prefilter for lists
//                      probed by >= hot.min queries                                                  qk_scan_rl.hip
// Replaces the choice the reference makes by SearchParams (serial_scan / batched_serial_scan / worker_scan,
// src/cpp/src/query_coordinator.cpp:659-673).
#include "qk_internal.h"
#include "qk_scan_types.h"

#include <algorithm>

// form feedback (see qk_scan_device): the form to use for this call of the shape `key`; *measure is set when the call is to be
// timed (the caller records measure->e0 / e1 around its launches)
static int qk_pick_form(qk_ctx *ctx, uint64_t key, int form_static, const bool admissible[3], qk_ctx::form_stat **measure) {
    *measure = nullptr;
    // (at most 32 entries, reserved once: the entry handed back through *measure stays where it is while the caller's launches are
    //  in flight, whatever is looked up meanwhile)
    if (ctx->form_stats.capacity() < 32) ctx->form_stats.reserve(32);
    qk_ctx::form_stat *st = nullptr;
    for (auto &f : ctx->form_stats)
        if (f.key == key) st = &f;
    if (!st) {
        if (ctx->form_stats.size() >= 32) {  // recycle the entry that has not been used for longest (its events stay)
            st = &ctx->form_stats[0];
            for (auto &f : ctx->form_stats)
                if (f.last_use < st->last_use) st = &f;
            hipEvent_t e0 = st->e0, e1 = st->e1;
            if (st->pending >= 0 && hipEventSynchronize(e1) != hipSuccess) (void)hipGetLastError();
            *st = qk_ctx::form_stat();
            st->e0 = e0;
            st->e1 = e1;
        } else {
            ctx->form_stats.emplace_back();
            st = &ctx->form_stats.back();
        }
        st->key = key;
    }
    st->calls++;
    st->last_use = ++ctx->form_clock;
    if (!st->e0 && (hipEventCreate(&st->e0) != hipSuccess || hipEventCreate(&st->e1) != hipSuccess)) {
        (void)hipGetLastError();
        return form_static;
    }
    if (st->pending >= 0) {  // harvest the measurement in flight, if the device is past it
        const hipError_t q = hipEventQuery(st->e1);
        if (q == hipSuccess) {
            float ms = 0.f;
            hipError_t eq = hipEventElapsedTime(&ms, st->e0, st->e1);
            if (ctx->form_times_set) ms = ctx->form_times[st->pending];  // (qk_ctx_set_form_times: the rule on injected figures)
            if (eq == hipSuccess && ms > 0.f) {
                const int f = st->pending;
                if (st->n[f] >= 2 && (ms > 1.5f * st->ms[f] || ms < 0.6f * st->ms[f])) {
                    // the same shape takes a very different time: the batches changed (how the queries concentrate on lists is
                    // not part of the key) -- every form is measured again, starting from this figure
                    for (int g = 0; g < 3; g++) st->n[g] = 0;
                }
                // (the first call of a form pays one-off costs -- function attributes, cold instruction cache: keep the smaller
                //  of the first two, then a running mean)
                st->ms[f] = st->n[f] == 0 ? ms : st->n[f] == 1 ? std::min(st->ms[f], ms) : 0.75f * st->ms[f] + 0.25f * ms;
                st->n[f]++;
                // a form whose FIRST call took more than twice the best figure known is not tried a second time in this
                // comparison (on a 50M index a losing form costs a 5 ms call where the winner takes 1.2)
                if (st->n[f] == 1)
                    for (int g = 0; g < 3; g++)
                        if (g != f && st->n[g] >= 2 && ms > 2.0f * st->ms[g]) st->n[f] = 2;
            } else {
                (void)hipGetLastError();
            }
            st->pending = -1;
        } else {
            (void)hipGetLastError();  // hipErrorNotReady
        }
    }
    int best = -1;
    for (int f = 0; f < 3; f++)
        if (admissible[f] && st->n[f] >= 2 && (best < 0 || st->ms[f] < st->ms[best])) best = f;
    if (st->pending >= 0) return best >= 0 ? best : form_static;  // one measurement at a time
    int next = -1;
    if (st->n[form_static] < 2) {
        next = form_static;
    } else {
        for (int f = 0; f < 3 && next < 0; f++)
            if (admissible[f] && st->n[f] < 2) next = f;
    }
    if (next < 0 && best >= 0 && st->calls % 512 == 0) {  // re-check a form that lost (the data under the index changes)
        // ... unless its last figure was more than twice the winner's: such a call is a latency spike on the serving path (5 ms
        // against 1.2 on a 50M index) and a form that far behind does not come back without the index changing -- which changes
        // the key (size bucket below) and starts a fresh comparison
        for (int t = 0; t < 3 && next < 0; t++) {
            const int f = (st->rr + t) % 3;
            if (admissible[f] && f != best && !(st->n[f] >= 1 && st->ms[f] > 2.0f * st->ms[best])) next = f;
        }
        if (next >= 0) st->rr = (next + 1) % 3;
    }
    if (next < 0 && best >= 0 && st->calls % 16 == 0) next = best;  // keep the winner's figure current (and notice a change of regime)
    if (next >= 0) {
        st->pending = next;
        *measure = st;
        return next;
    }
    return best >= 0 ? best : form_static;
}


int qk_scan_plan(qk_ctx *ctx, qk_store *s, const qk_scan_args &a, bool emit, int k, int P, int64_t npairs, ScanPlan *pl) {
    const int64_t Q = a.Q;
    // ---- geometry ----------------------------------------------------------------------------------
    const int nblk = s->nblk;
    int DB = (nblk % 8 == 0) ? 8 : (nblk % 4 == 0) ? 4 : (nblk % 2 == 0) ? 2 : 1;
    // pool capacity per query: k + slack, limited by LDS (one wave per workgroup, 160 KiB max)
    const size_t lds_budget = 160 * 1024 - 64;
    const size_t q_bytes = (size_t)nblk * 1024;
    static const int slack_min = std::max(4, qk_env_int("QK_SCAN_SLACK", 28));
#ifndef QK_SLACK_CAP
#define QK_SLACK_CAP 64
#endif
    int C = qk_round_up(k + std::max(slack_min, std::min(k, QK_SLACK_CAP)), 4);
    while ((size_t)16 * C * 12 + q_bytes > lds_budget && C > k + 4) C -= 4;
    if ((size_t)16 * C * 12 + q_bytes > lds_budget || C < k + 4 || C > 512)
        QK_FAIL(QK_ERR_UNSUPPORTED, "qk_scan: k=%d with d=%d does not fit the LDS top-k pools", k, s->d);
    // waves per workgroup: 1 unless the wave-private query tile keeps a CU below 4 resident waves; then 2 or 4 waves
    // share the tile (and, if that is what it takes to reach 4 waves, the pools give up part of their slack)
    int nw = 1;
    {
        auto waves_for = [&](int w, int cap) {
            const size_t need = q_bytes + (size_t)w * 16 * cap * 12 + 512;
            return need > 160 * 1024 ? 0 : (int)std::min<size_t>(8, w * ((160 * 1024) / need));
        };
        int best = waves_for(1, C);
        if (best < 4) {
            const int Cs = qk_round_up(k + 28, 4);
            for (int w = 2; w <= 4; w *= 2) {
                if (waves_for(w, C) > best) { best = waves_for(w, C); nw = w; }
            }
            if (best < 4 && Cs < C) {
                for (int w = 1; w <= 4; w *= 2)
                    if (waves_for(w, Cs) > best) { best = waves_for(w, Cs); nw = w; C = Cs; }
            }
        }
        {
            static const int w = qk_env_int("QK_SCAN_NW", 0);
            if ((w == 1 || w == 2 || w == 4) && waves_for(w, C) > 0) nw = w;
        }
    }
    // query-sharing workgroups: narrow rows and many queries per probed partition (a partition is otherwise streamed once
    // per 16-query tile).  The host only knows the average (pairs per present list); a batch whose queries cluster on few
    // partitions is hotter than that, so the threshold is low.
    int qshare = 0;
    {
        // measured (bench.py --nprobe 8 / 32, 1024 queries, 4096 lists): 0.589 -> 0.551 ms and 1.394 -> 1.109 ms
        static const int qs_min = qk_env_int("QK_SCAN_QSHARE_MIN", 2);
        const int64_t present = std::max<int64_t>(1, std::min<int64_t>(s->nlist, std::max<int64_t>(npairs, 1)));
        const int64_t per_list = npairs / present;
        const size_t per_wave = q_bytes + (size_t)16 * C * 12;
        if (nw == 1 && !a.per_pair && !emit && qs_min > 0 && per_list >= qs_min && have_scan_qs(DB, pick_maxch(C))) {
            // 2 waves per workgroup for moderately hot batches, 4 from ~6 queries per list on (measured: nprobe 8 -> 0.524 vs
            // 0.534 ms with 2 vs 4; nprobe 32 -> 1.199 vs 1.073 ms)
            static const int w_env = qk_env_int("QK_SCAN_QSHARE_NW", 0);
            const int w_want = w_env ? w_env : (per_list >= 6 ? 4 : 2);
            const int w = (w_want >= 4 && 4 * per_wave + 512 <= 160 * 1024) ? 4 : 2 * per_wave + 512 <= 160 * 1024 ? 2 : 1;
            if (w > 1) {
                nw = w;
                qshare = 1;
            }
        }
    }
    // Row-per-lane form (qk_scan_rl.hip, v_mfma_f32_4x4x1_16b_f32: 64 rows x 4 queries per instruction): narrow rows, k <= 32,
    // whole prepared batch.  The matrix work follows the live queries in steps of 4 and a pass over a partition serves up to
    // 32 queries from one read of its rows -- the regime where several queries of the batch probe the same partition.
    const int nw_tile = nw, qshare_tile = qshare, C_tile = C;  // the 16 x 16 tile form's geometry (form 0)
    bool use_rl = false, rl_avail = false;
    int C_walk = C;
    int64_t rl_per_list = 0;
    RlCost rlc{12, 8, 1, 16, 4, 32};
    int rl_app = 32;
    bool rl_small_pools = false;  // the row-per-lane forms only fit with pools of k + 16: admissible, measured by the feedback, not the static choice
    int rl_waves = 4;
    {
        // Measured (10M x 128, 1024 queries, k = 10; k_scan ms old form -> this form): nprobe 1: 0.252 -> 0.246, 2: 0.324 -> 0.313,
        // 4: 0.404 -> 0.362, 8: 0.488 -> 0.439, 16: 0.694 -> 0.616, 32: 0.958 -> 1.02 (MFMA-bound: 5 us of chain per chunk);
        // low-intrinsic-dimension corpus, nprobe 16: 0.862 -> 0.778 (0.80 of the HBM peak).  So: whenever supported, except
        // when the batch averages rl_max or more probing queries per list -- and except nprobe = 1, where the dynamic tail's
        // extra range boundaries cost the merge more (2211 -> 5840 records, 10 -> 23 us) than the scan gains (252 -> 247 us).
        static const int rl_env = qk_env_int("QK_SCAN_RL", -1);  // -1 auto, 0 never, 1 whenever supported
        static const int rl_max = qk_env_int("QK_SCAN_RL_MAX", 6);
        static const int rl_h0 = qk_env_int("QK_SCAN_RL_H0", 12);
        static const int rl_h1 = qk_env_int("QK_SCAN_RL_H1", 8);
        static const int rl_e = qk_env_int("QK_SCAN_RL_E", 1);
        static const int rl_ovh = qk_env_int("QK_SCAN_RL_OVH", 16);
        static const int rl_m = qk_env_int("QK_SCAN_RL_M", 4);
        const int64_t present = std::max<int64_t>(1, std::min<int64_t>(s->nlist, std::max<int64_t>(npairs, 1)));
        const int64_t per_list = npairs / present;
        // Pass width: a partition probed by more queries than a pass holds is streamed once per pass (1024 waves x 1-2 MB
        // partitions: nothing survives in the L2s); the width is bounded by the wave's quarter of the LDS (queries: qb x dpad
        // floats, + pools).  32 queries with pools of k + 32 entries is the product setting.  Measured against 48 queries with
        // pools of k + 16 entries and appends by quarter waves (the widest that fits at d = 128, k = 10; bytes-weighted re-reads
        // on the bench mixture at 32 / 48 / 64 queries per pass: nprobe 8 1.17 / 1.09 / 1.05, nprobe 16 1.47 / 1.28 / 1.16):
        // kernel ms nprobe 4 0.381 -> 0.377, 8 0.454 -> 0.446, 16 0.602 -> 0.606, 32 0.985 -> 0.985, merge +2 us each
        // -- the re-reads are not what the launch waits for.  The wider pass stays behind the probe switch.
        static const int rl_qb_env = qk_env_int("QK_SCAN_RL_QB", 0);  // probe: 32 ... 64 (0: 32)
        int qb = 32, C_rl = std::min(64, qk_round_up(k + 32, 4));
        static const int rl_waves_env = qk_env_int("QK_SCAN_RL_WAVES", 4);  // probe: waves per CU (3 leaves room for 64-query passes at d = 128)
        rl_waves = std::min(4, std::max(1, rl_waves_env));
        auto fits = [&](int q, int c) { return rl_waves * ((qk_scan_rl_lds_per_wave(nblk, c, q) + 15) & ~(size_t)15) <= (size_t)160 * 1024; };
        if (rl_qb_env > 32) {
            const int c16 = qk_round_up(k + 16, 4);
            for (int q = std::min(rl_qb_env & ~3, QK_RL_QB_MAX); q > 32; q -= 4)
                if (fits(q, c16)) {
                    qb = q;
                    C_rl = c16;
                    rl_app = 16;
                    break;
                }
        }
        // k = 29 ... 32 at d = 128: 32 pools of k + 32 entries do not fit next to the pass's queries (40.5 KB against the wave's
        // 40 KB) and these k fell back to the tile form (10M x 128, nprobe 16: k = 28 0.70 ms, k = 29 0.83, k = 32 0.88): pools of
        // k + 16 entries, appended to by quarter waves, do
        static const int app16_env = qk_env_int("QK_SCAN_RL_APP16", 0);  // probe: quarter-wave appends for every k
        if (qb == 32 && rl_app == 32 && (!fits(qb, C_rl) || app16_env) && fits(qb, qk_round_up(k + 16, 4))) {
            C_rl = qk_round_up(k + 16, 4);
            rl_app = 16;
            rl_small_pools = !app16_env;
        }
        rlc = RlCost{std::max(1, rl_h0), std::max(1, rl_h1), std::max(0, rl_e), std::max(0, rl_ovh), std::max(1, rl_m), qb};
        // (4 waves per CU, each with its own LDS copy of the pass's queries and its pools: d = 128 with k = 32 does not fit)
        const bool rl_fits = fits(qb, C_rl);
        // (per_pair -- the rounds of the recall-target search -- takes these forms too: every pool IS a (query, list) pair's, the bound
        //  is the caller's and is never published; round 4: its rounds ran 670 us each in the plain tile form)
        const bool rl_ok = nblk <= 8 && k <= 32 && rl_fits && !emit && npairs > 0 && ctx->qprep_xp4 != nullptr &&
                           a.xq4 == (const float4 *)ctx->qprep;
        use_rl = rl_ok && (rl_env == 1 || (rl_env < 0 && per_list < rl_max && P > 1));
        rl_avail = rl_ok && rl_env != 0 && P > 1;
        C_walk = C_rl;
        rl_per_list = per_list;
    }
    // Mixed work sequence (qk_scan_rl.hip, HOT form): lists probed by >= hot.min queries of the batch become dense items on
    // v_mfma_f32_16x16x4_f32 claimed by whole workgroups; the block width hq is what the workgroup's LDS (the four waves' slices
    // of the per-wave form together) holds next to one pool of C entries per query.
    HotCost hot{0, 0, 0, 0, 0, 0, 0, 0}, hot_cand{0, 0, 0, 0, 0, 0, 0, 0};
    bool mixed_avail = false, mixed_static = false;
    int C_mixed = C;
    {
        // Measured (10M x 128, 1024 queries, k = 10, scripts/nprobe_sweep.py; kernel ms per-wave walk alone -> mixed, lists with
        // >= 13 probing queries hot): skewed mixture nprobe 8 / 16 / 32 / 64: 0.461 / 0.623 / 1.027 / 1.808 -> 0.474 / 0.512 / 0.632 /
        // 0.878; uniformly probed corpus nprobe 8 / 16 / 32 / 64: 0.683 / 0.798 / 0.960 / 1.002 -> 0.686 / 0.827 / 0.841 / 1.048 (the
        // query-sharing form of k_scan, which used to take over from 6 probing queries per list on: 1.019 / 1.177 at nprobe 32 / 64).
        // At nprobe 2 / 4 (under two probing queries per list on average) the few hot lists are home lists -- every other row tile
        // has a true candidate -- and the items only add their fixed costs: 0.344 / 0.382 -> 0.413 / 0.447; the hot form stays off.
        // Round 4: the launch must END in the per-wave walk's fine-grained dynamic tail, not in items (47 us apiece: the wave end
        // times of round 3 spread from 414 to 545 us around a mean of 445 at nprobe 16, the slowest workgroups all inside an item).
        // The hot-first workgroups' share of the grid is therefore taken at 3x the items' modelled cost (ScanParams::hot_first_pct),
        // half of the walk is handed out dynamically, and with that balance the thresholds moved: hot from 18 probing queries
        // (13), mixed form from TWO probing queries per list on average (three).  Kernel ms before -> after, skewed mixture nprobe
        // 8 / 10 / 12 / 16 / 32 / 64: 0.459 / 0.484 / 0.503 / 0.516 / 0.602 / 0.806 -> 0.399 / 0.411 / 0.423 / 0.446 / 0.555 / 0.773;
        // uniformly probed corpus 16 / 32 / 64: 0.832 / 0.851 / 1.003 -> 0.818 / 0.865 / 1.012 (scripts/gpu_knobs.sh).
        static const int hot_min = qk_env_int("QK_SCAN_HOT_MIN", 18);   // 0: per-wave walk only
        static const int hot_unit = qk_env_int("QK_SCAN_HOT_UNIT", 300);
        static const int hot_w10 = qk_env_int("QK_SCAN_HOT_W10", 3);
        static const int hot_ht10 = qk_env_int("QK_SCAN_HOT_HT10", 30);
        static const int hot_ovh = qk_env_int("QK_SCAN_HOT_OVH", 64);
        static const int hot_hq = qk_env_int("QK_SCAN_HOT_HQ", 128);
        // (from THREE probing queries per list on average: at two -- nprobe 8-11 on the bench index -- the per-wave walk alone is
        //  as fast or faster on both corpora: skewed mixture nprobe 8 / 10 / 12 / 14 walk 0.447 / 0.481 / 0.515 / 0.565 ms, mixed
        //  0.475 / 0.480 / 0.490 / 0.498; uniformly probed corpus 0.683 / 0.714 / 0.738 / 0.763 against 0.692 / 0.747 / 0.785 / 0.809 --
        //  there the mixed form only pays from nprobe ~24 on, a skew the host cannot see; the rule follows the skewed case)
        // (the form used to start at THREE probing queries per list on average -- an index-wide mean that says nothing about a few
        //  very hot lists.  With the balance above it is at least as fast as the plain walk from one on -- mixture nprobe 2 / 4 / 6:
        //  0.339 / 0.387 / 0.418 -> 0.333 / 0.369 / 0.386 ms, uniformly probed corpus 2 / 4 / 6 / 8: 0.357 / 0.522 / 0.630 / 0.685 ->
        //  0.352 / 0.527 / 0.637 / 0.700 (no hot list there: the walk with half its sequence dynamic) -- so whether a list is hot is
        //  decided per list on the device, by the grouping stage that counts its probing queries, for every batch with nprobe > 1)
        static const int hot_per_list = qk_env_int("QK_SCAN_HOT_PER_LIST", 1);
        static const int hot_min_rows = qk_env_int("QK_SCAN_HOT_MIN_ROWS", 512);
        // Short lists: on the configs[0] shape (1M x 128 in 1024 lists of ~1000 rows, nprobe 10) items are 61 row tiles long and
        // their fixed costs show -- 256 / 1024 queries: scan 85 / 165 us with the round-2 forms (per-wave walk / query-sharing tile
        // form), 143 / 236 us mixed -- so the mixed form serves indexes whose lists average >= 1400 rows, the round-2 rule the rest
        static const int hot_mean_rows = qk_env_int("QK_SCAN_HOT_MEAN_ROWS", 1400);
        const int64_t mean_rows = s->ntotal / std::max<int64_t>(1, s->n_nonempty);
        const bool long_lists = mean_rows >= hot_mean_rows;
        static const int rl_env2 = qk_env_int("QK_SCAN_RL", -1);
        // (the walk part of the mixed form has the per-wave form's pools: k + 32 entries, or k + 16 with quarter-wave appends)
        const int C_rl2 = rl_app == 16 ? qk_round_up(k + 16, 4) : std::min(64, qk_round_up(k + 32, 4));
        const bool rl_possible = nblk <= 8 && k <= 32 && rl_waves == 4 && rlc.qb == 32 && !emit && npairs >= 1024 && P > 1 &&
                                 ctx->qprep_xp4 != nullptr && a.xq4 == (const float4 *)ctx->qprep && rl_env2 != 0 &&
                                 4 * ((qk_scan_rl_lds_per_wave(nblk, C_rl2, rlc.qb) + 15) & ~(size_t)15) <= (size_t)160 * 1024;
        if (rl_possible && hot_min > 0 && long_lists) {
            const size_t per_wave = std::max<size_t>((qk_scan_rl_lds_per_wave(nblk, C_rl2, rlc.qb) + 15) & ~(size_t)15, (size_t)(160 * 1024) / 4 - 512) & ~(size_t)15;
            // (pools of k + 22 entries: appends come four at a time at most, and a block's pools share the LDS with its query
            //  tiles in fp32 and in bf16)
            const int C_hot = std::min(64, qk_round_up(k + 22, 4));
            int hq = std::min(128, std::max(16, hot_hq)) & ~15;
            while (hq >= 32 && qk_scan_hot_lds(nblk, C_hot, hq) > 4 * per_wave) hq -= 16;
            if (hq >= 32) {
                hot_cand = HotCost{std::max(hot_min, 1), hq, std::max(16, hot_unit), std::max(1, hot_w10), std::max(1, hot_ht10), std::max(0, hot_ovh), C_hot, std::max(16, hot_min_rows)};
                // the mixed form serves every sharing level: by the static rule it replaces the query-sharing form of k_scan too
                mixed_avail = true;
                mixed_static = rl_per_list >= hot_per_list;
                C_mixed = C_rl2;
            }
        }
    }
    // ---- form feedback --------------------------------------------------------------------------------------------------
    // The static rule above knows index-wide means; how the queries of THIS batch concentrate on lists it cannot see, and that
    // decides: 1024 queries around a few neighbouring clusters of a 1M x 128 index in 400 lists (the "skewed" batches of the
    // dynamic workload, BASELINE.json configs[4]): scan 235 us on the query-sharing tile form, 760 us mixed (every list hot, most
    // row tiles true candidates: the prefilter skips nothing and the items' pools are appended to under locks); the same batch
    // shape spread over the whole 10M x 128 bench index: 403 us mixed, 478 us on the walk (scripts/skew_probe.py).  So the forms
    // a shape admits are MEASURED: whole calls (grouping + scan + merge) between two HIP events, read back at a later call
    // without synchronising; every admissible form twice, then the fastest, the others re-checked every 512 calls.  All forms
    // give the same bits (the parity suites run each of them), so which one answers is invisible to the caller.
    // (k = 29 ... 32 at d = 128: at nprobe 8 the tile form is still the faster one -- 0.625 against the mixed form's 0.66 ms at k = 32 --,
    //  from nprobe 16 on the mixed form is: 0.83 -> 0.66 at k = 29, 0.88 -> 0.86 at k = 32; the feedback decides, the static rule stays)
    const int form_static = rl_small_pools ? 0 : mixed_static ? 2 : use_rl ? 1 : 0;
    int form = form_static;
    qk_ctx::form_stat *fmeasure = nullptr;
    {
        const bool admissible[3] = {true, rl_avail, mixed_avail};
        if (ctx->form_feedback && !emit && npairs >= 1024 && P > 1 && (rl_avail || mixed_avail)) {
            // (size bucket: the store's row count to a quarter of a power of two and its list count to a power of two -- a store that
            //  grew, shrank or was re-partitioned by maintenance is a new shape with a fresh comparison, not a stale winner)
            int sb = 0, lb = 0;
            for (int64_t t = s->ntotal; t > 7; t >>= 1) sb++;
            const int sq = (int)((s->ntotal >> (sb > 1 ? sb - 1 : 0)) & 3);  // two bits below the leading one
            for (int64_t t = s->nlist; t > 1; t >>= 1) lb++;
            const uint64_t key = (s->uid * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(Q >> 6) << 40) ^ ((uint64_t)P << 24) ^ ((uint64_t)k << 8) ^ (uint64_t)a.metric ^
                                 (a.per_pair ? 1ull << 63 : 0ull) ^ ((uint64_t)(sb * 4 + sq) * 0xD6E8FEB86659FD93ull) ^ ((uint64_t)lb << 56) ^
                                 ((uint64_t)a.form_salt * 0xA24BAED4963EE407ull);
            form = qk_pick_form(ctx, key, form_static, admissible, &fmeasure);
        }
    }
    if (fmeasure && hipEventRecord(fmeasure->e0, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        fmeasure->pending = -1;
        fmeasure = nullptr;
    }
    if (form == 2) {
        hot = hot_cand;
        use_rl = true;
        nw = 1;
        qshare = 0;
        C = C_mixed;
    } else if (form == 1) {
        use_rl = true;
        nw = 1;
        qshare = 0;
        C = C_walk;
    } else {
        use_rl = false;
        nw = nw_tile;
        qshare = qshare_tile;
        C = C_tile;
    }
    pl->DB = DB;
    pl->C = C;
    pl->nw = nw;
    pl->qshare = qshare;
    pl->use_rl = use_rl;
    pl->form = form;
    pl->rl_per_list = rl_per_list;
    pl->rlc = rlc;
    pl->rl_app = rl_app;
    pl->rl_waves = rl_waves;
    pl->hot = hot;
    pl->measure = fmeasure;
    return QK_OK;
}

// qk_api.hip -- C-ABI search entry points (include/quake_hip.h): marshalling between caller memory and the
// device pipeline of qk_scan.hip.  No arithmetic lives here.
#include "qk_internal.h"

#include <algorithm>
#include <cstring>

#ifndef QK_HOST_PINNED_IO
#define QK_HOST_PINNED_IO 1
#endif

namespace {

struct Staged {  // device-side views of the caller's buffers for one call
    const float *x = nullptr;
    const int64_t *pids = nullptr;
    int64_t *out_ids = nullptr;
    float *out_dist = nullptr;
};

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// read back phase timings (events) and device scalars after the stream has drained
int qk_finish_timing(qk_ctx *ctx, qk_store *s, qk_timing *t, bool have_coarse, int scan_ev_base) {
    if (!t) return QK_OK;
    QK_HIP(hipStreamSynchronize(ctx->stream));
    const int32_t *hs = (const int32_t *)ctx->pinned;
    t->n_items = hs[0];  // active partitions
    t->partitions_scanned = hs[7];  // (query, partition) pairs that reached a present, non-empty partition
    int64_t rows_unique;
    memcpy(&rows_unique, hs + 2, sizeof(int64_t));
    t->scan_bytes = rows_unique * (int64_t)s->d * 4;
    if (ctx->timing) {
        float ms = 0.f;
        if (have_coarse) {
            QK_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[3]));
            t->coarse_ms = ms;
        }
        QK_HIP(hipEventElapsedTime(&ms, ctx->ev[scan_ev_base + 0], ctx->ev[scan_ev_base + 1]));
        t->group_ms = ms;
        QK_HIP(hipEventElapsedTime(&ms, ctx->ev[scan_ev_base + 1], ctx->ev[scan_ev_base + 2]));
        t->scan_ms = ms;
        QK_HIP(hipEventElapsedTime(&ms, ctx->ev[scan_ev_base + 2], ctx->ev[scan_ev_base + 3]));
        t->merge_ms = ms;
        QK_HIP(hipEventElapsedTime(&ms, ctx->ev[have_coarse ? 0 : scan_ev_base], ctx->ev[scan_ev_base + 3]));
        t->total_ms = ms;
    }
    return QK_OK;
}

namespace {
int check_metric(int metric) {
    if (metric != QK_METRIC_L2 && metric != QK_METRIC_IP) QK_FAIL(QK_ERR_INVALID, "Metric type not supported");
    return QK_OK;
}
}  // namespace

// scan with every pointer in `mem`; parent == nullptr && pids == nullptr -> all lists.
// defer_finish (device memory only; the device group of qk_group.hip): nothing here waits for the stream -- the scalars behind
// `timing` are requested but read later, by qk_finish_timing, once every member of the group has been enqueued.
int qk_run_search(qk_ctx *ctx, qk_store *parent, qk_store *s, const float *x, int64_t Q, const int64_t *pids, int P, int nprobe,
                  int k, int metric, int64_t *out_ids, float *out_dist, int mem, qk_timing *timing, bool coarse_only,
                  bool defer_finish, int64_t *probed_out) {
    QK_TRY(qk_check_overflow(ctx));  // a record-buffer overflow of an earlier launch is reported by the next call
    QK_HIP(hipSetDevice(ctx->device));
    if (timing) memset(timing, 0, sizeof(*timing));
    if (Q <= 0) return QK_OK;
    const int d = s->d;
    const bool use_parent = parent != nullptr;
    int kk = 0;
    if (use_parent) {
        if (parent->d != d) QK_FAIL(QK_ERR_INVALID, "parent store dimension %d != store dimension %d", parent->d, d);
        kk = (int)std::min<int64_t>(nprobe, parent->ntotal);  // query_coordinator.cpp:641
        if (kk > QK_MAX_NPROBE) QK_FAIL(QK_ERR_UNSUPPORTED, "nprobe=%d exceeds QK_MAX_NPROBE=%d", kk, QK_MAX_NPROBE);
    }
    if (!(use_parent && !coarse_only && kk > 0)) probed_out = nullptr;  // (nothing probed: the buffer may have no columns at all)
    const int Ps = coarse_only ? 0 : (use_parent ? kk : P);
    const int kout = coarse_only ? kk : k;
    // ---- stage caller buffers ------------------------------------------------------------------------
    size_t bx = al256((size_t)Q * d * 4), bp = al256((size_t)Q * std::max(Ps, 1) * 8);
    size_t bi = al256((size_t)Q * std::max(kout, 1) * 8), bd = al256((size_t)Q * std::max(kout, 1) * 4);
    Staged sv;
    // host buffers through the context's pinned staging (qk_internal.h, pin_io) up to 64 MB per call; larger calls transfer directly
    char *pin = nullptr;
    size_t pin_x = 0, pin_p = 0, pin_i = 0, pin_d = 0, pin_pr = 0;
    if (mem == QK_MEM_HOST && QK_HOST_PINNED_IO) {
        const size_t need = bx + bp + bi + bd + bp + 256;
        if (need <= ((size_t)64 << 20)) {
            if (need > ctx->pin_io_cap) {
                QK_HIP(hipStreamSynchronize(ctx->stream));
                if (ctx->pin_io) QK_HIP(hipHostFree(ctx->pin_io));
                ctx->pin_io = nullptr;
                ctx->pin_io_cap = 0;
                if (hipHostMalloc((void **)&ctx->pin_io, need + need / 2, hipHostMallocDefault) == hipSuccess) ctx->pin_io_cap = need + need / 2;
                else (void)hipGetLastError();
            }
            if (need <= ctx->pin_io_cap) {
                pin = ctx->pin_io;
                pin_x = 0;
                pin_p = bx;
                pin_i = bx + bp;
                pin_d = bx + bp + bi;
                pin_pr = bx + bp + bi + bd;
            }
        }
    }
    if (mem == QK_MEM_HOST) {
        QK_TRY(qk_stage_reserve(ctx, bx + bp + bi + bd + 256));
        char *b = ctx->stage;
        const void *hx = x;
        if (pin) {
            memcpy(pin + pin_x, x, (size_t)Q * d * 4);
            hx = pin + pin_x;
        }
        QK_HIP(hipMemcpyAsync(b, hx, (size_t)Q * d * 4, hipMemcpyHostToDevice, ctx->stream));
        sv.x = (const float *)b;
        b += bx;
        if (!use_parent && pids && Ps > 0) {
            const void *hp = pids;
            if (pin) {
                memcpy(pin + pin_p, pids, (size_t)Q * Ps * 8);
                hp = pin + pin_p;
            }
            QK_HIP(hipMemcpyAsync(b, hp, (size_t)Q * Ps * 8, hipMemcpyHostToDevice, ctx->stream));
        }
        sv.pids = (const int64_t *)b;
        b += bp;
        sv.out_ids = (int64_t *)b;
        b += bi;
        sv.out_dist = out_dist ? (float *)b : nullptr;  // (no distances asked for: none computed -- the kernels take NULL)
    } else {
        sv.x = x;
        sv.out_ids = out_ids;
        sv.out_dist = out_dist;
        if (use_parent && !coarse_only && probed_out) {
            sv.pids = probed_out;  // (qk_search_tracked: the nearest-centroid step writes the caller's [Q][nprobe] buffer, the scan reads it)
        } else if (use_parent && !coarse_only) {
            QK_TRY(qk_stage_reserve(ctx, bp + 256));
            sv.pids = (const int64_t *)ctx->stage;
        } else {
            sv.pids = pids;
        }
    }
    // small batches: the whole search in one launch (qk_small.hip) -- no prep / group / seed / merge launches
    if (use_parent && !coarse_only && kk > 0 && !probed_out && qk_small_supported(ctx, parent, s, Q, kk, k)) {
        const bool tm = ctx->timing && timing;
        // one event pair around the one kernel (an event record costs the stream a few microseconds): ev[4], ev[7] per call,
        // the scan pair of a deferred group otherwise
        qk_phase_events pe;
        QK_TRY(pe.begin(ctx, false, 4));
        for (int i : {0, 3})
            if (pe.dev[i]) {
                ctx->ev_free.push_back(pe.dev[i]);
                pe.dev[i] = nullptr;
            }
        if (tm) QK_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
        QK_TRY(pe.mark(1));
        ctx->last_scan_kernel = "k_search_small";
        QK_TRY(qk_search_small_device(ctx, parent, s, sv.x, Q, kk, k, metric, sv.out_ids, sv.out_dist, !ctx->squared_l2));
        QK_TRY(pe.mark(2));
        QK_TRY(pe.mark(3));  // (no event of its own: parks the group)
        if (tm) QK_HIP(hipEventRecord(ctx->ev[7], ctx->stream));
        if (mem == QK_MEM_HOST) {
            if (out_ids) QK_HIP(hipMemcpyAsync(pin ? (void *)(pin + pin_i) : (void *)out_ids, sv.out_ids, (size_t)Q * kout * 8, hipMemcpyDeviceToHost, ctx->stream));
            if (out_dist) QK_HIP(hipMemcpyAsync(pin ? (void *)(pin + pin_d) : (void *)out_dist, sv.out_dist, (size_t)Q * kout * 4, hipMemcpyDeviceToHost, ctx->stream));
            QK_HIP(hipStreamSynchronize(ctx->stream));
            if (pin && out_ids) memcpy(out_ids, pin + pin_i, (size_t)Q * kout * 8);
            if (pin && out_dist) memcpy(out_dist, pin + pin_d, (size_t)Q * kout * 4);
        }
        if (timing) {  // one kernel: everything is "scan"; the pair / byte counters are not collected on this path
            timing->partitions_scanned = Q * (int64_t)kk;
            timing->n_items = -1;  // (tells a deferred finish that this call left no scalars behind)
            if (defer_finish) return QK_OK;
            timing->n_items = 0;
            QK_HIP(hipStreamSynchronize(ctx->stream));
            if (tm) {
                float ms = 0.f;
                QK_HIP(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[7]));
                timing->scan_ms = timing->total_ms = ms;
            }
        }
        return QK_OK;
    }
    const float4 *xq4 = nullptr;
    const float *xn = nullptr;
    // (the launch is left pending when a nearest-centroid search follows: k_dense_argmin prepares the queries while it stages them;
    //  every other consumer launches the prep kernel first -- qk_prep_flush)
    QK_TRY(qk_prep_queries(ctx, sv.x, Q, d, &xq4, &xn, coarse_only ? 0 : qk_scan_zero_bytes((int64_t)s->parts.size(), Q),
                           use_parent && kk == 1));
    const unsigned long long *packed = nullptr;
    // ---- coarse --------------------------------------------------------------------------------------
    if (use_parent && kk <= 0 && coarse_only) return qk_prep_flush(ctx);
    if (use_parent && kk > 0) {
        qk_scan_args ca;
        ca.x = sv.x;
        ca.xq4 = xq4;
        ca.xn = xn;
        ca.Q = Q;
        ca.all_lists = true;
        ca.k = kk;
        ca.metric = metric;
        ca.out_ids = coarse_only ? sv.out_ids : (int64_t *)sv.pids;
        ca.out_dist = coarse_only ? sv.out_dist : nullptr;
        ca.record_events = timing != nullptr;
        if (!coarse_only && kk == 1 && k <= QK_MAX_K && !probed_out) ca.packed_out = &packed;  // nprobe = 1: see qk_scan_args::pids_packed
        QK_TRY(qk_scan_device(ctx, parent, ca, coarse_only ? timing : nullptr, 0));
    }
    // ---- scan ------------------------------------------------------------------------------------------
    if (!coarse_only) {
        qk_scan_args sa;
        sa.x = sv.x;
        sa.xq4 = xq4;
        sa.xn = xn;
        sa.Q = Q;
        sa.k = k;
        sa.metric = metric;
        sa.out_ids = sv.out_ids;
        sa.out_dist = sv.out_dist;
        if (!use_parent && !pids) {
            sa.all_lists = true;
        } else {
            sa.pids = sv.pids;
            sa.P = Ps;
            if (packed) {
                sa.pids = nullptr;
                sa.pids_packed = packed;
            }
        }
        sa.sqrt_l2 = !ctx->squared_l2;
        if (use_parent && kk <= 0) {  // empty parent: nothing to probe -> padding only
            QK_HIP(hipMemsetAsync((void *)sv.pids, 0xFF, (size_t)Q * 8, ctx->stream));
            sa.P = 1;
        }
        QK_TRY(qk_scan_device(ctx, s, sa, timing, 4));
    }
    // ---- results back ------------------------------------------------------------------------------------
    if (mem == QK_MEM_HOST) {
        const bool pr = probed_out && use_parent && !coarse_only && kk > 0;
        if (out_ids) QK_HIP(hipMemcpyAsync(pin ? (void *)(pin + pin_i) : (void *)out_ids, sv.out_ids, (size_t)Q * kout * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (out_dist) QK_HIP(hipMemcpyAsync(pin ? (void *)(pin + pin_d) : (void *)out_dist, sv.out_dist, (size_t)Q * kout * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (pr) QK_HIP(hipMemcpyAsync(pin ? (void *)(pin + pin_pr) : (void *)probed_out, sv.pids, (size_t)Q * kk * 8, hipMemcpyDeviceToHost, ctx->stream));
        QK_HIP(hipStreamSynchronize(ctx->stream));
        if (pin && out_ids) memcpy(out_ids, pin + pin_i, (size_t)Q * kout * 8);
        if (pin && out_dist) memcpy(out_dist, pin + pin_d, (size_t)Q * kout * 4);
        if (pin && pr) memcpy(probed_out, pin + pin_pr, (size_t)Q * kk * 8);
    }
    if (defer_finish) return QK_OK;
    if (timing) QK_TRY(qk_finish_timing(ctx, coarse_only ? parent : s, timing, use_parent && !coarse_only, coarse_only ? 0 : 4));
    if (timing || mem == QK_MEM_HOST) QK_TRY(qk_check_overflow(ctx));  // these paths have synchronised: report now
    return QK_OK;
}

extern "C" {

int qk_coarse(qk_ctx *ctx, qk_store *parent, const float *x, int64_t Q, int nprobe, int metric, int64_t *out_pids, float *out_dist,
              int mem) {
    if (!ctx || !parent || (Q > 0 && (!x || !out_pids))) QK_FAIL(QK_ERR_INVALID, "qk_coarse: null argument");
    if (nprobe <= 0) QK_FAIL(QK_ERR_INVALID, "qk_coarse: nprobe must be positive");
    QK_TRY(check_metric(metric));
    // nprobe >= 2 over a huge batch (the maintenance policy ranks 10^5 ... 10^6 rows of its delete candidates against the parent): the
    // prefiltered selection (qk_dense_pf.hip) takes batches of up to 65536 queries, beyond it the call fell to the key-matrix path --
    // 262144 rows x 19920 centroids: 26.5 ms against 4 x 1.2 ms in pieces of 65536
    const int64_t piece = 65536;
    if (Q > piece && nprobe >= 2) {
        const int64_t kk = std::min<int64_t>(nprobe, parent->ntotal);
        for (int64_t q0 = 0; q0 < Q; q0 += piece) {
            const int64_t qn = std::min(piece, Q - q0);
            QK_TRY(qk_run_search(ctx, parent, parent, x + q0 * parent->d, qn, nullptr, 0, nprobe, 0, metric, out_pids + q0 * kk,
                                 out_dist ? out_dist + q0 * kk : nullptr, mem, nullptr, true, false));
        }
        return QK_OK;
    }
    return qk_run_search(ctx, parent, parent, x, Q, nullptr, 0, nprobe, 0, metric, out_pids, out_dist, mem, nullptr, true, false);
}

int qk_scan(qk_ctx *ctx, qk_store *s, const float *x, int64_t Q, const int64_t *pids, int P, int k, int metric, int64_t *out_ids,
            float *out_dist, int mem, qk_timing *timing) {
    if (!ctx || !s || (Q > 0 && (!x || !out_ids))) QK_FAIL(QK_ERR_INVALID, "qk_scan: null argument");
    if (P < 0 || (P > 0 && !pids)) QK_FAIL(QK_ERR_INVALID, "qk_scan: bad partition id list");
    if (k <= 0) QK_FAIL(QK_ERR_INVALID, "qk_scan: k must be positive");
    QK_TRY(check_metric(metric));
    if (mem == QK_MEM_HOST && pids) {
        // the reference throws "List does not exist" from get_codes (dynamic_inverted_list.cpp:76-82)
        for (int64_t i = 0; i < Q * (int64_t)P; i++) {
            int64_t p = pids[i];
            if (p < 0) continue;
            if (p >= (int64_t)s->parts.size() || !s->parts[p].present)
                QK_FAIL(QK_ERR_NOT_FOUND, "List does not exist in get_codes (list %lld)", (long long)p);
        }
    }
    // P == 0: zero partitions to scan -> padded output (query_coordinator.cpp:459-497); the pipeline handles it
    static const int64_t dummy = -1;
    const int64_t *pp = pids;
    int PP = P;
    if (P == 0) {
        if (mem == QK_MEM_HOST) {
            // build a [Q][1] list of -1
            std::vector<int64_t> neg((size_t)std::max<int64_t>(Q, 1), -1);
            return qk_run_search(ctx, nullptr, s, x, Q, neg.data(), 1, 0, k, metric, out_ids, out_dist, mem, timing, false, false);
        }
        (void)dummy;
        QK_FAIL(QK_ERR_INVALID, "qk_scan: P == 0 needs host memory (pass a [Q][1] list of -1 instead)");
    }
    return qk_run_search(ctx, nullptr, s, x, Q, pp, PP, 0, k, metric, out_ids, out_dist, mem, timing, false, false);
}

int qk_search(qk_ctx *ctx, qk_store *parent, qk_store *s, const float *x, int64_t Q, int nprobe, int k, int metric,
              int64_t *out_ids, float *out_dist, int mem, qk_timing *timing) {
    if (!ctx || !s || (Q > 0 && (!x || !out_ids))) QK_FAIL(QK_ERR_INVALID, "qk_search: null argument");
    if (k <= 0) QK_FAIL(QK_ERR_INVALID, "qk_search: k must be positive");
    if (parent && nprobe <= 0) QK_FAIL(QK_ERR_INVALID, "qk_search: nprobe must be positive");
    QK_TRY(check_metric(metric));
    return qk_run_search(ctx, parent, s, x, Q, nullptr, 0, nprobe, k, metric, out_ids, out_dist, mem, timing, false, false);
}

int qk_search_tracked(qk_ctx *ctx, qk_store *parent, qk_store *s, const float *x, int64_t Q, int nprobe, int k, int metric,
                      int64_t *out_ids, float *out_dist, int64_t *out_probed, int mem, qk_timing *timing) {
    if (!ctx || !s || !parent || (Q > 0 && (!x || !out_ids || !out_probed))) QK_FAIL(QK_ERR_INVALID, "qk_search_tracked: null argument");
    if (k <= 0) QK_FAIL(QK_ERR_INVALID, "qk_search_tracked: k must be positive");
    if (nprobe <= 0) QK_FAIL(QK_ERR_INVALID, "qk_search_tracked: nprobe must be positive");
    QK_TRY(check_metric(metric));
    return qk_run_search(ctx, parent, s, x, Q, nullptr, 0, nprobe, k, metric, out_ids, out_dist, mem, timing, false, false, out_probed);
}

int qk_ctx_set_squared_l2(qk_ctx *ctx, int enabled) {
    if (!ctx) QK_FAIL(QK_ERR_INVALID, "qk_ctx_set_squared_l2: ctx is null");
    ctx->squared_l2 = enabled != 0;
    return QK_OK;
}

int qk_merge_topk(qk_ctx *ctx, const int64_t *in_ids, const float *in_key, int G, int64_t Q, int k, int metric, int64_t *out_ids,
                  float *out_dist) {
    if (!ctx || (Q > 0 && (!in_ids || !in_key || !out_ids))) QK_FAIL(QK_ERR_INVALID, "qk_merge_topk: null argument");
    if (G <= 0 || k <= 0) QK_FAIL(QK_ERR_INVALID, "qk_merge_topk: bad sizes");
    QK_TRY(check_metric(metric));
    QK_HIP(hipSetDevice(ctx->device));
    return qk_merge_topk_device(ctx, in_ids, in_key, G, Q, k, metric, out_ids, out_dist, true);
}

size_t qk_topk_block_bytes(int64_t per, int k) { return (per < 0 || k <= 0) ? 0 : qk_topk_block_bytes_(per, k); }

int qk_pack_topk(qk_ctx *ctx, const int64_t *ids, const float *key, int G, int64_t per, int k, void *packed) {
    if (!ctx || (per > 0 && (!ids || !key || !packed))) QK_FAIL(QK_ERR_INVALID, "qk_pack_topk: null argument");
    if (G <= 0 || k <= 0 || per < 0) QK_FAIL(QK_ERR_INVALID, "qk_pack_topk: bad sizes");
    QK_HIP(hipSetDevice(ctx->device));
    return qk_pack_topk_device(ctx, ids, key, G, per, k, packed);
}

int qk_merge_topk_packed(qk_ctx *ctx, const void *packed, int G, int64_t per, int k, int metric, int64_t *out_ids, float *out_dist) {
    if (!ctx || (per > 0 && (!packed || !out_ids))) QK_FAIL(QK_ERR_INVALID, "qk_merge_topk_packed: null argument");
    if (G <= 0 || k <= 0 || per < 0) QK_FAIL(QK_ERR_INVALID, "qk_merge_topk_packed: bad sizes");
    QK_TRY(check_metric(metric));
    QK_HIP(hipSetDevice(ctx->device));
    return qk_merge_topk_packed_device(ctx, packed, G, per, k, metric, out_ids, out_dist, true);
}

}  // extern "C"

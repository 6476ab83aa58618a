// qk_group.hip -- device group: ONE process, G devices, one index.
//
// The reference scales a search over cores with IndexBuildParams::num_workers: QueryCoordinator::initialize_workers
// (query_coordinator.cpp:50-74) -> PartitionManager::distribute_partitions (partition i -> core i % num_workers,
// partition_manager.cpp:557-603) -> worker_scan (per-core jobs, local buffers, batch_add into the global one,
// query_coordinator.cpp:243-469,98-240).  On MI355X a "worker" is a GPU:
//   - list p lives in the store of member p % G (its own HBM); list numbers stay global, a member's store simply does not
//     hold the others (the scan skips absent lists exactly like empty ones);
//   - the parent's centroids are REPLICATED on every member (33 MB at 65536 x 128) and re-replicated when the parent changes
//     (qk_store::version);
//   - a search: queries reach the lead member, the others pull them over xGMI; the coarse step is split by QUERIES (member j
//     ranks the centroids for its slice of the batch) and every member writes its slice of the [Q][nprobe] list straight into
//     every other member's copy (one kernel with peer stores, k_bcast); every member scans the whole batch over the lists IT
//     holds with squared keys, packs its [Q][k] (ids, keys) into one 12-byte-per-entry block and writes that block straight
//     into the lead's receive buffer (qk_pack_topk's layout, peer stores); the lead merges the G blocks under the (key, id)
//     order (qk_merge_topk_packed's kernel) and hands the answer out.  Ordering between devices is events only -- no host
//     thread per device, no host synchronisation inside a call on device buffers.
// Members may share a physical device (num_workers > GPUs, and the one-GPU test box): every step is then the same code with
// local instead of peer addresses.
#include "qk_internal.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int QK_GROUP_MAX = 64;
constexpr int64_t QK_GROUP_SPLIT_MIN_Q = 64;  // queries per member from which the coarse step is split by queries

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Member {
    qk_ctx *ctx = nullptr;
    qk_store *store = nullptr;
    qk_store *parent = nullptr;  // replica of the parent's lists (the centroids) on this member's device
    char *buf = nullptr;         // per-call buffers: queries | list numbers | local ids | local keys
    size_t cap = 0;
    hipEvent_t ev_coarse = nullptr, ev_done = nullptr;
};

// One submit thread per member (round 6).  A search enqueues ~45 us of host work per member (two pipeline enqueues, a peer
// broadcast, a pack, events); one host thread doing that for 8 members took 0.36-0.39 ms per call -- more than a member's device
// work at the 8-GPU configs[3] shape (0.22 ms), so the ONE host thread bounded the group.  The per-member pieces of a call are
// independent (own context, own stream, own buffers; only events cross), so they are handed to G - 1 persistent threads while the
// caller does member 0's: a call costs the host one member's share plus three fork-joins.  The workers spin for ~0.2 ms after a
// piece (the next piece of the same call, or the next call of a serving loop, arrives within that) and sleep on a condition
// variable otherwise.  Error text is thread-local in this library: a worker's message travels back with its status.
struct MemberPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> gen{0};
    std::atomic<int> pending{0}, sleepers{0};
    std::atomic<bool> stop{false};
    const std::function<int(int)> *fn = nullptr;  // published by the bump of `gen`
    int rc[64];
    std::string err[64];

    void worker(int j) {
        uint64_t seen = 0;
        for (;;) {
            int spins = 0;
            while (gen.load() == seen && !stop.load()) {
                if (++spins < 8000) {  // (~0.2 ms of pauses)
                    __builtin_ia32_pause();
                    continue;
                }
                std::unique_lock<std::mutex> lk(mu);
                sleepers.fetch_add(1);
                cv.wait(lk, [&] { return gen.load() != seen || stop.load(); });
                sleepers.fetch_sub(1);
                spins = 0;
            }
            if (stop.load()) return;
            seen = gen.load();
            rc[j] = (*fn)(j);
            if (rc[j] != QK_OK) err[j] = qk_last_error();
            pending.fetch_sub(1);
        }
    }
    void start(int G) {
        for (int j = 1; j < G; j++) th.emplace_back([this, j] { worker(j); });
    }
    void shutdown() {
        stop.store(true);
        {
            std::lock_guard<std::mutex> lk(mu);
            cv.notify_all();
        }
        for (auto &t : th) t.join();
        th.clear();
    }
    // f(j) for every member j in [0, G): j = 0 here, the others on their threads; returns the first failure in member order
    int run(int G, const std::function<int(int)> &f) {
        fn = &f;
        pending.store(G - 1);
        gen.fetch_add(1);
        if (sleepers.load() > 0) {
            std::lock_guard<std::mutex> lk(mu);
            cv.notify_all();
        }
        const int rc0 = f(0);
        while (pending.load() != 0) __builtin_ia32_pause();
        if (rc0 != QK_OK) return rc0;
        for (int j = 1; j < G; j++)
            if (rc[j] != QK_OK) {
                qk_set_error("%s", err[j].c_str());
                return rc[j];
            }
        return QK_OK;
    }
};

}  // namespace

struct qk_group {
    int G = 0, d = 0;
    std::vector<Member> m;
    MemberPool *pool = nullptr;  // submit threads (G >= 2); nullptr or submit_threads == false: the caller's thread does every member
    bool submit_threads = true;  // qk_group_set_submit_threads
    uint64_t parent_uid = 0, parent_version = 0;  // what the replicas were made from
    bool parent_valid = false;
    char *recv = nullptr;  // on the lead: G packed blocks (+ the staging of a host answer)
    size_t recv_cap = 0;
    hipEvent_t ev_x = nullptr, ev_all = nullptr;  // lead: queries are there / every slice of the list numbers is everywhere
    hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};  // lead: phases of a timed call
};

namespace {

struct BcastArgs {
    unsigned long long *dst[QK_GROUP_MAX];
    int n;
};

// one member's slice of the [Q][nprobe] list numbers -> the same place in every other member's copy (peer stores)
__global__ __launch_bounds__(256) void k_bcast(const unsigned long long *__restrict__ src, BcastArgs a, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const unsigned long long v = src[i];
    for (int t = 0; t < a.n; t++) a.dst[t][i] = v;
}

__global__ __launch_bounds__(256) void k_gather_rows(const float *__restrict__ x, const int64_t *__restrict__ sel, int64_t n, int d,
                                                     float *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * d) return;
    const int64_t i = idx / d;
    out[idx] = x[sel[i] * d + (idx - i * d)];
}

__global__ __launch_bounds__(256) void k_gather_i64(const int64_t *__restrict__ src, const int64_t *__restrict__ sel, int64_t n,
                                                    int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = src[sel[i]];
}

// adaptive search over the members: pair (q, i) of a round was scanned by the member that holds its list -- its k entries come from
// that member's per-pair result (peer reads), the pairs without a list get the padding
struct GatherArgs {
    const int64_t *ids[QK_GROUP_MAX];
    const float *key[QK_GROUP_MAX];
    int G;
};
__global__ __launch_bounds__(256) void k_aps_gather(const int64_t *__restrict__ round_pids, int64_t npairs, int k, int metric, GatherArgs a,
                                                    int64_t *__restrict__ out_ids, float *__restrict__ out_key) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs * k) return;
    const int64_t pair = i / k;
    const int64_t p = round_pids[pair];
    if (p < 0) {
        out_ids[i] = -1;
        out_key[i] = metric == QK_METRIC_L2 ? __builtin_inff() : -__builtin_inff();
        return;
    }
    const int o = (int)(p % a.G);
    out_ids[i] = a.ids[o][i];
    out_key[i] = a.key[o][i];
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256); }

int sync_all(qk_group *g) {
    for (auto &mb : g->m) {
        QK_HIP(hipSetDevice(mb.ctx->device));
        QK_HIP(hipStreamSynchronize(mb.ctx->stream));
    }
    return QK_OK;
}

// buffers other members write into: nobody may still be using the old ones when they are replaced
int reserve_call_buffers(qk_group *g, size_t per_member, size_t lead_recv) {
    bool grow = lead_recv > g->recv_cap;
    for (auto &mb : g->m) grow = grow || per_member > mb.cap;
    if (!grow) return QK_OK;
    QK_TRY(sync_all(g));
    for (auto &mb : g->m) {
        if (per_member <= mb.cap) continue;
        QK_HIP(hipSetDevice(mb.ctx->device));
        if (mb.buf) QK_HIP(hipFree(mb.buf));
        mb.buf = nullptr;
        mb.cap = 0;
        const size_t want = per_member + per_member / 4 + 4096;
        if (hipMalloc((void **)&mb.buf, want) != hipSuccess) {
            (void)hipGetLastError();
            QK_FAIL(QK_ERR_OOM, "qk_group: %zu bytes of call buffers could not be allocated on device %d", want, mb.ctx->device);
        }
        mb.cap = want;
    }
    if (lead_recv > g->recv_cap) {
        Member &lead = g->m[0];
        QK_HIP(hipSetDevice(lead.ctx->device));
        if (g->recv) QK_HIP(hipFree(g->recv));
        g->recv = nullptr;
        g->recv_cap = 0;
        const size_t want = lead_recv + lead_recv / 4 + 4096;
        if (hipMalloc((void **)&g->recv, want) != hipSuccess) {
            (void)hipGetLastError();
            QK_FAIL(QK_ERR_OOM, "qk_group: %zu bytes of receive buffer could not be allocated", want);
        }
        g->recv_cap = want;
    }
    return QK_OK;
}

// Members read buffers on `device` directly (the parent's lists, a caller's device tensors): fine when it is a member's device --
// qk_group_create enabled peer access among those --, else peer access is enabled on demand, or the call fails cleanly instead of
// faulting on the device.
int reachable(qk_group *g, int device, const char *what) {
    for (auto &mb : g->m)
        if (mb.ctx->device == device) return QK_OK;
    for (auto &mb : g->m) {
        int can = 0;
        QK_HIP(hipDeviceCanAccessPeer(&can, mb.ctx->device, device));
        if (!can) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_group: %s lives on device %d, which member device %d cannot access", what, device, mb.ctx->device);
        QK_HIP(hipSetDevice(mb.ctx->device));
        const hipError_t e = hipDeviceEnablePeerAccess(device, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        else if (e != hipSuccess) QK_FAIL(QK_ERR_HIP, "qk_group: hipDeviceEnablePeerAccess(%d -> %d): %s", mb.ctx->device, device, hipGetErrorString(e));
    }
    return QK_OK;
}

// the parent's lists (the centroids, ids = list numbers) on every member's device; redone when the parent has changed
int sync_parent(qk_group *g, qk_store *parent) {
    if (parent->d != g->d) QK_FAIL(QK_ERR_INVALID, "parent store dimension %d != group dimension %d", parent->d, g->d);
    qk_ctx *pc = parent->ctx;
    QK_TRY(reachable(g, pc->device, "the parent store"));
    QK_HIP(hipSetDevice(pc->device));
    QK_TRY(qk_store_sync_table(parent));  // (bumps parent->version when the parent was touched since)
    if (g->parent_valid && g->parent_uid == parent->uid && g->parent_version == parent->version) return QK_OK;
    g->parent_valid = false;
    QK_TRY(sync_all(g));  // searches in flight read the old replicas
    int64_t mx = 0;
    for (const qk_part &pt : parent->parts)
        if (pt.present) mx = std::max(mx, pt.size);
    float *tv = nullptr;
    int64_t *ti = nullptr;
    QK_HIP(hipSetDevice(pc->device));
    if (mx > 0) {
        QK_HIP(hipMalloc((void **)&tv, (size_t)mx * parent->d * 4));
        if (hipMalloc((void **)&ti, (size_t)mx * 8) != hipSuccess) {
            (void)hipGetLastError();
            hipFree(tv);
            QK_FAIL(QK_ERR_OOM, "qk_group: no room to replicate the parent");
        }
    }
    int rc = QK_OK;
    for (auto &mb : g->m) {
        if (!mb.parent) rc = qk_store_create(mb.ctx, g->d, &mb.parent);
        if (rc == QK_OK) rc = qk_store_reset(mb.parent);
        if (rc != QK_OK) break;
    }
    for (size_t p = 0; rc == QK_OK && p < parent->parts.size(); p++) {
        const qk_part &pt = parent->parts[p];
        if (!pt.present) continue;
        if (pt.size > 0) {
            rc = qk_store_get_list(parent, (int64_t)p, tv, ti, QK_MEM_DEVICE);
            if (rc == QK_OK && (hipSetDevice(pc->device) != hipSuccess || hipStreamSynchronize(pc->stream) != hipSuccess)) {
                qk_set_error("qk_group: reading the parent failed");
                rc = QK_ERR_HIP;
            }
        }
        for (auto &mb : g->m) {
            if (rc != QK_OK) break;
            rc = qk_store_add_list(mb.parent, (int64_t)p);
            if (rc == QK_OK && pt.size > 0) rc = qk_store_add_entries(mb.parent, (int64_t)p, pt.size, ti, tv, QK_MEM_DEVICE);
        }
    }
    hipSetDevice(pc->device);
    if (tv) hipFree(tv);
    if (ti) hipFree(ti);
    QK_TRY(rc);
    g->parent_uid = parent->uid;
    g->parent_version = parent->version;
    g->parent_valid = true;
    return QK_OK;
}

int check_metric(int metric) {
    if (metric != QK_METRIC_L2 && metric != QK_METRIC_IP) QK_FAIL(QK_ERR_INVALID, "Metric type not supported");
    return QK_OK;
}

// the per-member pieces of a call: on the submit threads when the group has them, else one after the other on the caller's thread
int for_members(qk_group *g, const std::function<int(int)> &f) {
    if (g->pool && g->submit_threads && g->G > 1) return g->pool->run(g->G, f);
    for (int j = 0; j < g->G; j++) QK_TRY(f(j));
    return QK_OK;
}

// parent != nullptr: QueryCoordinator::search at fixed nprobe; else scan_partitions over pids [Q][P] (P == 0: padding only)
int group_search(qk_group *g, qk_store *parent, const float *x, int64_t Q, const int64_t *pids_in, int P_in, int nprobe, int k,
                 int metric, int64_t *out_ids, float *out_dist, int mem, qk_timing *timing) {
    if (timing) memset(timing, 0, sizeof(*timing));
    if (Q <= 0) return QK_OK;
    const int G = g->G, d = g->d;
    Member &lead = g->m[0];
    for (auto &mb : g->m) QK_TRY(qk_check_overflow(mb.ctx));
    const bool use_parent = parent != nullptr;
    int kk = 0;
    if (use_parent) {
        QK_TRY(sync_parent(g, parent));
        kk = (int)std::min<int64_t>(nprobe, parent->ntotal);  // query_coordinator.cpp:641
        if (kk > QK_MAX_NPROBE) QK_FAIL(QK_ERR_UNSUPPORTED, "nprobe=%d exceeds QK_MAX_NPROBE=%d", kk, QK_MAX_NPROBE);
    }
    const bool no_lists = !use_parent && P_in == 0;  // zero partitions: padded output (query_coordinator.cpp:459-497)
    const int P = use_parent ? std::max(kk, 1) : (no_lists ? 1 : P_in);
    const bool split = use_parent && kk > 0 && G > 1 && Q >= QK_GROUP_SPLIT_MIN_Q * G;
    const size_t bx = al256((size_t)Q * d * 4), bp = al256((size_t)Q * P * 8), bi = al256((size_t)Q * k * 8),
                 bd = al256((size_t)Q * k * 4);
    const size_t blk = qk_topk_block_bytes_(Q, k);
    QK_TRY(reserve_call_buffers(g, bx + bp + bi + bd, al256((size_t)G * blk) + bi + bd));
    auto xb = [&](Member &mb) { return (float *)mb.buf; };
    auto pb = [&](Member &mb) { return (int64_t *)(mb.buf + bx); };
    auto ib = [&](Member &mb) { return (int64_t *)(mb.buf + bx + bp); };
    auto kb = [&](Member &mb) { return (float *)(mb.buf + bx + bp + bi); };
    const bool tm = timing != nullptr;

    // ---- the batch reaches the lead, the others pull it from there ---------------------------------------------------------
    QK_HIP(hipSetDevice(lead.ctx->device));
    hipStream_t ls = lead.ctx->stream;
    if (tm) QK_HIP(hipEventRecord(g->tev[0], ls));
    QK_HIP(hipMemcpyAsync(xb(lead), x, (size_t)Q * d * 4, hipMemcpyDefault, ls));
    if (!use_parent) {
        if (no_lists) QK_HIP(hipMemsetAsync(pb(lead), 0xFF, (size_t)Q * 8, ls));
        else QK_HIP(hipMemcpyAsync(pb(lead), pids_in, (size_t)Q * P * 8, hipMemcpyDefault, ls));
    }
    QK_HIP(hipEventRecord(g->ev_x, ls));
    // ---- per member (submit threads when the group has them): pull the batch; coarse step split by queries -- member j ranks the
    //      centroids for its slice and writes the slice into every other member's copy ------------------------------------------
    const int64_t per_q = (Q + G - 1) / G;
    auto pull_and_rank = [&](int j) -> int {
        Member &mb = g->m[j];
        QK_HIP(hipSetDevice(mb.ctx->device));
        hipStream_t st = mb.ctx->stream;
        if (j > 0) {
            QK_HIP(hipStreamWaitEvent(st, g->ev_x, 0));
            QK_HIP(hipMemcpyAsync(xb(mb), xb(lead), (size_t)Q * d * 4, hipMemcpyDefault, st));
            if (!use_parent) QK_HIP(hipMemcpyAsync(pb(mb), pb(lead), (size_t)Q * P * 8, hipMemcpyDefault, st));
        }
        if (!split) return QK_OK;
        const int64_t q0 = (int64_t)j * per_q, qn = std::min(per_q, Q - q0);
        if (qn > 0) {
            QK_TRY(qk_run_search(mb.ctx, mb.parent, mb.parent, xb(mb) + q0 * d, qn, nullptr, 0, nprobe, 0, metric,
                                 pb(mb) + q0 * kk, nullptr, QK_MEM_DEVICE, nullptr, true, true));
            BcastArgs ba;
            ba.n = 0;
            for (int t = 0; t < G; t++)
                if (t != j) ba.dst[ba.n++] = (unsigned long long *)(pb(g->m[t]) + q0 * kk);
            const int64_t n8 = qn * kk;
            hipLaunchKernelGGL(k_bcast, dim3(grid_for(n8)), dim3(256), 0, st, (const unsigned long long *)(pb(mb) + q0 * kk), ba, n8);
            QK_HIP(hipGetLastError());
        }
        if (j > 0) QK_HIP(hipEventRecord(mb.ev_coarse, st));
        return QK_OK;
    };
    QK_TRY(for_members(g, pull_and_rank));
    if (split) {
        // two-hop event sync: members -> lead -> members (2 (G - 1) waits instead of G^2); the members' half is the first thing
        // their scan piece does (an event must have been recorded on the host before a wait on it is enqueued)
        QK_HIP(hipSetDevice(lead.ctx->device));
        for (int j = 1; j < G; j++) QK_HIP(hipStreamWaitEvent(ls, g->m[j].ev_coarse, 0));
        if (tm) QK_HIP(hipEventRecord(g->tev[1], ls));
        QK_HIP(hipEventRecord(g->ev_all, ls));
    } else if (tm) {
        QK_HIP(hipEventRecord(g->tev[1], ls));
    }
    // ---- every member: the whole batch over the lists it holds; its block goes straight into the lead's receive buffer ------
    std::vector<qk_timing> mt(tm ? (size_t)G : 0);
    auto scan_and_pack = [&](int j) -> int {
        Member &mb = g->m[j];
        QK_HIP(hipSetDevice(mb.ctx->device));
        if (split && j > 0) QK_HIP(hipStreamWaitEvent(mb.ctx->stream, g->ev_all, 0));
        qk_timing *tj = tm ? &mt[(size_t)j] : nullptr;
        if (split || !use_parent)
            QK_TRY(qk_run_search(mb.ctx, nullptr, mb.store, xb(mb), Q, pb(mb), split ? kk : P, 0, k, metric, ib(mb), kb(mb),
                                 QK_MEM_DEVICE, tj, false, true));
        else  // few queries (or an empty parent): every member ranks the centroids itself -- same list on every member
            QK_TRY(qk_run_search(mb.ctx, mb.parent, mb.store, xb(mb), Q, nullptr, 0, nprobe, k, metric, ib(mb), kb(mb),
                                 QK_MEM_DEVICE, tj, false, true));
        QK_TRY(qk_pack_topk_device(mb.ctx, ib(mb), kb(mb), 1, Q, k, g->recv + (size_t)j * blk));
        if (j > 0) QK_HIP(hipEventRecord(mb.ev_done, mb.ctx->stream));
        return QK_OK;
    };
    QK_TRY(for_members(g, scan_and_pack));
    // ---- lead: merge of the G blocks (the cross-worker batch_add, query_coordinator.cpp:167-173,231-235) --------------------
    QK_HIP(hipSetDevice(lead.ctx->device));
    for (int j = 1; j < G; j++) QK_HIP(hipStreamWaitEvent(ls, g->m[j].ev_done, 0));
    if (tm) QK_HIP(hipEventRecord(g->tev[2], ls));
    int64_t *o_ids = out_ids;
    float *o_dist = out_dist;
    if (mem == QK_MEM_HOST) {
        o_ids = (int64_t *)(g->recv + al256((size_t)G * blk));
        o_dist = out_dist ? (float *)(g->recv + al256((size_t)G * blk) + bi) : nullptr;
    }
    QK_TRY(qk_merge_topk_packed_device(lead.ctx, g->recv, G, Q, k, metric, o_ids, o_dist, true));
    if (tm) QK_HIP(hipEventRecord(g->tev[3], ls));
    if (mem == QK_MEM_HOST) {
        QK_HIP(hipMemcpyAsync(out_ids, o_ids, (size_t)Q * k * 8, hipMemcpyDeviceToHost, ls));
        if (out_dist) QK_HIP(hipMemcpyAsync(out_dist, o_dist, (size_t)Q * k * 4, hipMemcpyDeviceToHost, ls));
        QK_HIP(hipStreamSynchronize(ls));
    }
    if (tm) {
        QK_HIP(hipStreamSynchronize(ls));
        float ms = 0.f;
        QK_HIP(hipEventElapsedTime(&ms, g->tev[0], g->tev[1]));
        timing->coarse_ms = ms;
        QK_HIP(hipEventElapsedTime(&ms, g->tev[1], g->tev[2]));
        timing->scan_ms = ms;
        QK_HIP(hipEventElapsedTime(&ms, g->tev[2], g->tev[3]));
        timing->merge_ms = ms;
        QK_HIP(hipEventElapsedTime(&ms, g->tev[0], g->tev[3]));
        timing->total_ms = ms;
        for (int j = 0; j < G; j++) {
            Member &mb = g->m[j];
            qk_timing &t = mt[(size_t)j];
            if (t.n_items < 0) {  // the one-launch small-batch search leaves no scalars
                t.n_items = 0;
            } else {
                QK_HIP(hipSetDevice(mb.ctx->device));
                QK_TRY(qk_finish_timing(mb.ctx, mb.store, &t, false, 4));
            }
            timing->n_items += t.n_items;
            timing->scan_bytes += t.scan_bytes;
            timing->partitions_scanned += t.partitions_scanned;
        }
    }
    if (tm || mem == QK_MEM_HOST)
        for (auto &mb : g->m) QK_TRY(qk_check_overflow(mb.ctx));
    return QK_OK;
}

Member *owner(qk_group *g, int64_t list_no) { return &g->m[(size_t)(list_no % g->G)]; }

}  // namespace

extern "C" {

int qk_group_create(const int *devices, int G, int d, qk_group **out) {
    if (!devices || !out) QK_FAIL(QK_ERR_INVALID, "qk_group_create: null argument");
    if (G < 1 || G > QK_GROUP_MAX) QK_FAIL(QK_ERR_INVALID, "qk_group_create: the number of members must be in [1, %d] (got %d)", QK_GROUP_MAX, G);
    if (d <= 0) QK_FAIL(QK_ERR_INVALID, "qk_group_create: d must be positive (got %d)", d);
    int ndev = 0;
    QK_HIP(hipGetDeviceCount(&ndev));
    for (int j = 0; j < G; j++)
        if (devices[j] < 0 || devices[j] >= ndev)
            QK_FAIL(QK_ERR_INVALID, "qk_group_create: device %d out of range (%d devices)", devices[j], ndev);
    // every member reads and writes every other member's buffers directly: peer access between all distinct devices
    for (int a = 0; a < G; a++)
        for (int b = 0; b < G; b++) {
            if (devices[a] == devices[b]) continue;
            bool seen = false;  // (each ordered pair once)
            for (int a2 = 0; a2 <= a && !seen; a2++)
                for (int b2 = 0; b2 < (a2 == a ? b : G) && !seen; b2++)
                    seen = devices[a2] == devices[a] && devices[b2] == devices[b];
            if (seen) continue;
            int can = 0;
            QK_HIP(hipDeviceCanAccessPeer(&can, devices[a], devices[b]));
            if (!can) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_group_create: device %d cannot access device %d (no peer access)", devices[a], devices[b]);
            QK_HIP(hipSetDevice(devices[a]));
            hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else if (e != hipSuccess) QK_FAIL(QK_ERR_HIP, "qk_group_create: hipDeviceEnablePeerAccess(%d -> %d): %s", devices[a], devices[b], hipGetErrorString(e));
        }
    qk_group *g = new qk_group();
    g->G = G;
    g->d = d;
    g->m.resize((size_t)G);
    int rc = QK_OK;
    for (int j = 0; j < G && rc == QK_OK; j++) {
        Member &mb = g->m[(size_t)j];
        rc = qk_ctx_create(devices[j], &mb.ctx);
        if (rc == QK_OK) rc = qk_store_create(mb.ctx, d, &mb.store);
        if (rc == QK_OK) {
            mb.ctx->squared_l2 = true;  // members hand merge keys to the lead; sqrt happens after the merge
            if (hipEventCreateWithFlags(&mb.ev_coarse, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&mb.ev_done, hipEventDisableTiming) != hipSuccess) {
                qk_set_error("qk_group_create: event creation failed");
                rc = QK_ERR_HIP;
            }
        }
    }
    if (rc == QK_OK) {
        hipSetDevice(devices[0]);
        if (hipEventCreateWithFlags(&g->ev_x, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_all, hipEventDisableTiming) != hipSuccess) rc = QK_ERR_HIP;
        for (auto &e : g->tev)
            if (rc == QK_OK && hipEventCreate(&e) != hipSuccess) rc = QK_ERR_HIP;
        if (rc != QK_OK) qk_set_error("qk_group_create: event creation failed");
    }
    if (rc != QK_OK) {
        qk_group_destroy(g);
        return rc;
    }
    if (G > 1) {
        g->pool = new MemberPool();
        g->pool->start(G);
    }
    *out = g;
    return QK_OK;
}

int qk_group_set_submit_threads(qk_group *g, int enabled) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_set_submit_threads: group is null");
    g->submit_threads = enabled != 0;
    return QK_OK;
}

int qk_group_destroy(qk_group *g) {
    if (!g) return QK_OK;
    if (g->pool) {
        g->pool->shutdown();
        delete g->pool;
        g->pool = nullptr;
    }
    for (auto &mb : g->m)
        if (mb.ctx) {
            hipSetDevice(mb.ctx->device);
            hipStreamSynchronize(mb.ctx->stream);
        }
    if (!g->m.empty() && g->m[0].ctx) hipSetDevice(g->m[0].ctx->device);
    if (g->recv) hipFree(g->recv);
    if (g->ev_x) hipEventDestroy(g->ev_x);
    if (g->ev_all) hipEventDestroy(g->ev_all);
    for (auto e : g->tev)
        if (e) hipEventDestroy(e);
    for (auto &mb : g->m) {
        if (!mb.ctx) continue;
        hipSetDevice(mb.ctx->device);
        if (mb.buf) hipFree(mb.buf);
        if (mb.ev_coarse) hipEventDestroy(mb.ev_coarse);
        if (mb.ev_done) hipEventDestroy(mb.ev_done);
        if (mb.parent) qk_store_destroy(mb.parent);
        if (mb.store) qk_store_destroy(mb.store);
        qk_ctx_destroy(mb.ctx);
    }
    delete g;
    return QK_OK;
}

int qk_group_size(qk_group *g) { return g ? g->G : 0; }

int qk_group_member(qk_group *g, int i, qk_ctx **ctx, qk_store **store) {
    if (!g || i < 0 || i >= g->G) QK_FAIL(QK_ERR_INVALID, "qk_group_member: bad arguments");
    if (ctx) *ctx = g->m[(size_t)i].ctx;
    if (store) *store = g->m[(size_t)i].store;
    return QK_OK;
}

int qk_group_owner(qk_group *g, int64_t list_no) { return (!g || list_no < 0) ? -1 : (int)(list_no % g->G); }

int qk_group_set_stream(qk_group *g, void *hip_stream) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_set_stream: group is null");
    return qk_ctx_set_stream(g->m[0].ctx, hip_stream);
}

int qk_group_set_null_stream(qk_group *g) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_set_null_stream: group is null");
    return qk_ctx_set_null_stream(g->m[0].ctx);
}

int qk_group_get_stream(qk_group *g, void **hip_stream, int *kind) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_get_stream: group is null");
    return qk_ctx_get_stream(g->m[0].ctx, hip_stream, kind);
}

int qk_group_synchronize(qk_group *g) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_synchronize: group is null");
    QK_TRY(sync_all(g));
    for (auto &mb : g->m) QK_TRY(qk_check_overflow(mb.ctx));
    return QK_OK;
}

int qk_group_set_form_feedback(qk_group *g, int enabled) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_set_form_feedback: group is null");
    for (auto &mb : g->m) mb.ctx->form_feedback = enabled != 0;
    return QK_OK;
}

// ---- the store surface, routed to the member that holds the list --------------------------------------------------------------
int qk_group_reset(qk_group *g) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_reset: group is null");
    QK_TRY(sync_all(g));
    for (auto &mb : g->m) QK_TRY(qk_store_reset(mb.store));
    return QK_OK;
}

int qk_group_add_list(qk_group *g, int64_t list_no) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_add_list: group is null");
    if (list_no < 0) QK_FAIL(QK_ERR_INVALID, "qk_store_add_list: negative list number");
    return qk_store_add_list(owner(g, list_no)->store, list_no);
}

int qk_group_remove_list(qk_group *g, int64_t list_no) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_remove_list: group is null");
    if (list_no < 0) return QK_OK;
    return qk_store_remove_list(owner(g, list_no)->store, list_no);
}

int qk_group_add_entries(qk_group *g, int64_t list_no, int64_t n, const int64_t *ids, const float *vecs, int mem) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_add_entries: group is null");
    if (list_no < 0) QK_FAIL(QK_ERR_NOT_FOUND, "List does not exist in add_entries (list %lld)", (long long)list_no);
    return qk_store_add_entries(owner(g, list_no)->store, list_no, n, ids, vecs, mem);
}

int qk_group_list_size(qk_group *g, int64_t list_no, int64_t *out) {
    if (!g || !out) QK_FAIL(QK_ERR_INVALID, "qk_group_list_size: null argument");
    if (list_no < 0) QK_FAIL(QK_ERR_NOT_FOUND, "List does not exist in list_size (list %lld)", (long long)list_no);
    return qk_store_list_size(owner(g, list_no)->store, list_no, out);
}

int qk_group_list_sizes(qk_group *g, const int64_t *list_nos, int64_t n, int64_t *out) {
    if (!g || (n > 0 && (!list_nos || !out))) QK_FAIL(QK_ERR_INVALID, "qk_group_list_sizes: null argument");
    for (int64_t i = 0; i < n; i++) QK_TRY(qk_group_list_size(g, list_nos[i], out + i));
    return QK_OK;
}

int64_t qk_group_ntotal(qk_group *g) {
    int64_t n = 0;
    if (g)
        for (auto &mb : g->m) n += mb.store->ntotal;
    return n;
}

int64_t qk_group_nlist(qk_group *g) {
    int64_t n = 0;
    if (g)
        for (auto &mb : g->m) n += mb.store->nlist;
    return n;
}

int qk_group_d(qk_group *g) { return g ? g->d : 0; }

int64_t qk_group_device_bytes(qk_group *g) {
    int64_t n = 0;
    if (g)
        for (auto &mb : g->m) n += qk_store_device_bytes(mb.store);
    return n;
}

int qk_group_list_ids(qk_group *g, int64_t *out_host, int64_t *n) {
    if (!g || !n) QK_FAIL(QK_ERR_INVALID, "qk_group_list_ids: null argument");
    int64_t cnt = 0;
    size_t mx = 0;
    for (auto &mb : g->m) mx = std::max(mx, mb.store->parts.size());
    for (size_t p = 0; p < mx; p++) {  // ascending: list p can only be present in member p % G
        const qk_store *s = g->m[p % (size_t)g->G].store;
        if (p < s->parts.size() && s->parts[p].present) {
            if (out_host) out_host[cnt] = (int64_t)p;
            cnt++;
        }
    }
    *n = cnt;
    return QK_OK;
}

int qk_group_get_list(qk_group *g, int64_t list_no, float *vecs_out, int64_t *ids_out, int mem) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_get_list: group is null");
    if (list_no < 0) QK_FAIL(QK_ERR_NOT_FOUND, "List does not exist in get_codes (list %lld)", (long long)list_no);
    Member *o = owner(g, list_no);
    if (mem == QK_MEM_DEVICE && (vecs_out || ids_out)) {  // the owner writes the caller's device buffer: it must be able to reach it
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, vecs_out ? (const void *)vecs_out : (const void *)ids_out) == hipSuccess)
            QK_TRY(reachable(g, at.device, "the destination buffer"));
        else
            (void)hipGetLastError();
    }
    QK_TRY(qk_store_get_list(o->store, list_no, vecs_out, ids_out, mem));
    if (mem == QK_MEM_DEVICE) {  // the caller's stream is not the owner's: complete on return
        QK_HIP(hipSetDevice(o->ctx->device));
        QK_HIP(hipStreamSynchronize(o->ctx->stream));
    }
    return QK_OK;
}

int qk_group_get_lists(qk_group *g, const int64_t *list_nos, int64_t n, float *vecs_out, int64_t *ids_out, int mem) {
    if (!g || (n > 0 && !list_nos)) QK_FAIL(QK_ERR_INVALID, "qk_group_get_lists: null argument");
    int64_t at = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t sz = 0;
        QK_TRY(qk_group_list_size(g, list_nos[i], &sz));
        QK_TRY(qk_group_get_list(g, list_nos[i], vecs_out ? vecs_out + at * g->d : nullptr, ids_out ? ids_out + at : nullptr, mem));
        at += sz;
    }
    return QK_OK;
}

int qk_group_get_vector(qk_group *g, int64_t id, float *vec_out_host, int *found) {
    if (!g || !vec_out_host || !found) QK_FAIL(QK_ERR_INVALID, "qk_group_get_vector: null argument");
    *found = 0;
    // first match in ascending list order (dynamic_inverted_list.cpp:280-293): the lowest holding list over the members
    int64_t best = -1;
    Member *bm = nullptr;
    for (auto &mb : g->m) {
        qk_store_ensure_index(mb.store);
        const int32_t holder = mb.store->id_to_list.find(id);
        if (holder >= 0 && (best < 0 || holder < best)) {
            best = holder;
            bm = &mb;
        }
    }
    if (!bm) return QK_OK;
    return qk_store_get_vector(bm->store, id, vec_out_host, found);
}

int qk_group_remove_ids(qk_group *g, int64_t n, const int64_t *ids_host, int64_t *n_removed) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_remove_ids: group is null");
    if (n_removed) *n_removed = 0;
    if (n <= 0) return QK_OK;
    if (!ids_host) QK_FAIL(QK_ERR_INVALID, "qk_group_remove_ids: null ids");
    int64_t total = 0;
    for (auto &mb : g->m) {
        int64_t r = 0;
        QK_TRY(qk_store_remove_ids(mb.store, n, ids_host, &r));
        total += r;
    }
    if (n_removed) *n_removed = total;
    return QK_OK;
}

// PartitionManager::add's loop (partition_manager.cpp:236-258) over the members: vector i goes to list assign[i], i.e. to member
// assign[i] % G; inside a list the append order is the input order (every member sees its rows in input order).
int qk_group_add_batch(qk_group *g, int64_t n, const int64_t *ids, const float *vecs, const int64_t *assign, int mem) {
    if (!g) QK_FAIL(QK_ERR_INVALID, "qk_group_add_batch: group is null");
    if (n == 0) return QK_OK;
    if (n < 0 || !ids || !vecs || !assign) QK_FAIL(QK_ERR_INVALID, "qk_group_add_batch: bad arguments");
    const int G = g->G, d = g->d;
    if (G == 1) return qk_store_add_batch(g->m[0].store, n, ids, vecs, assign, mem);
    std::vector<int64_t> h_assign((size_t)n);
    if (mem == QK_MEM_HOST) {
        memcpy(h_assign.data(), assign, (size_t)n * 8);
    } else {
        QK_HIP(hipMemcpy(h_assign.data(), assign, (size_t)n * 8, hipMemcpyDefault));
    }
    std::vector<std::vector<int64_t>> sel((size_t)G);
    for (int64_t i = 0; i < n; i++) {
        const int64_t p = h_assign[(size_t)i];
        Member *o = p >= 0 ? owner(g, p) : nullptr;
        if (!o || p >= (int64_t)o->store->parts.size() || !o->store->parts[(size_t)p].present)
            QK_FAIL(QK_ERR_NOT_FOUND, "List does not exist in add_entries (list %lld)", (long long)p);  // before anything is stored
        sel[(size_t)(p % G)].push_back(i);
    }
    for (int j = 0; j < G; j++) {
        const std::vector<int64_t> &sj = sel[(size_t)j];
        const int64_t nj = (int64_t)sj.size();
        if (nj == 0) continue;
        Member &mb = g->m[(size_t)j];
        if (mem == QK_MEM_HOST) {
            std::vector<float> v((size_t)nj * d);
            std::vector<int64_t> id((size_t)nj), as((size_t)nj);
            for (int64_t t = 0; t < nj; t++) {
                memcpy(v.data() + (size_t)t * d, vecs + (size_t)sj[(size_t)t] * d, (size_t)d * 4);
                id[(size_t)t] = ids[sj[(size_t)t]];
                as[(size_t)t] = h_assign[(size_t)sj[(size_t)t]];
            }
            QK_TRY(qk_store_add_batch(mb.store, nj, id.data(), v.data(), as.data(), QK_MEM_HOST));
        } else {  // the member gathers its rows from the caller's (possibly peer) memory, then appends them locally
            QK_HIP(hipSetDevice(mb.ctx->device));
            hipStream_t st = mb.ctx->stream;
            char *tmp = nullptr;
            const size_t bs = al256((size_t)nj * 8), bv = al256((size_t)nj * d * 4);
            QK_HIP(hipMalloc((void **)&tmp, 3 * bs + bv));
            int64_t *dsel = (int64_t *)tmp, *did = (int64_t *)(tmp + bs), *das = (int64_t *)(tmp + 2 * bs);
            float *dv = (float *)(tmp + 3 * bs);
            int rc = QK_OK;
            if (hipMemcpyAsync(dsel, sj.data(), (size_t)nj * 8, hipMemcpyHostToDevice, st) != hipSuccess) rc = QK_ERR_HIP;
            if (rc == QK_OK) {
                hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(nj * d)), dim3(256), 0, st, vecs, dsel, nj, d, dv);
                hipLaunchKernelGGL(k_gather_i64, dim3(grid_for(nj)), dim3(256), 0, st, ids, dsel, nj, did);
                hipLaunchKernelGGL(k_gather_i64, dim3(grid_for(nj)), dim3(256), 0, st, assign, dsel, nj, das);
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = QK_ERR_HIP;
            }
            if (rc == QK_OK) rc = qk_store_add_batch(mb.store, nj, did, dv, das, QK_MEM_DEVICE);
            else qk_set_error("qk_group_add_batch: gathering the rows of member %d failed", j);
            hipSetDevice(mb.ctx->device);
            hipFree(tmp);
            QK_TRY(rc);
        }
    }
    return QK_OK;
}

// init_partitions in bulk (qk_store_build_csr) over the members: list p -> member p % G.  Host data is uploaded ONCE, in chunks,
// to the lead; every member ingests its lists' rows from there (peer reads) -- or straight from the caller's device arrays.
int qk_group_build_csr(qk_group *g, int64_t nlist, const int64_t *offsets, const int64_t *ids, const float *vecs, int mem) {
    if (!g || !offsets || nlist < 0) QK_FAIL(QK_ERR_INVALID, "qk_group_build_csr: bad arguments");
    const int G = g->G, d = g->d;
    const int64_t total = offsets[nlist];
    if (total > 0 && (!ids || !vecs)) QK_FAIL(QK_ERR_INVALID, "qk_group_build_csr: null data");
    QK_TRY(sync_all(g));
    std::vector<int64_t> host_ids;
    const int64_t *hid = ids;
    if (mem == QK_MEM_DEVICE && total > 0) {
        host_ids.resize((size_t)total);
        QK_HIP(hipSetDevice(g->m[0].ctx->device));
        QK_HIP(hipDeviceSynchronize());  // whatever stream produced the caller's arrays
        QK_HIP(hipMemcpy(host_ids.data(), ids, (size_t)total * 8, hipMemcpyDefault));
        hid = host_ids.data();
    }
    std::vector<qk_csr_build> b((size_t)G);
    int rc = QK_OK;
    int begun = 0;
    for (int j = 0; j < G && rc == QK_OK; j++) {
        rc = qk_store_csr_begin(g->m[(size_t)j].store, nlist, offsets, hid, G, j, &b[(size_t)j]);
        begun = j + 1;
    }
    if (rc == QK_OK && total > 0) {
        Member &lead = g->m[0];
        const int64_t CH = mem == QK_MEM_HOST ? std::max<int64_t>(1, (int64_t)(128u << 20) / ((int64_t)d * 4 + 8)) : total;
        if (mem == QK_MEM_HOST) {
            hipSetDevice(lead.ctx->device);
            rc = qk_stage_reserve(lead.ctx, (size_t)CH * ((size_t)d * 4 + 8) + 512);
        }
        for (int64_t i0 = 0; rc == QK_OK && i0 < total; i0 += CH) {
            const int64_t n = std::min(CH, total - i0);
            const float *dv = vecs + i0 * d;
            const int64_t *di = ids + i0;
            if (mem == QK_MEM_HOST) {
                const size_t vb = (size_t)n * d * 4;
                char *si = lead.ctx->stage + al256(vb);
                hipSetDevice(lead.ctx->device);
                if (hipMemcpyAsync(lead.ctx->stage, dv, vb, hipMemcpyHostToDevice, lead.ctx->stream) != hipSuccess ||
                    hipMemcpyAsync(si, di, (size_t)n * 8, hipMemcpyHostToDevice, lead.ctx->stream) != hipSuccess ||
                    hipStreamSynchronize(lead.ctx->stream) != hipSuccess) {
                    qk_set_error("qk_group_build_csr: H2D copy failed");
                    rc = QK_ERR_HIP;
                    break;
                }
                dv = (const float *)lead.ctx->stage;
                di = (const int64_t *)si;
            }
            for (int j = 0; j < G && rc == QK_OK; j++) rc = qk_store_csr_chunk(g->m[(size_t)j].store, &b[(size_t)j], dv, di, n, i0);
            if (mem == QK_MEM_HOST)  // the staging chunk is overwritten next
                for (int j = 0; j < G && rc == QK_OK; j++) {
                    hipSetDevice(g->m[(size_t)j].ctx->device);
                    if (hipStreamSynchronize(g->m[(size_t)j].ctx->stream) != hipSuccess) {
                        qk_set_error("qk_group_build_csr: ingest failed on member %d", j);
                        rc = QK_ERR_HIP;
                    }
                }
        }
    }
    int rc_end = rc;
    for (int j = 0; j < begun; j++) {
        const int e = qk_store_csr_end(g->m[(size_t)j].store, &b[(size_t)j], rc);
        if (rc_end == QK_OK) rc_end = e;
    }
    return rc_end;
}

// kmeans_refine_partitions over lists that live on several members: the lists meet in a temporary store on the member that holds
// the first of them (rows move device to device), are refined there exactly as qk_store_refine_lists refines a local set (same
// concatenation order, same arithmetic), and go back to their owners.
int qk_group_refine_lists(qk_group *g, const int64_t *list_nos, int64_t m, float *centroids, int metric, int refinement_iterations,
                          int mem) {
    if (!g || !list_nos || !centroids || m <= 0) QK_FAIL(QK_ERR_INVALID, "qk_store_refine_lists: bad arguments");
    QK_TRY(check_metric(metric));
    bool one = true;
    for (int64_t c = 0; c < m; c++) {
        if (list_nos[c] < 0) QK_FAIL(QK_ERR_NOT_FOUND, "List does not exist in refine_partitions (list %lld)", (long long)list_nos[c]);
        one = one && owner(g, list_nos[c]) == owner(g, list_nos[0]);
    }
    Member *home = owner(g, list_nos[0]);
    if (one) return qk_store_refine_lists(home->store, list_nos, m, centroids, metric, refinement_iterations, mem);
    int64_t mx = 0;
    for (int64_t c = 0; c < m; c++) {
        int64_t sz = 0;
        QK_TRY(qk_group_list_size(g, list_nos[c], &sz));  // "List does not exist" before anything moves
        for (int64_t c2 = 0; c2 < c; c2++)
            if (list_nos[c2] == list_nos[c]) QK_FAIL(QK_ERR_INVALID, "qk_store_refine_lists: duplicate list %lld", (long long)list_nos[c]);
        mx = std::max(mx, sz);
    }
    qk_store *tmp = nullptr;
    QK_TRY(qk_store_create(home->ctx, g->d, &tmp));
    struct Cleanup {
        qk_store *s;
        std::vector<std::pair<int, void *>> bufs;
        ~Cleanup() {
            for (auto &b : bufs) {
                hipSetDevice(b.first);
                hipFree(b.second);
            }
            qk_store_destroy(s);
        }
    } cl{tmp, {}};
    // a transfer buffer on every member involved (the rows of one list at a time)
    std::vector<float *> tv((size_t)g->G, nullptr);
    std::vector<int64_t *> ti((size_t)g->G, nullptr);
    auto bufs_of = [&](Member *o) -> int {
        const size_t j = (size_t)(o - g->m.data());
        if (tv[j] || mx == 0) return QK_OK;
        QK_HIP(hipSetDevice(o->ctx->device));
        QK_HIP(hipMalloc((void **)&tv[j], (size_t)mx * g->d * 4));
        cl.bufs.push_back({o->ctx->device, tv[j]});
        QK_HIP(hipMalloc((void **)&ti[j], (size_t)mx * 8));
        cl.bufs.push_back({o->ctx->device, ti[j]});
        return QK_OK;
    };
    for (int64_t c = 0; c < m; c++) {
        const int64_t p = list_nos[c];
        Member *o = owner(g, p);
        const size_t j = (size_t)(o - g->m.data());
        int64_t sz = 0;
        QK_TRY(qk_store_list_size(o->store, p, &sz));
        QK_TRY(qk_store_add_list(tmp, p));
        if (sz == 0) continue;
        QK_TRY(bufs_of(o));
        QK_TRY(qk_store_get_list(o->store, p, tv[j], ti[j], QK_MEM_DEVICE));
        QK_HIP(hipSetDevice(o->ctx->device));
        QK_HIP(hipStreamSynchronize(o->ctx->stream));
        QK_TRY(qk_store_add_entries(tmp, p, sz, ti[j], tv[j], QK_MEM_DEVICE));
    }
    QK_TRY(qk_store_refine_lists(tmp, list_nos, m, centroids, metric, refinement_iterations, mem));
    QK_TRY(bufs_of(home));
    const size_t hj = (size_t)(home - g->m.data());
    // the refined lists can be longer than any of the old ones
    int64_t mx2 = 0;
    for (int64_t c = 0; c < m; c++) mx2 = std::max(mx2, tmp->parts[(size_t)list_nos[c]].size);
    float *hv = tv[hj];
    int64_t *hi = ti[hj];
    if (mx2 > mx) {
        QK_HIP(hipSetDevice(home->ctx->device));
        QK_HIP(hipMalloc((void **)&hv, (size_t)mx2 * g->d * 4));
        cl.bufs.push_back({home->ctx->device, hv});
        QK_HIP(hipMalloc((void **)&hi, (size_t)mx2 * 8));
        cl.bufs.push_back({home->ctx->device, hi});
    }
    for (int64_t c = 0; c < m; c++) {  // partition_manager.cpp:481-483: the old partition goes, the new one takes its number
        const int64_t p = list_nos[c];
        Member *o = owner(g, p);
        const int64_t sz = tmp->parts[(size_t)p].size;
        QK_TRY(qk_store_remove_list(o->store, p));
        QK_TRY(qk_store_add_list(o->store, p));
        if (sz == 0) continue;
        QK_TRY(qk_store_get_list(tmp, p, hv, hi, QK_MEM_DEVICE));
        QK_HIP(hipSetDevice(home->ctx->device));
        QK_HIP(hipStreamSynchronize(home->ctx->stream));
        QK_TRY(qk_store_add_entries(o->store, p, sz, hi, hv, QK_MEM_DEVICE));
    }
    return QK_OK;
}

// ---- search --------------------------------------------------------------------------------------------------------------------
int qk_group_scan(qk_group *g, const float *x, int64_t Q, const int64_t *pids, int P, int k, int metric, int64_t *out_ids,
                  float *out_dist, int mem, qk_timing *timing) {
    if (!g || (Q > 0 && (!x || !out_ids))) QK_FAIL(QK_ERR_INVALID, "qk_group_scan: null argument");
    if (P < 0 || (P > 0 && !pids)) QK_FAIL(QK_ERR_INVALID, "qk_group_scan: bad partition id list");
    if (k <= 0) QK_FAIL(QK_ERR_INVALID, "qk_group_scan: k must be positive");
    QK_TRY(check_metric(metric));
    if (mem == QK_MEM_HOST && pids) {
        // the reference throws "List does not exist" from get_codes (dynamic_inverted_list.cpp:76-82)
        for (int64_t i = 0; i < Q * (int64_t)P; i++) {
            const int64_t p = pids[i];
            if (p < 0) continue;
            const qk_store *s = owner(g, p)->store;
            if (p >= (int64_t)s->parts.size() || !s->parts[(size_t)p].present)
                QK_FAIL(QK_ERR_NOT_FOUND, "List does not exist in get_codes (list %lld)", (long long)p);
        }
    }
    return group_search(g, nullptr, x, Q, pids, P, 0, k, metric, out_ids, out_dist, mem, timing);
}

// QueryCoordinator::search with recall_target > 0 and workers (the APS hook of worker_scan, query_coordinator.cpp:364-428, whose
// outcome depends on thread timing upstream; here: the deterministic walk of qk_search_aps).  The rounds run on the lead -- candidates
// from its replica of the parent, boundary distances, the sequential rule --; a round's list of partitions goes to every member,
// every member scans the pairs whose lists it holds (per-pair results), the lead picks each pair's answer from its owner (peer
// reads) and replays the rule.  Results and partitions visited equal the one-store search.
int qk_group_search_aps(qk_group *g, qk_store *parent, const float *x, int64_t Q, int k, int metric, float recall_target,
                        float recompute_threshold, int use_precomputed, float initial_search_fraction, int64_t *out_ids, float *out_dist,
                        int32_t *out_nscanned, int mem, qk_timing *timing) {
    if (!g || !parent) QK_FAIL(QK_ERR_INVALID, "qk_search_aps: group / parent is null (adaptive search needs a parent index)");
    if (Q > 0 && (!x || !out_ids)) QK_FAIL(QK_ERR_INVALID, "qk_group_search_aps: null argument");
    QK_TRY(check_metric(metric));
    if (Q <= 0) {
        if (timing) memset(timing, 0, sizeof(*timing));
        return QK_OK;
    }
    const int G = g->G, d = g->d;
    Member &lead = g->m[0];
    QK_TRY(sync_parent(g, parent));
    for (auto &mb : g->m) QK_TRY(qk_check_overflow(mb.ctx));
    const int kk = k <= 0 ? 1 : k;
    std::vector<const float4 *> xq4((size_t)G, nullptr);
    std::vector<const float *> xn((size_t)G, nullptr);
    size_t bx = 0, bp = 0, bt = 0, bi = 0;
    const qk_aps_scan_fn scan = [&](const qk_aps_round &r) -> int {
        const size_t npairs = (size_t)r.Q * r.CH;
        if (r.round == 0) {  // (sized for the longest round of the call)
            const size_t npairs_max = (size_t)r.Q * std::max(r.CH, r.CH_max);
            bx = al256((size_t)r.Q * d * 4);
            bp = al256(npairs_max * 8);
            bt = al256((size_t)r.Q * 4);
            bi = al256(npairs_max * r.k * 8);
            QK_TRY(reserve_call_buffers(g, bx + bp + bt + bi + al256(npairs_max * r.k * 4), 0));
        }
        auto xb = [&](Member &mb) { return (float *)mb.buf; };
        auto pb = [&](Member &mb) { return (int64_t *)(mb.buf + bx); };
        auto tb = [&](Member &mb) { return (uint32_t *)(mb.buf + bx + bp); };
        auto ib = [&](Member &mb) { return (int64_t *)(mb.buf + bx + bp + bt); };
        auto kb = [&](Member &mb) { return (float *)(mb.buf + bx + bp + bt + bi); };
        QK_HIP(hipSetDevice(lead.ctx->device));
        QK_HIP(hipEventRecord(g->ev_x, lead.ctx->stream));  // the round's list and bounds are in place on the lead
        GatherArgs ga;
        ga.G = G;
        for (int j = 0; j < G; j++) {
            Member &mb = g->m[(size_t)j];
            QK_HIP(hipSetDevice(mb.ctx->device));
            hipStream_t st = mb.ctx->stream;
            if (j > 0) QK_HIP(hipStreamWaitEvent(st, g->ev_x, 0));
            const float *xj = r.x;
            if (j > 0) {
                if (r.round == 0) QK_HIP(hipMemcpyAsync(xb(mb), r.x, (size_t)r.Q * d * 4, hipMemcpyDefault, st));
                xj = xb(mb);
            }
            if (r.round == 0) {
                if (j == 0) {
                    xq4[0] = r.xq4;
                    xn[0] = r.xn;
                } else {
                    QK_TRY(qk_prep_queries(mb.ctx, xj, r.Q, d, &xq4[(size_t)j], &xn[(size_t)j]));
                }
            }
            const int64_t *pj = r.round_pids;
            const uint32_t *tj = r.run_tau;
            if (j > 0) {
                QK_HIP(hipMemcpyAsync(pb(mb), r.round_pids, npairs * 8, hipMemcpyDefault, st));
                QK_HIP(hipMemcpyAsync(tb(mb), r.run_tau, (size_t)r.Q * 4, hipMemcpyDefault, st));
                pj = pb(mb);
                tj = tb(mb);
            }
            qk_scan_args sa;
            sa.x = xj;
            sa.xq4 = xq4[(size_t)j];
            sa.xn = xn[(size_t)j];
            sa.Q = r.Q;
            sa.pids = pj;
            sa.P = r.CH;
            sa.k = r.k;
            sa.metric = r.metric;
            sa.out_ids = ib(mb);
            sa.out_dist = kb(mb);
            sa.per_pair = true;
            sa.form_salt = 1 + std::min(r.round, 6);
            sa.tau_init = tj;
            sa.seed_first = r.round == 0;  // (a member that holds the query's nearest list learns a bound for its pairs of that query)
            sa.sqrt_l2 = false;
            QK_TRY(qk_scan_device(mb.ctx, mb.store, sa, nullptr, 4));
            if (j > 0) QK_HIP(hipEventRecord(mb.ev_done, st));
            ga.ids[j] = ib(mb);
            ga.key[j] = kb(mb);
        }
        QK_HIP(hipSetDevice(lead.ctx->device));
        for (int j = 1; j < G; j++) QK_HIP(hipStreamWaitEvent(lead.ctx->stream, g->m[(size_t)j].ev_done, 0));
        hipLaunchKernelGGL(k_aps_gather, dim3(grid_for((int64_t)npairs * r.k)), dim3(256), 0, lead.ctx->stream, r.round_pids, (int64_t)npairs,
                           r.k, r.metric, ga, r.pr_ids, r.pr_key);
        QK_HIP(hipGetLastError());
        return QK_OK;
    };
    // (the lead's context returns merge keys to the group's searches; the adaptive search's own answer is a plain distance)
    lead.ctx->squared_l2 = false;
    const int rc = qk_aps_run(lead.ctx, lead.parent, qk_group_nlist(g), d, x, Q, kk, metric, recall_target, recompute_threshold,
                              use_precomputed, initial_search_fraction, out_ids, out_dist, out_nscanned, mem, timing, scan);
    lead.ctx->squared_l2 = true;
    return rc;
}

int qk_group_search(qk_group *g, qk_store *parent, const float *x, int64_t Q, int nprobe, int k, int metric, int64_t *out_ids,
                    float *out_dist, int mem, qk_timing *timing) {
    if (!g || !parent || (Q > 0 && (!x || !out_ids))) QK_FAIL(QK_ERR_INVALID, "qk_group_search: null argument");
    if (k <= 0) QK_FAIL(QK_ERR_INVALID, "qk_group_search: k must be positive");
    if (nprobe <= 0) QK_FAIL(QK_ERR_INVALID, "qk_group_search: nprobe must be positive");
    QK_TRY(check_metric(metric));
    return group_search(g, parent, x, Q, nullptr, 0, nprobe, k, metric, out_ids, out_dist, mem, timing);
}

}  // extern "C"

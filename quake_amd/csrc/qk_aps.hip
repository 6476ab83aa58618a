// qk_aps.hip -- adaptive partition scanning (recall-target search) on the device.
//
// Replaces, for SearchParams::recall_target > 0 and batched_scan == false:
//   QueryCoordinator::search          src/cpp/src/query_coordinator.cpp:612-657  (M = nlist * initial_search_fraction candidates)
//   serial_scan, use_aps branch       src/cpp/src/query_coordinator.cpp:471-611  (scan, radius, recall profile, stop)
//   compute_boundary_distances        src/cpp/include/geometry.h:57-113
//   incomplete_beta (+ table/lookup)  src/cpp/include/geometry.h:115-211
//   log_hyperspherical_cap_volume     src/cpp/include/geometry.h:247-295
//   compute_recall_profile            src/cpp/include/geometry.h:345-407
//
// The reference walks one query's candidate partitions one after the other on a CPU thread.  Here the whole batch
// advances in ROUNDS: a round scans, for every query that has not stopped, its next few ranked partitions in one launch
// of the scan pipeline with the results of every (query, partition) pair kept apart (qk_scan_args::per_pair); one wave
// per query then replays the reference's sequential rule over those partitions in rank order -- merge, radius,
// profile, estimate, stop -- so the answer and the number of partitions counted as scanned are those of the sequential
// walk; partitions scanned past the stopping point in the same round are discarded.  How many partitions a query takes
// into the next round is predicted from its current profile.
#include "qk_internal.h"
#include "qk_device.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int APS_NX = 1001;      // geometry.h:7  NUM_X_VALUES
constexpr double APS_STOP = 1.0e-8;   // geometry.h:9
constexpr double APS_TINY = 1.0e-30;  // geometry.h:10
// The schedule of the rounds does not change any result (partitions scanned past the stopping point are discarded); it decides how
// many launches a batch takes.  Measured on the bench mixture (10M x 128, 4096 lists, 81 candidates, target 0.9 / 0.99: the walk
// visits 48 / 53 lists): first round 2 / cap 32 -> 4 rounds, 3.27 / 3.41 ms; 4 / 48 -> 3 rounds, 3.14 / 3.17; 4 / 80 -> 2 rounds,
// 2.85 / 3.02; a longer first round loses (8 / 64: 3.29, 16 / 64: 4.20 -- its lists are scanned without a bound).
constexpr int APS_CH = 80;        // partitions per query and round (upper bound)
constexpr int APS_FIRST = 4;      // partitions of the first round

// ---- regularised incomplete beta I_x(a, b): Lentz's continued fraction, the published algorithm geometry.h uses.
// One definition serves the host (table of the precomputed path) and the device (IP metric / use_precomputed = false).
__host__ __device__ inline double beta_cf(double a, double b, double x) {
    // caller guarantees 0 <= x <= (a+1)/(a+b+2)
    const double lbeta = lgamma(a) + lgamma(b) - lgamma(a + b);
    const double front = exp(log(x) * a + log(1.0 - x) * b - lbeta) / a;
    double f = 1.0, c = 1.0, dd = 0.0;
    for (int i = 0; i <= 200; ++i) {
        const int m = i / 2;
        double num;
        if (i == 0)
            num = 1.0;
        else if ((i & 1) == 0)
            num = (m * (b - m) * x) / ((a + 2.0 * m - 1.0) * (a + 2.0 * m));
        else
            num = -((a + m) * (a + b + m) * x) / ((a + 2.0 * m) * (a + 2.0 * m + 1));
        dd = 1.0 + num * dd;
        if (fabs(dd) < APS_TINY) dd = APS_TINY;
        dd = 1.0 / dd;
        c = 1.0 + num / c;
        if (fabs(c) < APS_TINY) c = APS_TINY;
        const double cd = c * dd;
        f *= cd;
        if (fabs(1.0 - cd) < APS_STOP) return front * (f - 1.0);
    }
    return INFINITY;  // did not converge
}
__host__ __device__ inline double inc_beta(double a, double b, double x) {
    if (x < 0.0 || x > 1.0) return INFINITY;
    // symmetry I_x(a,b) = 1 - I_{1-x}(b,a): the two thresholds sum to 1, so the swapped argument is below its own
    if (x > (a + 1.0) / (a + b + 2.0)) return 1.0 - beta_cf(b, a, 1.0 - x);
    return beta_cf(a, b, x);
}

// table lookup with linear interpolation (geometry.h:182-211); NaN clamps to 1 like std::max(0, std::min(1, x))
__device__ inline double beta_lookup(const double *table, double x) {
    const double t = (x < 1.0) ? x : 1.0;
    x = (0.0 < t) ? t : 0.0;
    const double scaled = x * (APS_NX - 1);
    int xi = (int)scaled;
    xi = xi > APS_NX - 2 ? APS_NX - 2 : xi;
    xi = xi < 0 ? 0 : xi;
    const double y1 = table[xi], y2 = table[xi + 1];
    const double dx = 1.0 / (APS_NX - 1);
    const double x1 = xi * dx;
    return y1 + (x - x1) * (y2 - y1) / dx;
}

// log of the cap-volume ratio (geometry.h:247-295 with ratio = true)
__device__ inline double log_cap_volume(double radius, double bdist, int d, bool precomputed, bool euclid, const double *table) {
    double h = radius - bdist;
    const double t = (h < 2 * radius) ? h : 2 * radius;
    h = (0.0 < t) ? t : 0.0;
    if (euclid) {
        const double x = sqrt((2 * radius * h - h * h) / (radius * radius));
        const double ib = precomputed ? beta_lookup(table, x) : inc_beta((d + 1.0) / 2.0, 0.5, x);
        if (ib <= 0.0 || isnan(ib) || isinf(ib)) return -INFINITY;
        return log(0.5) + log(ib);
    }
    const double s1 = sin(radius / 2.0), s2 = sin(bdist / 2.0);
    const double l1 = log(inc_beta((d - 1) / 2.0, 0.5, s1 * s1));
    const double l2 = log(inc_beta((d - 1) / 2.0, 0.5, s2 * s2));
    return log(0.5) + l1 - l2;
}

// element (row, col) of the tile-major arena (qk_internal.h)
__device__ inline float arena_at(const float *vecs, int nblk, int64_t row, int col) {
    const int64_t tile = row >> 4;
    const int r = (int)(row & 15), c = col >> 4, t = col & 15;
    return vecs[(((tile * nblk + c) * 64 + (t & 3) * 16 + r) << 2) + (t >> 2)];
}

// ---- candidate rows + boundary distances (geometry.h:57-113): one thread per (query, candidate) ----------------------
struct BoundaryParams {
    const float *x;          // [Q][d]
    const int64_t *pids;     // [Q][M] ranked candidate partitions
    const int32_t *row_of;   // partition id -> arena row of its centroid (or -1)
    int64_t n_row_of;
    const float *cvecs;      // parent arena
    int cnblk;
    int64_t Q;
    int M, d, euclid;
    float *bd;               // [Q][M]
};

__global__ __launch_bounds__(256) void k_aps_boundary(BoundaryParams B) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B.Q * B.M) return;
    const int64_t q = idx / B.M;
    const int j = (int)(idx - q * B.M);
    if (j == 0) {
        B.bd[idx] = -1.0f;
        return;
    }
    const int64_t p0 = B.pids[q * B.M], pj = B.pids[idx];
    const int64_t r0 = (p0 >= 0 && p0 < B.n_row_of) ? B.row_of[p0] : -1;
    const int64_t rj = (pj >= 0 && pj < B.n_row_of) ? B.row_of[pj] : -1;
    if (r0 < 0 || rj < 0) {
        B.bd[idx] = -1.0f;
        return;
    }
    const float *xq = B.x + q * B.d;
    // a row's 16 columns of block c are four float4 of the tile-major arena (float4 g' holds columns 16c + {g', 4+g', 8+g', 12+g'}):
    // read as float4 -- a quarter of the loads of an element-by-element walk -- and fed to the chains in column order
    const float4 *v4 = (const float4 *)B.cvecs;
    const int64_t t0 = r0 >> 4, tj = rj >> 4;
    const int rr0 = (int)(r0 & 15), rrj = (int)(rj & 15);
    const int nblk = B.cnblk;
    float out;
    if (B.euclid) {
        float a2 = 0.0f, dot = 0.0f;
        for (int c = 0; c < nblk; c++) {
            float4 f0[4], fj[4];
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                f0[gq] = v4[(t0 * nblk + c) * 64 + gq * 16 + rr0];
                fj[gq] = v4[(tj * nblk + c) * 64 + gq * 16 + rrj];
            }
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int i = 16 * c + t;
                if (i < B.d) {
                    const float4 a = f0[t & 3], b4 = fj[t & 3];
                    const int tp = t >> 2;
                    const float c0 = tp == 0 ? a.x : tp == 1 ? a.y : tp == 2 ? a.z : a.w;
                    const float cj = tp == 0 ? b4.x : tp == 1 ? b4.y : tp == 2 ? b4.z : b4.w;
                    const float v = cj - c0;       // line vector c_j - c_0
                    const float res = xq[i] - c0;  // residual q - c_0
                    a2 = __fmaf_rn(v, v, a2);
                    dot = __fmaf_rn(res, v, dot);
                }
            }
        }
        const float a = sqrtf(a2);
        out = fabsf(dot - 0.5f * a2) / a;
    } else {
        float n2 = 0.0f;
        for (int c = 0; c < nblk; c++) {
            float4 f0[4], fj[4];
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                f0[gq] = v4[(t0 * nblk + c) * 64 + gq * 16 + rr0];
                fj[gq] = v4[(tj * nblk + c) * 64 + gq * 16 + rrj];
            }
#pragma unroll
            for (int t = 0; t < 16; t++) {
                if (16 * c + t < B.d) {
                    const float4 a = f0[t & 3], b4 = fj[t & 3];
                    const int tp = t >> 2;
                    const float c0 = tp == 0 ? a.x : tp == 1 ? a.y : tp == 2 ? a.z : a.w;
                    const float cj = tp == 0 ? b4.x : tp == 1 ? b4.y : tp == 2 ? b4.z : b4.w;
                    const float mid = c0 + (cj - c0) / 2.0f;
                    n2 = __fmaf_rn(mid, mid, n2);
                }
            }
        }
        const float nrm = sqrtf(n2);
        float ang = 0.0f;
        for (int c = 0; c < nblk; c++) {
            float4 f0[4], fj[4];
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                f0[gq] = v4[(t0 * nblk + c) * 64 + gq * 16 + rr0];
                fj[gq] = v4[(tj * nblk + c) * 64 + gq * 16 + rrj];
            }
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int i = 16 * c + t;
                if (i < B.d) {
                    const float4 a = f0[t & 3], b4 = fj[t & 3];
                    const int tp = t >> 2;
                    const float c0 = tp == 0 ? a.x : tp == 1 ? a.y : tp == 2 ? a.z : a.w;
                    const float cj = tp == 0 ? b4.x : tp == 1 ? b4.y : tp == 2 ? b4.z : b4.w;
                    const float mid = (c0 + (cj - c0) / 2.0f) / nrm;
                    ang = __fmaf_rn(xq[i], mid, ang);
                }
            }
        }
        out = (float)acos((double)ang);
    }
    B.bd[idx] = out;
}

// ---- per-round pid matrix ---------------------------------------------------------------------------------------------
// (CH = this round's row length: no query wants more)
__global__ void k_aps_round_pids(const int64_t *pids, const int32_t *next_p, const int32_t *want, int64_t Q, int M, int CH,
                                 int64_t *round_pids) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Q * CH) return;
    const int64_t q = idx / CH;
    const int i = (int)(idx - q * CH);
    const int p = next_p[q] + i;
    round_pids[idx] = (i < want[q] && p < M) ? pids[q * M + p] : -1;
}

// ---- the sequential rule, one wave per query ------------------------------------------------------------------------
struct UpdateParams {
    int64_t Q;
    int M, k, d, CH, CHr, metric;  // CH: most partitions a query takes into a round; CHr: row length of THIS round's pair arrays
    float recall_target, recompute_threshold;
    int precomputed;
    const double *table;
    const int64_t *pids;      // [Q][M]
    const float *bd;          // [Q][M]
    float *probs;             // [Q][M]  (persisted between rounds)
    const int64_t *pr_ids;    // [Q*CHr][k] results of this round's pairs
    const float *pr_key;      // [Q*CHr][k] squared L2 / inner product
    uint32_t *run_ord;        // [Q][k]
    int64_t *run_id;          // [Q][k]
    int32_t *run_cnt, *have_probs, *next_p, *want, *nscan;
    uint32_t *run_tau;        // [Q] ~(k-th key of the running result), 0 while it holds fewer than k: next round's bound
    float *radius;
    int32_t *n_active;        // [0] queries that go on after this round, [1] workgroups that have finished this round (ticket)
    int32_t *host_flag;       // host-mapped: the last workgroup of the round leaves n_active + 1 here (0 = round still running)
    int64_t *out_ids;         // [Q][k]
    float *out_dist;          // [Q][k]
    int sqrt_l2;
};

// End of a query's round.  The host decides on the next round from ONE number -- how many queries go on -- and reads it from
// host-mapped memory without synchronising the stream: every workgroup takes a ticket, the last one of the round publishes the count
// (+ 1: 0 means "round still running") with a system-scope store and leaves both counters at zero for the next round.
__device__ __forceinline__ void aps_round_done(const UpdateParams &U, int lane, bool goes_on) {
    if (lane != 0) return;
    if (goes_on) atomicAdd(&U.n_active[0], 1);
    __threadfence();
    const int t = atomicAdd(&U.n_active[1], 1);
    if (t == (int)U.Q - 1) {
        const int left = __hip_atomic_load(&U.n_active[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        U.n_active[0] = 0;
        U.n_active[1] = 0;
        __threadfence();
        __hip_atomic_store(U.host_flag, left + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int MAXCH>
__global__ __launch_bounds__(64) void k_aps_update(UpdateParams U) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int k = U.k, M = U.M;
    const int w = U.want[q];
    if (w <= 0) {  // stopped in an earlier round
        aps_round_done(U, lane, false);
        return;
    }
    int64_t *pool_id = (int64_t *)smem;                                  // [2k]
    uint32_t *pool_ord = (uint32_t *)(smem + (size_t)2 * k * 8);         // [2k]
    float *probs = (float *)(smem + (size_t)2 * k * 12);                 // [M]
    float *bdl = probs + M;                                              // [M] this query's boundary distances
    const bool euclid = U.metric == QK_METRIC_L2;
    int cnt = U.run_cnt[q];
    for (int e = lane; e < cnt; e += 64) {
        pool_ord[e] = U.run_ord[q * k + e];
        pool_id[e] = U.run_id[q * k + e];
    }
    bool have = U.have_probs[q] != 0;
    if (have)
        for (int jj = lane; jj < M; jj += 64) probs[jj] = U.probs[q * M + jj];
    for (int jj = lane; jj < M; jj += 64) bdl[jj] = U.bd[q * M + jj];
    float qr = U.radius[q];
    const int p0 = U.next_p[q];
    bool stop = false;
    int nscan = p0;
    // The walk is a chain of dependent steps.  What it needs from global memory is found out up front, 64 pairs per round trip:
    // which steps have a partition at all (pids != -1) and which pairs hold any entry (a pair's entries are sorted, valid ones
    // first: entry 0 decides) -- most pairs of a later round hold none, their steps touch LDS only.  The entries of the pairs that
    // do hold some are requested one non-empty pair ahead.
    uint64_t *m_valid = (uint64_t *)(bdl + M);            // [ceil(w / 64)] step has a partition  (2k * 12 + 8M bytes in: 8-byte aligned)
    uint64_t *m_full = m_valid + ((U.CHr + 63) >> 6);      // [ceil(w / 64)] ... and its pair holds entries
    const int wsteps = min(w, M - p0);
    for (int base = 0; base < wsteps; base += 64) {
        const int i = base + lane;
        bool va = false, fu = false;
        if (i < wsteps) {
            va = U.pids[q * M + p0 + i] != -1;
            fu = va && U.pr_ids[(q * U.CHr + i) * k] >= 0;
        }
        const uint64_t bv = __ballot(va), bf = __ballot(fu);
        if (lane == 0) {
            m_valid[base >> 6] = bv;
            m_full[base >> 6] = bf;
        }
    }
    __builtin_amdgcn_wave_barrier();
    auto next_full = [&](int from) -> int {  // first step >= from whose pair holds entries, or wsteps
        for (int wd = from >> 6; wd < ((wsteps + 63) >> 6); wd++) {
            uint64_t bits = m_full[wd];
            if (wd == (from >> 6)) bits &= ~0ull << (from & 63);
            if (bits) return (wd << 6) + __ffsll((unsigned long long)bits) - 1;
        }
        return wsteps;
    };
    int64_t pf_id = -1;
    float pf_key = 0.0f;
    int pf_at = next_full(0);
    auto fetch = [&](int i) {
        pf_id = -1;
        if (i < wsteps && lane < k) {
            pf_id = U.pr_ids[(q * U.CHr + i) * k + lane];
            pf_key = U.pr_key[(q * U.CHr + i) * k + lane];
        }
    };
    fetch(pf_at);
    // the estimate of step p is the sum of probs[0 .. p) taken in order from 0.0f: while the profile stands, the chain is continued
    // from where the previous step left it (the same additions in the same order)
    float est_run = 0.0f;
    int est_n = 0;
    for (int i = 0; i < wsteps; i++) {
        const int p = p0 + i;
        nscan = p + 1;
        if (!((m_valid[i >> 6] >> (i & 63)) & 1ull)) continue;  // query_coordinator.cpp:540
        int64_t c_id = -1;
        float c_key = 0.0f;
        const bool full = i == pf_at;
        if (full) {
            c_id = pf_id;
            c_key = pf_key;
            pf_at = next_full(i + 1);
            fetch(pf_at);
        }
        // merge this partition's top-k into the running one
        const int64_t *nid = U.pr_ids + (q * U.CHr + i) * k;
        const float *nkey = U.pr_key + (q * U.CHr + i) * k;
        int added = 0;
        for (int base = 0; full && base < k; base += 64) {
            const int e = base + lane;
            int64_t id = -1;
            uint32_t o = 0xFFFFFFFFu;
            if (e < k) {
                float v;
                if (base == 0) {
                    id = c_id;
                    v = c_key;
                } else {
                    id = nid[e];
                    v = nkey[e];
                }
                o = euclid ? ord_from_l2(v) : ord_from_ip(v);
            }
            const bool ok = id >= 0;
            const uint64_t m = __ballot(ok);
            if (ok) {
                const int sl = cnt + __popcll(m & ((1ull << lane) - 1ull));
                pool_ord[sl] = o;
                pool_id[sl] = id;
            }
            cnt += __popcll(m);
            added += __popcll(m);
        }
        // (a pair that brought nothing leaves the pool as the last step sorted it)
        if (added) cnt = compact_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane);
        // radius = k-th distance, or the buffer's sentinel while it holds fewer than k (list_scanning.h:57-63,187-191)
        float cur;
        if (cnt >= k) {
            const uint32_t ok_ = pool_ord[k - 1];
            cur = euclid ? sqrtf(__uint_as_float(ok_)) : ip_from_ord(ok_);
        } else {
            cur = euclid ? 3.402823466e+38f : -INFINITY;
        }
        const float change = fabsf(cur - qr) / cur;
        if (change > U.recompute_threshold) {
            qr = cur;
            // compute_recall_profile (geometry.h:345-407)
            for (int jj = lane; jj < M; jj += 64) {
                float pj = 0.0f;
                if (jj >= 1) {
                    const float b = bdl[jj];
                    if (!(b >= qr)) {
                        const double vr = exp(log_cap_volume((double)qr, (double)b, U.d, U.precomputed != 0, euclid, U.table));
                        pj = (float)((vr > 0.0) ? vr : 0.0);
                    }
                }
                probs[jj] = pj;
            }
            __builtin_amdgcn_wave_barrier();
            const float p1 = probs[1];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) probs[0] = (float)(2.0 * p1);
            __builtin_amdgcn_wave_barrier();
            double sum = 0.0;  // sequential, in candidate order (every lane runs the same chain)
            for (int jj = 0; jj < M; jj++) sum += (double)probs[jj];
            __builtin_amdgcn_wave_barrier();
            if (sum > 0.0) {
                for (int jj = lane; jj < M; jj += 64) probs[jj] = (float)((double)probs[jj] / sum);
            } else {
                for (int jj = lane; jj < M; jj += 64) probs[jj] = (float)(1.0 / M);
            }
            __builtin_amdgcn_wave_barrier();
            have = true;
            est_run = 0.0f;
            est_n = 0;
        }
        float est = 0.0f;
        if (have) {
            for (; est_n < p; est_n++) est_run += probs[est_n];
            est = est_run;
        }
        if (est >= U.recall_target) {
            stop = true;
            break;
        }
    }
    const int np = min(p0 + w, M);
    const bool done = stop || np >= M;
    if (done) {
        if (!stop) nscan = M;
        for (int e = lane; e < k; e += 64) {
            int64_t oid = -1;
            float od = euclid ? INFINITY : -INFINITY;
            if (e < cnt) {
                oid = pool_id[e];
                const uint32_t o = pool_ord[e];
                if (euclid) {
                    const float d2 = __uint_as_float(o);
                    od = U.sqrt_l2 ? sqrtf(d2) : d2;
                } else {
                    od = ip_from_ord(o);
                }
            }
            U.out_ids[q * k + e] = oid;
            if (U.out_dist) U.out_dist[q * k + e] = od;
        }
        if (lane == 0) {
            U.want[q] = 0;
            U.nscan[q] = nscan;
        }
        aps_round_done(U, lane, false);
        return;
    }
    // carry the state into the next round
    for (int e = lane; e < cnt; e += 64) {
        U.run_ord[q * k + e] = pool_ord[e];
        U.run_id[q * k + e] = pool_id[e];
    }
    if (have)
        for (int jj = lane; jj < M; jj += 64) U.probs[q * M + jj] = probs[jj];
    // partitions for the next round: up to where the current profile says the estimate will reach the target
    int wn = U.CH;
    if (have) {
        float est = 0.0f;
        int t = 0;
        for (; t < M; t++) {
            if (t >= np && est >= U.recall_target) break;  // the walk would stop AT partition t (after scanning it)
            est += probs[t];
        }
        wn = min(U.CH, max(1, t - np + 1));
    }
    wn = min(wn, M - np);
    if (lane == 0) {
        // nothing worse than the running k-th can enter a later running top-k (it only shrinks): safe for every partition
        // of the next rounds.  (A bound learnt INSIDE a round is not: it could remove entries that belong to the running
        // result of an earlier step of the walk and change that step's radius.)
        U.run_tau[q] = cnt >= k ? ~pool_ord[k - 1] : 0u;
        U.run_cnt[q] = cnt;
        U.have_probs[q] = have ? 1 : 0;
        U.radius[q] = qr;
        U.next_p[q] = np;
        U.want[q] = wn;
    }
    aps_round_done(U, lane, true);
}

__global__ void k_aps_init(int64_t Q, int M, int first, int metric, int32_t *run_cnt, int32_t *have_probs, int32_t *next_p, int32_t *want,
                           int32_t *nscan, float *radius, uint32_t *run_tau) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    run_cnt[q] = 0;
    run_tau[q] = 0u;
    have_probs[q] = 0;
    next_p[q] = 0;
    want[q] = min(first, M);  // (at least 2: the estimate is empty before the second partition, nobody stops earlier)
    nscan[q] = 0;
    radius[q] = metric == QK_METRIC_L2 ? 1000000.0f : -1000000.0f;  // query_coordinator.cpp:523-527
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

extern "C" int qk_search_aps(qk_ctx *ctx, qk_store *parent, qk_store *s, const float *x, int64_t Q, int k, int metric,
                             float recall_target, float recompute_threshold, int use_precomputed, float initial_search_fraction,
                             int64_t *out_ids, float *out_dist, int32_t *out_nscanned, int mem, qk_timing *timing) {
    if (!ctx || !s || !parent) QK_FAIL(QK_ERR_INVALID, "qk_search_aps: ctx / parent / store is null (adaptive search needs a parent index)");
    // a round's scan: the probed (query, list) pairs against the ONE store, results kept apart per pair
    const qk_aps_scan_fn scan = [ctx, s](const qk_aps_round &r) -> int {
        qk_scan_args sa;
        sa.x = r.x;
        sa.xq4 = r.xq4;
        sa.xn = r.xn;
        sa.Q = r.Q;
        sa.pids = r.round_pids;
        sa.P = r.CH;
        sa.k = r.k;
        sa.metric = r.metric;
        sa.out_ids = r.pr_ids;
        sa.out_dist = r.pr_key;
        sa.per_pair = true;
        sa.form_salt = 1 + std::min(r.round, 6);  // every round is a workload of its own (fewer queries go on, with other bounds)
        sa.tau_init = r.run_tau;
        sa.seed_first = r.round == 0;  // (no running result yet: the bound comes from a sample of every query's nearest list)
        sa.sqrt_l2 = false;  // merge keys: squared distances
        return qk_scan_device(ctx, s, sa, nullptr, 4);
    };
    return qk_aps_run(ctx, parent, s->nlist, s->d, x, Q, k, metric, recall_target, recompute_threshold, use_precomputed,
                      initial_search_fraction, out_ids, out_dist, out_nscanned, mem, timing, scan);
}

// The rounds of an adaptive search on `ctx` (candidates, boundary distances, per-round list of partitions, the sequential rule);
// WHO scans a round's pairs is the caller's: one store (qk_search_aps), or the members of a device group (qk_group_search_aps).
int qk_aps_run(qk_ctx *ctx, qk_store *parent, int64_t nlist, int d, const float *x, int64_t Q, int k, int metric, float recall_target,
               float recompute_threshold, int use_precomputed, float initial_search_fraction, int64_t *out_ids, float *out_dist,
               int32_t *out_nscanned, int mem, qk_timing *timing, const qk_aps_scan_fn &scan) {
    if (metric != QK_METRIC_L2 && metric != QK_METRIC_IP) QK_FAIL(QK_ERR_INVALID, "Metric type not supported");
    if (!(recall_target > 0.0f)) QK_FAIL(QK_ERR_INVALID, "qk_search_aps: recall_target must be > 0");
    QK_HIP(hipSetDevice(ctx->device));
    if (timing) memset(timing, 0, sizeof(*timing));
    if (Q <= 0) return QK_OK;
    if (k <= 0) k = 1;  // query_coordinator.cpp:490
    if (k > QK_MAX_K) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_search_aps: k=%d exceeds QK_MAX_K=%d", k, QK_MAX_K);
    if (parent->d != d) QK_FAIL(QK_ERR_INVALID, "parent store dimension %d != store dimension %d", parent->d, d);
    // query_coordinator.cpp:638-640: (int)(nlist * initial_search_fraction) in float arithmetic, at least 1
    int M = (int)((float)nlist * initial_search_fraction);
    if (M < 1) M = 1;
    M = (int)std::min<int64_t>(M, parent->ntotal);
    if (M < 2) QK_FAIL(QK_ERR_INVALID, "Boundary distances must have at least 2 partitions to create an estimate.");  // geometry.h:350
    if (M > QK_MAX_NPROBE) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_search_aps: %d candidate partitions exceed QK_MAX_NPROBE=%d", M, QK_MAX_NPROBE);
    hipStream_t st = ctx->stream;
    static const int aps_ch = std::max(2, qk_env_int("QK_APS_CH", APS_CH));
    static const int aps_first = std::max(2, qk_env_int("QK_APS_FIRST", APS_FIRST));
    // rows of a later round: as long as the walk may get (M), memory permitting (1 GiB of per-pair results) -- a query takes what its
    // profile predicts, and a cap under the walk's length costs a further pass over the lists (M = 204 on the bench index: walks
    // of 118 lists on average, cap 80 -> 4 rounds, 3.7 ms)
    const int CH = std::min(M, std::max(aps_ch, (int)std::min<size_t>((size_t)M, ((size_t)1 << 30) / ((size_t)Q * (size_t)k * 12))));

    // ---- partition id -> arena row of its centroid (host mirror of the parent's ids) and the table of the precomputed path
    // (geometry.h:163-180): both live on the device from one call to the next -- rebuilt when the parent changed (its version is
    // bumped by the table sync that follows every change) or the dimension did.  1001 continued fractions and a walk over the
    // parent's ids are 0.2-0.4 ms of host time per call otherwise.
    QK_TRY(qk_store_sync_table(parent));
    if (!ctx->aps_rowof || ctx->aps_rowof_uid != parent->uid || ctx->aps_rowof_version != parent->version) {
        int64_t max_id = -1;
        for (auto &pt : parent->parts)
            if (pt.present)
                for (int64_t id : pt.ids) max_id = std::max(max_id, id);
        if (max_id < 0) QK_FAIL(QK_ERR_INVALID, "qk_search_aps: empty parent index");
        if (max_id > (int64_t)1 << 26) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_search_aps: partition ids above 2^26 are not supported");
        std::vector<int32_t> row_of((size_t)max_id + 1, -1);
        for (auto &pt : parent->parts)
            if (pt.present)
                for (size_t r = 0; r < pt.ids.size(); r++)
                    if (pt.ids[r] >= 0) row_of[(size_t)pt.ids[r]] = (int32_t)(pt.row_off + (int64_t)r);
        QK_HIP(hipStreamSynchronize(ctx->stream));  // (an earlier call's kernels may still read the old map)
        if (row_of.size() > ctx->aps_rowof_cap) {
            if (ctx->aps_rowof) QK_HIP(hipFree(ctx->aps_rowof));
            ctx->aps_rowof = nullptr;
            ctx->aps_rowof_cap = 0;
            const size_t cap = row_of.size() + row_of.size() / 4 + 256;
            QK_HIP(hipMalloc((void **)&ctx->aps_rowof, cap * 4));
            ctx->aps_rowof_cap = cap;
        }
        QK_HIP(hipMemcpy(ctx->aps_rowof, row_of.data(), row_of.size() * 4, hipMemcpyHostToDevice));
        ctx->aps_rowof_n = (int64_t)row_of.size();
        ctx->aps_rowof_uid = parent->uid;
        ctx->aps_rowof_version = parent->version;
    }
    if (!ctx->aps_table || ctx->aps_table_d != d) {
        std::vector<double> table(APS_NX);
        const double dx = 1.0 / (APS_NX - 1);
        const double a = (d + 1.0) / 2.0, b = 0.5;
        for (int i = 0; i < APS_NX; i++) table[i] = inc_beta(a, b, i * dx);
        QK_HIP(hipStreamSynchronize(ctx->stream));
        if (!ctx->aps_table) QK_HIP(hipMalloc((void **)&ctx->aps_table, APS_NX * 8));
        QK_HIP(hipMemcpy(ctx->aps_table, table.data(), APS_NX * 8, hipMemcpyHostToDevice));
        ctx->aps_table_d = d;
    }

    // ---- state ----------------------------------------------------------------------------------------------------------------
    const size_t QM = (size_t)Q * M, Qk = (size_t)Q * k, QCk = (size_t)Q * CH * k;
    size_t need = 0;
    auto take = [&](size_t b) { size_t o = need; need += al256(b); return o; };
    const size_t o_x = take((size_t)Q * d * 4), o_pids = take(QM * 8), o_bd = take(QM * 4), o_probs = take(QM * 4);
    const size_t o_rord = take(Qk * 4), o_rid = take(Qk * 8), o_q = take((size_t)Q * 4 * 7 + 64);
    const size_t o_rp = take((size_t)Q * CH * 8), o_pri = take(QCk * 8), o_prk = take(QCk * 4);
    const size_t o_oi = take(Qk * 8), o_od = take(Qk * 4);
    const size_t o_na = take(256);
    QK_TRY(qk_aps_reserve(ctx, need));
    char *B = ctx->aps;
    const float *dx_;
    int64_t *d_out_ids;
    float *d_out_dist;
    if (mem == QK_MEM_HOST) {
        QK_HIP(hipMemcpyAsync(B + o_x, x, (size_t)Q * d * 4, hipMemcpyHostToDevice, st));
        dx_ = (const float *)(B + o_x);
        d_out_ids = (int64_t *)(B + o_oi);
        d_out_dist = (float *)(B + o_od);
    } else {
        dx_ = x;
        d_out_ids = out_ids;
        d_out_dist = out_dist ? out_dist : (float *)(B + o_od);
    }
    int64_t *pids = (int64_t *)(B + o_pids);
    float *bd = (float *)(B + o_bd), *probs = (float *)(B + o_probs);
    uint32_t *run_ord = (uint32_t *)(B + o_rord);
    int64_t *run_id = (int64_t *)(B + o_rid);
    int32_t *run_cnt = (int32_t *)(B + o_q), *have_probs = run_cnt + Q, *next_p = have_probs + Q, *want = next_p + Q,
            *nscan = want + Q;
    float *radius = (float *)(nscan + Q);
    uint32_t *run_tau = (uint32_t *)(radius + Q);
    int64_t *round_pids = (int64_t *)(B + o_rp), *pr_ids = (int64_t *)(B + o_pri);
    float *pr_key = (float *)(B + o_prk);
    const double *d_table = ctx->aps_table;
    const int32_t *d_row_of = ctx->aps_rowof;
    int32_t *n_active = (int32_t *)(B + o_na);

    // every way out of this function -- also the error exits inside the round loop -- destroys the timing events and leaves no
    // kernel behind that still writes the context's round state (ctx->aps, the flag words): the next call reuses both
    struct CallGuard {
        hipStream_t st;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        bool drain = true;  // set false on the successful way out (which has synchronised or hands the stream on in order)
        ~CallGuard() {
            if (drain && hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError();
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } guard{st};
    hipEvent_t &e0 = guard.e0, &e1 = guard.e1;
    if (timing) {
        QK_HIP(hipEventCreate(&e0));
        QK_HIP(hipEventCreate(&e1));
        QK_HIP(hipEventRecord(e0, st));
    }
    const float4 *xq4 = nullptr;
    const float *xn = nullptr;
    QK_TRY(qk_prep_queries(ctx, dx_, Q, d, &xq4, &xn));
    // ---- candidates: the parent's M nearest centroids (query_coordinator.cpp:643) -------------------------------------------
    {
        qk_scan_args ca;
        ca.x = dx_;
        ca.xq4 = xq4;
        ca.xn = xn;
        ca.Q = Q;
        ca.all_lists = true;
        ca.k = M;
        ca.metric = metric;
        ca.out_ids = pids;
        ca.out_dist = nullptr;
        QK_TRY(qk_scan_device(ctx, parent, ca, nullptr, 0));
    }
    {
        BoundaryParams bp;
        bp.x = dx_;
        bp.pids = pids;
        bp.row_of = d_row_of;
        bp.n_row_of = ctx->aps_rowof_n;
        bp.cvecs = parent->vecs;
        bp.cnblk = parent->nblk;
        bp.Q = Q;
        bp.M = M;
        bp.d = d;
        bp.euclid = metric == QK_METRIC_L2 ? 1 : 0;
        bp.bd = bd;
        hipLaunchKernelGGL(k_aps_boundary, dim3((unsigned)((QM + 255) / 256)), dim3(256), 0, st, bp);
        hipLaunchKernelGGL(k_aps_init, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, st, Q, M, std::min(aps_first, CH), metric, run_cnt, have_probs, next_p,
                           want, nscan, radius, run_tau);
        QK_HIP(hipGetLastError());
    }
    // ---- rounds ---------------------------------------------------------------------------------------------------------------
    UpdateParams up;
    up.Q = Q;
    up.M = M;
    up.k = k;
    up.d = d;
    up.CH = CH;
    up.metric = metric;
    up.recall_target = recall_target;
    up.recompute_threshold = recompute_threshold;
    up.precomputed = use_precomputed ? 1 : 0;
    up.table = d_table;
    up.pids = pids;
    up.bd = bd;
    up.probs = probs;
    up.pr_ids = pr_ids;
    up.pr_key = pr_key;
    up.run_ord = run_ord;
    up.run_id = run_id;
    up.run_cnt = run_cnt;
    up.have_probs = have_probs;
    up.next_p = next_p;
    up.want = want;
    up.nscan = nscan;
    up.radius = radius;
    up.run_tau = run_tau;
    up.n_active = n_active;
    // one host-mapped word per round (rounds <= M + 2 <= QK_MAX_NPROBE + 2)
    if (!ctx->aps_flags) {
        QK_HIP(hipHostMalloc((void **)&ctx->aps_flags, (size_t)(QK_MAX_NPROBE + 8) * 4, hipHostMallocMapped));
        QK_HIP(hipHostGetDevicePointer((void **)&ctx->aps_flags_dev, ctx->aps_flags, 0));
    }
    QK_HIP(hipMemsetAsync(n_active, 0, 8, st));
    up.out_ids = d_out_ids;
    up.out_dist = d_out_dist;
    up.sqrt_l2 = ctx->squared_l2 ? 0 : 1;
    const size_t lds_up = (size_t)2 * k * 12 + (size_t)M * 8 + 64 + (size_t)2 * ((CH + 63) / 64 + 1) * 8;
    const int maxch_u = 2 * k <= 64 ? 1 : 2 * k <= 128 ? 2 : 2 * k <= 256 ? 4 : 2 * k <= 512 ? 8 : 16;
    {  // (once per call, not per round)
        switch (maxch_u) {
            case 1: QK_HIP(hipFuncSetAttribute((const void *)k_aps_update<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_up)); break;
            case 2: QK_HIP(hipFuncSetAttribute((const void *)k_aps_update<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_up)); break;
            case 4: QK_HIP(hipFuncSetAttribute((const void *)k_aps_update<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_up)); break;
            case 8: QK_HIP(hipFuncSetAttribute((const void *)k_aps_update<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_up)); break;
            default: QK_HIP(hipFuncSetAttribute((const void *)k_aps_update<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_up)); break;
        }
    }
    int rounds = 0;
    int64_t pairs_scanned = 0;
    for (;;) {
        // the first round's rows are as long as its schedule (every query takes `first` partitions): its pair arrays, grouping and
        // per-pair merges are those of a FIRST-probe search, not of CH mostly empty slots per query
        const int CHr = rounds == 0 ? std::min(aps_first, CH) : CH;
        up.CHr = CHr;
        hipLaunchKernelGGL(k_aps_round_pids, dim3((unsigned)(((size_t)Q * CHr + 255) / 256)), dim3(256), 0, st, pids, next_p, want, Q, M,
                           CHr, round_pids);
        qk_aps_round rd;
        rd.x = dx_;
        rd.xq4 = xq4;
        rd.xn = xn;
        rd.Q = Q;
        rd.round_pids = round_pids;
        rd.CH = CHr;
        rd.CH_max = CH;
        rd.k = k;
        rd.metric = metric;
        rd.pr_ids = pr_ids;
        rd.pr_key = pr_key;
        rd.run_tau = run_tau;
        rd.round = rounds;
        QK_TRY(scan(rd));
        QK_HIP(hipSetDevice(ctx->device));
        volatile int32_t *flag = ctx->aps_flags + rounds;
        *flag = 0;
        up.host_flag = ctx->aps_flags_dev + rounds;
#define QK_UP(MC) hipLaunchKernelGGL((k_aps_update<MC>), dim3((unsigned)Q), dim3(64), lds_up, st, up);
        switch (maxch_u) {
            case 1: QK_UP(1) break;
            case 2: QK_UP(2) break;
            case 4: QK_UP(4) break;
            case 8: QK_UP(8) break;
            default: QK_UP(16) break;
        }
#undef QK_UP
        QK_HIP(hipGetLastError());
        // the round's verdict arrives through host-mapped memory: the stream is not synchronised, the next round is enqueued the
        // moment the count is visible (a device that has stopped answering shows up as an error of the query below)
        int32_t got = 0;
        for (long long spin = 0; (got = *flag) == 0; spin++) {
            __builtin_ia32_pause();  // (a round is 0.1-1 ms of device work: the waiting core stays off its sibling's issue slots)
            if ((spin & 0x3FFF) == 0x3FFF) {
                const hipError_t e = hipStreamQuery(st);
                if (e == hipSuccess) {  // everything enqueued has run: the flag is final
                    got = *flag;
                    if (got == 0) QK_FAIL(QK_ERR_HIP, "qk_search_aps: a round finished without reporting");
                    break;
                }
                if (e != hipErrorNotReady) QK_FAIL(QK_ERR_HIP, "qk_search_aps: %s", hipGetErrorString(e));
            }
        }
        rounds++;
        const int32_t left = got - 1;
        if (left <= 0) break;
        if (rounds > M + 2) QK_FAIL(QK_ERR_HIP, "qk_search_aps: rounds did not terminate");
    }
    if (timing) QK_HIP(hipEventRecord(e1, st));
    // ---- results back ---------------------------------------------------------------------------------------------------------
    if (mem == QK_MEM_HOST) {
        if (out_ids) QK_HIP(hipMemcpyAsync(out_ids, d_out_ids, Qk * 8, hipMemcpyDeviceToHost, st));
        if (out_dist) QK_HIP(hipMemcpyAsync(out_dist, d_out_dist, Qk * 4, hipMemcpyDeviceToHost, st));
        if (out_nscanned) QK_HIP(hipMemcpyAsync(out_nscanned, nscan, (size_t)Q * 4, hipMemcpyDeviceToHost, st));
        QK_HIP(hipStreamSynchronize(st));
    } else if (out_nscanned) {
        QK_HIP(hipMemcpyAsync(out_nscanned, nscan, (size_t)Q * 4, hipMemcpyDeviceToDevice, st));
    }
    if (timing) {
        QK_HIP(hipStreamSynchronize(st));
        float ms = 0.f;
        QK_HIP(hipEventElapsedTime(&ms, e0, e1));
        timing->total_ms = ms;
        timing->n_items = rounds;
    }
    (void)pairs_scanned;
    guard.drain = false;
    return QK_OK;
}

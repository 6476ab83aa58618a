// qk_dense.hip -- dense form of the scan: every query of the batch against ONE list.
//
// This is the coarse step of QueryCoordinator::search (src/cpp/src/query_coordinator.cpp:628-644: the parent is a flat
// QuakeIndex whose single partition holds the centroids; batched_scan_list(x, centroids, ..., k = nprobe),
// src/cpp/include/list_scanning.h:313-366) and the flat-index search (query_coordinator.cpp:624-626).
// FAISS does this as sgemm + norm fix-up + heap; here:
//   k_dense_ord<DB,NQ>  Q x n distance keys on v_mfma_f32_16x16x4_f32 (same canonical fmaf chain as k_scan),
//                       NQ*16 queries staged in LDS per workgroup, list rows streamed as A operands   -> MFMA-bound
//   k_select_rows<MAXCH> one wave per query: threshold-filter + rank-compaction top-k over its key row -> L2/HBM-bound
#include "qk_internal.h"

#include <cstring>
#include "qk_device.h"

#include <algorithm>
#include <cstring>

struct DenseParams {
    const float4 *vecs;  // arena
    const float *norms;
    int64_t row_off;     // first arena row of the list (multiple of 16)
    int nrows;
    int nblk;
    const float4 *xq4;   // [Q][nblk][4] fragment-ordered queries
    const float *xn;     // [Q]
    int64_t Q;
    int d;
    int metric;
    uint32_t *D;         // [Q][ld] keys
    int64_t ld;          // nrows rounded up to 16
    int tiles_per_wg;    // row tiles per workgroup (split over its 4 waves)
};

// (L2 is a template parameter: with the metric as a runtime flag hipcc kept one uniform branch per result element in the
//  epilogue -- 32 scalar branches per 16-row tile, each waiting on the VALU -- instead of unswitching the loop)
template <int DB, int NQ, bool L2>
__global__ __launch_bounds__(256) void k_dense_ord(DenseParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int nblk = P.nblk;
    constexpr bool l2 = L2;
    float4 *qs = (float4 *)smem;                               // [NQ][nblk*64]
    float *xn_s = (float *)(smem + (size_t)NQ * nblk * 1024);  // [NQ*16]
    const int64_t q_base = (int64_t)blockIdx.x * (NQ * 16);

    for (int t = wave; t < NQ * nblk; t += 4) {
        const int nq = t / nblk, cb = t - nq * nblk;
        const int64_t row = q_base + nq * 16 + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < P.Q) v = P.xq4[(row * nblk + cb) * 4 + g];
        qs[(size_t)nq * nblk * 64 + cb * 64 + lane] = v;
    }
    if (tid < NQ * 16) {
        const int64_t row = q_base + tid;
        xn_s[tid] = (row < P.Q && l2) ? P.xn[row] : 0.0f;
    }
    __syncthreads();
    float xnj[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; nq++) xnj[nq] = xn_s[nq * 16 + j];

    const int ntile_all = (P.nrows + 15) >> 4;
    const int wg_t0 = blockIdx.y * P.tiles_per_wg;
    const int wg_t1 = min(ntile_all, wg_t0 + P.tiles_per_wg);
    const int tpw = (wg_t1 - wg_t0 + 3) >> 2;
    const int t0 = wg_t0 + wave * tpw, t1 = min(wg_t1, t0 + tpw);
    const int ncd = nblk / DB;
    if (t1 <= t0) return;
    const int64_t tile_abs0 = (P.row_off >> 4) + t0;
    const float4 *src = P.vecs + tile_abs0 * nblk * 64 + lane;
    const float4 *nsrc = (const float4 *)(P.norms + (tile_abs0 << 4)) + g;
    const int nsteps = (t1 - t0) * ncd;
    float4 a0[DB], a1[DB];
    float4 yn_cur = make_float4(0.f, 0.f, 0.f, 0.f), yn_next = yn_cur;
    f32x4 acc[NQ];
    int dch = 0, tile = t0, ldch = 0, ltile = 0;

#define DN_LOAD(A, S)                                                 \
    {                                                                 \
        const float4 *pp_ = src + (int64_t)(S) * (DB * 64);           \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) A[b_] = pp_[b_ * 64]; \
        if (ldch == 0) {                                              \
            if (l2) yn_next = nsrc[(int64_t)ltile * 4];               \
            ltile++;                                                  \
        }                                                             \
        if (++ldch == ncd) ldch = 0;                                  \
    }

#define DN_STEP(A)                                                                                           \
    {                                                                                                        \
        if (dch == 0) {                                                                                      \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++) acc[nq_] = (f32x4){0.f, 0.f, 0.f, 0.f};     \
        }                                                                                                    \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) {                                                  \
            /* operands of the NQ query tiles first, then each k-step across the NQ independent accumulators */ \
            /* (an accumulator still sees its own k-steps in order: same bits)                                */ \
            float4 bq_[NQ];                                                                                  \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                bq_[nq_] = qs[(size_t)nq_ * nblk * 64 + (dch * DB + b_) * 64 + lane];                        \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].x, bq_[nq_].x, acc[nq_], 0, 0, 0);     \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].y, bq_[nq_].y, acc[nq_], 0, 0, 0);     \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].z, bq_[nq_].z, acc[nq_], 0, 0, 0);     \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].w, bq_[nq_].w, acc[nq_], 0, 0, 0);     \
        }                                                                                                    \
        if (++dch == ncd) {                                                                                  \
            dch = 0;                                                                                         \
            const float yv_[4] = {yn_cur.x, yn_cur.y, yn_cur.z, yn_cur.w};                                   \
            const int row0_ = (tile << 4) + 4 * g;                                                           \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++) {                                           \
                const int64_t q_ = q_base + nq_ * 16 + j;                                                    \
                uint4 o_;                                                                                    \
                uint32_t *op_ = (uint32_t *)&o_;                                                             \
                _Pragma("unroll") for (int reg_ = 0; reg_ < 4; reg_++) {                                     \
                    const float v_ = acc[nq_][reg_];                                                         \
                    const uint32_t k_ = l2 ? ord_from_l2(l2_expanded(xnj[nq_], yv_[reg_], v_)) : ord_from_ip(v_); \
                    op_[reg_] = (row0_ + reg_ < P.nrows) ? k_ : 0xFFFFFFFFu;                                 \
                }                                                                                            \
                if (q_ < P.Q) *(uint4 *)(P.D + q_ * P.ld + row0_) = o_;                                      \
            }                                                                                                \
            yn_cur = yn_next;                                                                                \
            tile++;                                                                                          \
        }                                                                                                    \
    }

    DN_LOAD(a0, 0);
    yn_cur = yn_next;
    int s = 0;
    while (s < nsteps) {
        if (s + 1 < nsteps) DN_LOAD(a1, s + 1);
        DN_STEP(a0);
        s++;
        if (s >= nsteps) break;
        if (s + 1 < nsteps) DN_LOAD(a0, s + 1);
        DN_STEP(a1);
        s++;
    }
#undef DN_LOAD
#undef DN_STEP
}

struct SelectParams {
    const uint32_t *D;
    int64_t ld;
    int nrows;
    const int64_t *ids;  // arena ids + row_off
    int k;
    int Cm;
    int metric;
    int sqrt_l2;
    int64_t *out_ids;
    float *out_dist;
    // long rows: every key row is cut into `slices` slices of slice_len keys (a multiple of 1024), one wave each; the wave
    // leaves the k best of its slice as (id, key) in out_ids / out_ord [Q][slices][k] and k_merge_slices finishes the query
    int slices;
    int slice_len;
    uint32_t *out_ord;
};

template <int MAXCH>
__global__ __launch_bounds__(64) void k_select_rows(SelectParams S0) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    SelectParams S = S0;
    int64_t q = blockIdx.x;
    const uint32_t *row;
    if (S.slices > 1) {  // this wave's slice of query qq's row
        const int64_t qq = q / S.slices;
        const int sl = (int)(q - qq * S.slices);
        const int r0 = sl * S.slice_len;
        row = S.D + qq * S.ld + r0;
        S.ids += r0;
        S.nrows = max(0, min(S.slice_len, S.nrows - r0));
    } else {
        row = S.D + q * S.ld;
    }
    const int k = S.k, Cm = S.Cm;
    // every wave starts its walk at a different 4 KB block of its keys (the walk order does not matter to a selection): rows and
    // slices start at multiples of 16 KB, and thousands of waves stepping through them in phase hit the same few memory channels
    const int niter = (S.nrows + 1023) >> 10;
    const int rot = niter > 1 ? (int)(blockIdx.x % (unsigned)niter) : 0;
    int64_t *pool_id = (int64_t *)smem;
    uint32_t *pool_ord = (uint32_t *)(smem + (size_t)Cm * 8);
    uint32_t tau = 0xFFFFFFFFu;
    int cnt = 0;
    // Pool entries [n_ids, cnt) hold the ROW of a candidate, not yet its id: an id load inside the append path is a dependent
    // memory round trip per passing key; the ids of all new entries are fetched together right before the pool is selected
    // (the (key, id) order needs them there), one round trip per selection.
    int n_ids = 0;
    auto to_ids = [&](int n) {
        for (int e = n_ids + lane; e < n; e += 64) pool_id[e] = S.ids[pool_id[e]];
        n_ids = n;
    };
    // pass 1: a bound without any top-k bookkeeping.  Every lane keeps the m = ceil(k/64) smallest keys it sees; the k-th
    // smallest of those 64*m values has at least k keys <= it, so it bounds the k-th best overall.  Without it the first keys
    // all pass (tau = inf) and the pool is compacted over and over (k = 81: 199 us per launch instead of ~15).
    if (k <= 512) {  // (this kernel serves k up to 960; beyond 8 values per lane it starts without a bound)
        const int m = (k + 63) >> 6;  // 1..8
        uint32_t lm[8];
#pragma unroll
        for (int t = 0; t < 8; t++) lm[t] = 0xFFFFFFFFu;
        for (int it = 0; it < niter; it++) {
            const int base = ((it + rot) % niter) * 1024;
            uint4 kv4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r0 = base + u * 256 + lane * 4;
                kv4[u] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                if (r0 < S.nrows) kv4[u] = *(const uint4 *)(row + r0);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t kk4[4] = {kv4[u].x, kv4[u].y, kv4[u].z, kv4[u].w};
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++) {
                    uint32_t v = kk4[c4];
                    if (m == 1) {
                        lm[0] = min(lm[0], v);
                    } else {
#pragma unroll
                        for (int t = 0; t < 8; t++) {  // sorted insertion into lm[0..m)
                            if (t < m) {
                                const uint32_t lo = min(lm[t], v);
                                v = max(lm[t], v);
                                lm[t] = lo;
                            }
                        }
                    }
                }
            }
        }
        // rank of every kept value among the 64*m under (value, slot, lane); the one with rank k-1 is the bound
        uint32_t bound = 0xFFFFFFFFu;
        for (int t = 0; t < m; t++) {
            const uint32_t mine = t == 0 ? lm[0] : t == 1 ? lm[1] : t == 2 ? lm[2] : t == 3 ? lm[3] : t == 4 ? lm[4] : t == 5 ? lm[5] : t == 6 ? lm[6] : lm[7];
            int rk = 0;
            for (int tt = 0; tt < m; tt++) {
                const uint32_t theirs = tt == 0 ? lm[0] : tt == 1 ? lm[1] : tt == 2 ? lm[2] : tt == 3 ? lm[3] : tt == 4 ? lm[4] : tt == 5 ? lm[5] : tt == 6 ? lm[6] : lm[7];
                for (int l = 0; l < 64; l++) {
                    const uint32_t ot = __builtin_amdgcn_readlane(theirs, l);
                    rk += (ot < mine || (ot == mine && (tt < t || (tt == t && l < lane)))) ? 1 : 0;
                }
            }
            const uint64_t mk = __ballot(rk == k - 1);
            if (mk) bound = __builtin_amdgcn_readlane(mine, __ffsll((unsigned long long)mk) - 1);
        }
        tau = bound;
    }
    // 4 keys per lane per load (16 B), 4 loads in flight: 1024 rows per wave step
    for (int it = 0; it < niter; it++) {
        const int base = ((it + rot) % niter) * 1024;
        uint4 kv4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r0 = base + u * 256 + lane * 4;
            kv4[u] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (r0 < S.nrows) kv4[u] = *(const uint4 *)(row + r0);  // ld is a multiple of 16, tail keys are 0xFFFFFFFF
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r0 = base + u * 256 + lane * 4;
            const uint32_t kk[4] = {kv4[u].x, kv4[u].y, kv4[u].z, kv4[u].w};
            // one ballot for the 4 keys of the step; the per-key work only runs when something passes
            const bool any = r0 < S.nrows && (min(min(kk[0], kk[1]), min(kk[2], kk[3])) <= tau) &&
                             (min(min(kk[0], kk[1]), min(kk[2], kk[3])) != 0xFFFFFFFFu);
            if (!__ballot(any)) continue;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int r = r0 + t;
                const bool pass = r < S.nrows && kk[t] <= tau && kk[t] != 0xFFFFFFFFu;
                const uint64_t m = __ballot(pass);
                if (m) {
                    if (pass) {
                        const int sl = cnt + __popcll(m & ((1ull << lane) - 1ull));
                        pool_ord[sl] = kk[t];
                        pool_id[sl] = r;  // the ROW for now: its id is fetched when the pool is next selected (see to_ids)
                    }
                    cnt += __popcll(m);
                    if (cnt > Cm - 64) {  // (unsorted k best + their bound; the final compaction sorts)
                        to_ids(cnt);
                        uint32_t kth;
                        cnt = select_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane, kth);
                        n_ids = cnt;
                        if (cnt >= k) tau = min(tau, kth);
                    }
                }
            }
        }
    }
    to_ids(cnt);
    cnt = compact_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane);
    if (S.slices > 1) {
        for (int e = lane; e < k; e += 64) {
            S.out_ids[q * k + e] = e < cnt ? pool_id[e] : -1;
            S.out_ord[q * k + e] = e < cnt ? pool_ord[e] : 0xFFFFFFFFu;
        }
        return;
    }
    for (int e = lane; e < k; e += 64) {
        int64_t oid = -1;
        float od = S.metric == QK_METRIC_IP ? -INFINITY : INFINITY;
        if (e < cnt) {
            oid = pool_id[e];
            const uint32_t o = pool_ord[e];
            if (S.metric == QK_METRIC_L2) {
                const float d2 = __uint_as_float(o);
                od = S.sqrt_l2 ? sqrtf(d2) : d2;
            } else {
                od = ip_from_ord(o);
            }
        }
        S.out_ids[q * k + e] = oid;
        if (S.out_dist) S.out_dist[q * k + e] = od;
    }
}


// The slices of a query (k_select_rows with slices > 1): S x k candidates, each slice sorted under (key, id) -> the k best overall,
// sorted, as ids + distances.  One wave per query; S * k <= 1024.
template <int MAXCH>
__global__ __launch_bounds__(64) void k_merge_slices(const int64_t *in_ids, const uint32_t *in_ord, int slices, int k, int metric, int sqrt_l2,
                                                     int64_t *out_ids, float *out_dist) {
    __shared__ int64_t pool_id[MAXCH * 64];
    __shared__ uint32_t pool_ord[MAXCH * 64];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int n_in = slices * k;  // <= MAXCH * 64
    // live entries first (a slice shorter than k is padded with 0xFFFFFFFF keys): ballot-compacted copy
    int cnt = 0;
    // (all loads of the wave in flight before the first ballot: a load per round of the loop below was a memory round trip each)
    uint32_t ov[MAXCH];
    int64_t iv[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; c++) {
        const int e = c * 64 + lane;
        ov[c] = e < n_in ? in_ord[q * n_in + e] : 0xFFFFFFFFu;
        iv[c] = e < n_in ? in_ids[q * n_in + e] : -1;
    }
    // the k-th smallest key of all candidates, on the registers (bisection on the key bits); when exactly k candidates lie at or
    // under it -- always, unless equal keys straddle the k-th place -- only those go to the pool and one wave-wide rank sort orders
    // them (256 candidates, k = 32: 12.4 -> ~5 us against the pool-wide select + sort, which stays for the tie case and for n < k)
    uint32_t T = 0;
    for (int b = 31; b >= 0; b--) {
        const uint32_t tr = T | (1u << b);
        int c = 0;
#pragma unroll
        for (int i = 0; i < MAXCH; i++) c += __popcll(__ballot(ov[i] < tr));
        if (c < k) T = tr;
    }
    int c_le = 0;
#pragma unroll
    for (int i = 0; i < MAXCH; i++) c_le += __popcll(__ballot(ov[i] <= T && ov[i] != 0xFFFFFFFFu));
    const bool exact_cut = c_le <= k && k <= 64;  // (fewer than k live candidates: T = 0xFFFFFFFF, all of them pass)
#pragma unroll
    for (int c = 0; c < MAXCH; c++) {
        const bool live = ov[c] != 0xFFFFFFFFu && (!exact_cut || ov[c] <= T);  // (padding of a slice shorter than k: never a live key)
        const uint64_t m = __ballot(live);
        if (live) {
            const int sl = cnt + __popcll(m & ((1ull << lane) - 1ull));
            pool_ord[sl] = ov[c];
            pool_id[sl] = iv[c];
        }
        cnt += __popcll(m);
    }
    if (exact_cut) cnt = compact_pool<1>(pool_ord, pool_id, cnt, k, lane);
    else
    cnt = compact_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane);
    for (int e = lane; e < k; e += 64) {
        int64_t oid = -1;
        float od = metric == QK_METRIC_IP ? -INFINITY : INFINITY;
        if (e < cnt) {
            oid = pool_id[e];
            const uint32_t o = pool_ord[e];
            if (metric == QK_METRIC_L2) {
                const float d2 = __uint_as_float(o);
                od = sqrt_l2 ? sqrtf(d2) : d2;
            } else {
                od = ip_from_ord(o);
            }
        }
        out_ids[q * k + e] = oid;
        if (out_dist) out_dist[q * k + e] = od;
    }
}

// ---- large-k selection (448 < k <= QK_MAX_NPROBE): the coarse step with nprobe / APS candidate counts beyond the LDS pools ----
// One workgroup of 256 threads per query over its key row (L2-resident):
//   1. T = k-th smallest key by bisection on the 32 key bits (one counting pass over the row per bit);
//   2. if more rows tie on T than are needed, I = the id that cuts them, by bisection on the id bits;
//   3. the k survivors ((key, id) < (T, I]) are gathered into LDS and bitonic-sorted under the total order (key, id).
// Exact under the same (key, id) order as every other selection in the library; cost ~ (32 + 63) passes over 4*n bytes.
struct SelectLargeParams {
    const uint32_t *D;   // [Q][ld] keys
    int64_t ld;
    int nrows;
    const int64_t *ids;  // arena ids + row_off
    int k;               // <= QK_MAX_NPROBE
    int kp;              // next power of two >= k
    int metric;
    int sqrt_l2;
    int64_t *out_ids;
    float *out_dist;
};

// key / id source of one query for select_large
struct RowSrc {  // one list: keys in a row of the key matrix, ids in arena order
    const uint32_t *row;
    const int64_t *ids;
    int n;
    __device__ __forceinline__ uint32_t key(int i) const { return row[i]; }
    __device__ __forceinline__ int64_t id(int i) const { return ids[i]; }
};
struct PairSrc {  // several lists: the query's keys are the concatenation of its P lists (qk_widek_device)
    const uint32_t *row;       // keys + pair_base[q*P]
    const int64_t *pbase;      // pair_base + q*P  (P + 1 entries, absolute)
    const int64_t *pids;       // pids + q*P, or nullptr (pair r -> list r)
    const int64_t *pt_off;
    const int64_t *ids;        // arena ids
    int P, n;
    __device__ __forceinline__ uint32_t key(int i) const { return row[i]; }
    __device__ __forceinline__ int64_t id(int i) const {
        const int64_t pos = pbase[0] + i;
        int lo = 0, hi = P;  // pbase[lo] <= pos < pbase[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pbase[mid] <= pos) lo = mid; else hi = mid;
        }
        const int64_t pid = pids ? pids[lo] : lo;
        return ids[pt_off[pid] + (pos - pbase[lo])];
    }
};

__device__ __forceinline__ int block_sum_256(int v, int *s_red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

template <class Src>
__device__ __forceinline__ void select_large(const Src &src, int k, int kp, int metric, int sqrt_l2, int64_t *out_ids, float *out_dist,
                                             unsigned char *smem, int *s_red, int *s_cnt) {
    const int tid = threadIdx.x;
    const int n = src.n;
    uint32_t *e_ord = (uint32_t *)smem;                     // [kp]
    int64_t *e_id = (int64_t *)(smem + (size_t)kp * 4);     // [kp]
    const int kk = min(k, n);  // fewer rows than k: everything is kept, the rest is padding
    // 1. k-th smallest key
    uint32_t T = 0;
    for (int b = 31; b >= 0; b--) {
        const uint32_t tr = T | (1u << b);
        int c = 0;
        for (int i = tid; i < n; i += 256) c += src.key(i) < tr ? 1 : 0;
        if (block_sum_256(c, s_red) < kk) T = tr;
    }
    int c_lt = 0, c_eq = 0;
    for (int i = tid; i < n; i += 256) {
        const uint32_t v = src.key(i);
        c_lt += v < T ? 1 : 0;
        c_eq += v == T ? 1 : 0;
    }
    c_lt = block_sum_256(c_lt, s_red);
    c_eq = block_sum_256(c_eq, s_red);
    const int need = kk - c_lt;  // ties on T to take, 1 <= need <= c_eq (0 when kk == 0)
    // 2. id cut among the ties (ids are unique and non-negative)
    int64_t I = INT64_MAX;
    if (need < c_eq) {
        uint64_t Iu = 0;
        for (int b = 62; b >= 0; b--) {
            const uint64_t tr = Iu | (1ull << b);
            int c = 0;
            for (int i = tid; i < n; i += 256) c += (src.key(i) == T && (uint64_t)src.id(i) < tr) ? 1 : 0;
            if (block_sum_256(c, s_red) < need) Iu = tr;
        }
        I = (int64_t)Iu;  // the need-th smallest tied id
    }
    // 3. gather + sort
    if (tid == 0) *s_cnt = 0;
    for (int i = tid; i < kp; i += 256) {
        e_ord[i] = 0xFFFFFFFFu;
        e_id[i] = INT64_MAX;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const uint32_t v = src.key(i);
        if (v < T || (v == T && need > 0)) {
            const int64_t vid = src.id(i);
            if (v < T || vid <= I) {
                const int sl = atomicAdd(s_cnt, 1);
                if (sl < kp) {
                    e_ord[sl] = v;
                    e_id[sl] = vid;
                }
            }
        }
    }
    __syncthreads();
    for (int size = 2; size <= kp; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (kp >> 1); i += 256) {
                const int lo = 2 * i - (i & (stride - 1));  // index of the lower element of the pair
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t a = e_ord[lo], b = e_ord[hi];
                const int64_t ia = e_id[lo], ib = e_id[hi];
                const bool gt = a > b || (a == b && ia > ib);
                if (gt == up) {
                    e_ord[lo] = b;
                    e_ord[hi] = a;
                    e_id[lo] = ib;
                    e_id[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    for (int e = tid; e < k; e += 256) {
        int64_t oid = -1;
        float od = metric == QK_METRIC_IP ? -INFINITY : INFINITY;
        if (e < kk) {
            oid = e_id[e];
            const uint32_t o = e_ord[e];
            if (metric == QK_METRIC_L2) {
                const float d2 = __uint_as_float(o);
                od = sqrt_l2 ? sqrtf(d2) : d2;
            } else {
                od = ip_from_ord(o);
            }
        }
        out_ids[e] = oid;
        if (out_dist) out_dist[e] = od;
    }
}

__global__ __launch_bounds__(256) void k_select_rows_large(SelectLargeParams S) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_red[4];
    __shared__ int s_cnt;
    const int64_t q = blockIdx.x;
    RowSrc src;
    src.row = S.D + q * S.ld;
    src.ids = S.ids;
    src.n = S.nrows;
    select_large(src, S.k, S.kp, S.metric, S.sqrt_l2, S.out_ids + q * S.k, S.out_dist ? S.out_dist + q * S.k : nullptr, smem, s_red, &s_cnt);
}

// ---- wide k over several lists (QK_MAX_K < k <= QK_MAX_WIDE_K) ------------------------------------------------------------
// qk_scan_device in key-emission mode writes every (pair, row) key; here: list sizes per pair, their prefix sums, and the
// exact selection over each query's concatenated key segments.
struct WideKParams {
    const uint32_t *keys;
    const int64_t *pair_base;  // [npairs + 1]
    const int64_t *pids;       // [Q][P] or nullptr
    const int64_t *pt_off;
    const int64_t *ids;
    int P, k, kp, metric, sqrt_l2;
    int64_t *out_ids;
    float *out_dist;
};

__global__ void k_pair_sizes(const int64_t *pids, int64_t npairs, int P, const int32_t *pt_size, int npids, int64_t *sizes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > npairs) return;
    int64_t sz = 0;
    if (i < npairs) {
        const int64_t p = pids ? pids[i] : (i % P);
        if (p >= 0 && p < npids && pt_size[p] > 0) sz = pt_size[p];
    }
    sizes[i] = sz;  // sizes[npairs] = 0: the exclusive scan then ends with the total
}

__global__ __launch_bounds__(256) void k_select_pairs_large(WideKParams W) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_red[4];
    __shared__ int s_cnt;
    const int64_t q = blockIdx.x;
    PairSrc src;
    src.pbase = W.pair_base + q * W.P;
    src.row = W.keys + src.pbase[0];
    src.pids = W.pids ? W.pids + q * W.P : nullptr;
    src.pt_off = W.pt_off;
    src.ids = W.ids;
    src.P = W.P;
    src.n = (int)(src.pbase[W.P] - src.pbase[0]);
    select_large(src, W.k, W.kp, W.metric, W.sqrt_l2, W.out_ids + q * W.k, W.out_dist ? W.out_dist + q * W.k : nullptr, smem, s_red, &s_cnt);
}

// ---- fused argmin: the coarse step with nprobe = 1 (and PartitionManager::add's k = 1 parent search) -------------------------
// Same decomposition as k_dense_ord, but nothing is materialised: every lane keeps the best (key, row) of the rows it sees for
// each of its NQ queries, the workgroup reduces them and folds its candidate into best64[q] = (key << 32 | id) with one
// atomicMin per query -- the (key, id) total order as an integer order (ids < 2^32, checked by the host).  Two rows with the
// same key are ordered by id right away (the two ids are loaded then: an exact tie is rare).
struct ArgminParams {
    const float4 *vecs;
    const float *norms;
    const int64_t *ids;  // arena ids + row_off
    int64_t row_off;
    int nrows;
    int nblk;
    const float4 *xq4;
    const float *xn;
    int64_t Q;
    int metric;
    int tiles_per_wg;
    unsigned long long *best64;  // [Q], preset to ~0
    // FUSE: the query preparation of the batch (k_prep_queries, qk_scan.hip) done here, while the queries are staged
    const float *x;              // [Q][d] raw queries, 16-byte aligned, d % 4 == 0
    int d;
    float4 *xq4_out;             // fragment-ordered copy, written by the workgroups of row chunk 0
    float *xn_out;               // canonical squared norms, likewise
    float4 *xp4_out;             // row-major copy padded to 16 columns, likewise
    uint4 *zero16;               // region the scan of this batch wants cleared (everybody clears a share)
    int64_t n_zero16;
    unsigned long long *best64_next;  // the NEXT batch's "nothing yet" array (never this one: it is being written)
    int64_t n_next;
};

template <int DB, int NQ, bool L2, bool FUSE = false>
__global__ __launch_bounds__(256) void k_dense_argmin(ArgminParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int nblk = P.nblk;
    constexpr bool l2 = L2;
    float4 *qs = (float4 *)smem;                               // [NQ][nblk*64]
    float *xn_s = (float *)(smem + (size_t)NQ * nblk * 1024);  // [NQ*16]
    unsigned long long *red = (unsigned long long *)(xn_s + NQ * 16);  // [4][NQ*16]
    const int64_t q_base = (int64_t)blockIdx.x * (NQ * 16);

    if (FUSE) {
        // ---- the batch's query preparation, folded into the staging ------------------------------------------------------
        const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x, nwg = (int64_t)gridDim.x * gridDim.y;
        for (int64_t i = wg * 256 + tid; i < P.n_zero16; i += nwg * 256) P.zero16[i] = make_uint4(0u, 0u, 0u, 0u);
        for (int64_t i = wg * 256 + tid; i < P.n_next; i += nwg * 256) P.best64_next[i] = ~0ull;
        const bool writer = blockIdx.y == 0;  // one workgroup per query block leaves the prepared copies behind
        const int d4 = P.d >> 2, p4 = nblk * 4;  // float4 per raw row / per padded row
        float *qf = (float *)qs;
        // the NQ*16 rows of this block are contiguous in x: coalesced float4 loads, every component to its fragment slot
        // (row r of query tile nq, column c = 16 cb + 4 t' + g'  ->  float ((nq*nblk + cb)*64 + g'*16 + r)*4 + t', see qk_internal.h)
        constexpr int U = 8;  // loads in flight per thread (d = 128, NQ = 4: the whole block in one round trip)
        const int total4 = NQ * 16 * p4;
        for (int i0 = 0; i0 < total4; i0 += 256 * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + u * 256 + tid;
                const int r_all = i / p4, c4 = i - r_all * p4;
                const int64_t row = q_base + r_all;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < total4 && row < P.Q && c4 < d4) v[u] = ((const float4 *)P.x)[row * d4 + c4];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + u * 256 + tid;
                if (i < total4) {
                    const int r_all = i / p4, c4 = i - r_all * p4;
                    const int64_t row = q_base + r_all;
                    if (writer && row < P.Q) P.xp4_out[row * p4 + c4] = v[u];
                    const int nq = r_all >> 4, r = r_all & 15, cb = c4 >> 2, tt = c4 & 3;
                    float *dst = qf + ((size_t)(nq * nblk + cb) * 64 + r) * 4 + tt;  // g' = 0; + 64 floats per g'
                    dst[0] = v[u].x;
                    dst[64] = v[u].y;
                    dst[128] = v[u].z;
                    dst[192] = v[u].w;
                }
            }
        }
        __syncthreads();
        if (tid < NQ * 16) {  // canonical squared norm: one k-ordered fmaf chain per query (k_prep_queries' arithmetic; the padded
                              // columns hold 0: fma(0, 0, acc) == acc)
            const int nq = tid >> 4, r = tid & 15;
            const float *base = qf + ((size_t)nq * nblk * 64 + r) * 4;
            float acc = 0.0f;
            for (int cb = 0; cb < nblk; cb++) {
                float e[16];
#pragma unroll
                for (int w = 0; w < 16; w++) e[w] = base[(size_t)cb * 256 + (w & 3) * 64 + (w >> 2)];
#pragma unroll
                for (int w = 0; w < 16; w++) acc = __fmaf_rn(e[w], e[w], acc);
            }
            const int64_t row = q_base + tid;
            xn_s[tid] = (row < P.Q && l2) ? acc : 0.0f;
            if (writer && row < P.Q) P.xn_out[row] = acc;
        }
        if (writer) {
            for (int t = wave; t < NQ * nblk; t += 4) {
                const int nq = t / nblk, cb = t - nq * nblk;
                const int64_t row = q_base + nq * 16 + j;
                if (row < P.Q) P.xq4_out[(row * nblk + cb) * 4 + g] = qs[(size_t)nq * nblk * 64 + cb * 64 + lane];
            }
        }
    } else {
    for (int t = wave; t < NQ * nblk; t += 4) {
        const int nq = t / nblk, cb = t - nq * nblk;
        const int64_t row = q_base + nq * 16 + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < P.Q) v = P.xq4[(row * nblk + cb) * 4 + g];
        qs[(size_t)nq * nblk * 64 + cb * 64 + lane] = v;
    }
    if (tid < NQ * 16) {
        const int64_t row = q_base + tid;
        xn_s[tid] = (row < P.Q && l2) ? P.xn[row] : 0.0f;
    }
    }
    __syncthreads();
    // running minimum per (lane, query) of key << 32 | id -- the (key, id) order as ONE integer order (ids < 2^32 here), so
    // the update is a 64-bit compare and two selects; the ids of a lane's 4 rows travel with the tile like the norms do
    float xnj[NQ];
    unsigned long long best[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; nq++) {
        xnj[nq] = xn_s[nq * 16 + j];
        best[nq] = ~0ull;
    }
    const int ntile_all = (P.nrows + 15) >> 4;
    const int wg_t0 = blockIdx.y * P.tiles_per_wg;
    const int wg_t1 = min(ntile_all, wg_t0 + P.tiles_per_wg);
    const int tpw = (wg_t1 - wg_t0 + 3) >> 2;
    const int t0 = wg_t0 + wave * tpw, t1 = min(wg_t1, t0 + tpw);
    const int ncd = nblk / DB;
    if (t1 > t0) {
        const int64_t tile_abs0 = (P.row_off >> 4) + t0;
        const float4 *src = P.vecs + tile_abs0 * nblk * 64 + lane;
        const float4 *nsrc = (const float4 *)(P.norms + (tile_abs0 << 4)) + g;
        const longlong2 *isrc = (const longlong2 *)(P.ids + ((int64_t)t0 << 4)) + 2 * g;  // +8 longlong2 per tile (ids of the list)
        longlong2 ia_cur = {0, 0}, ib_cur = {0, 0}, ia_next = {0, 0}, ib_next = {0, 0};
        const int nsteps = (t1 - t0) * ncd;
        float4 a0[DB], a1[DB];
        float4 yn_cur = make_float4(0.f, 0.f, 0.f, 0.f), yn_next = yn_cur;
        f32x4 acc[NQ];
        int dch = 0, tile = t0, ldch = 0, ltile = 0;
#define AM_LOAD(A, S)                                                 \
    {                                                                 \
        const float4 *pp_ = src + (int64_t)(S) * (DB * 64);           \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) A[b_] = pp_[b_ * 64]; \
        if (ldch == 0) {                                              \
            if (l2) yn_next = nsrc[(int64_t)ltile * 4];               \
            ia_next = isrc[(int64_t)ltile * 8];                       \
            ib_next = isrc[(int64_t)ltile * 8 + 1];                   \
            ltile++;                                                  \
        }                                                             \
        if (++ldch == ncd) ldch = 0;                                  \
    }
#define AM_STEP(A)                                                                                           \
    {                                                                                                        \
        if (dch == 0) {                                                                                      \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++) acc[nq_] = (f32x4){0.f, 0.f, 0.f, 0.f};     \
        }                                                                                                    \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) {                                                  \
            /* operands of the NQ query tiles first, then each k-step across the NQ independent accumulators */ \
            /* (an accumulator still sees its own k-steps in order: same bits)                                */ \
            float4 bq_[NQ];                                                                                  \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                bq_[nq_] = qs[(size_t)nq_ * nblk * 64 + (dch * DB + b_) * 64 + lane];                        \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].x, bq_[nq_].x, acc[nq_], 0, 0, 0);     \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].y, bq_[nq_].y, acc[nq_], 0, 0, 0);     \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].z, bq_[nq_].z, acc[nq_], 0, 0, 0);     \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].w, bq_[nq_].w, acc[nq_], 0, 0, 0);     \
        }                                                                                                    \
        if (++dch == ncd) {                                                                                  \
            dch = 0;                                                                                         \
            const float yv_[4] = {yn_cur.x, yn_cur.y, yn_cur.z, yn_cur.w};                                   \
            const uint32_t iv_[4] = {(uint32_t)ia_cur.x, (uint32_t)ia_cur.y, (uint32_t)ib_cur.x, (uint32_t)ib_cur.y}; \
            const int row0_ = (tile << 4) + 4 * g;                                                           \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++) {                                           \
                _Pragma("unroll") for (int reg_ = 0; reg_ < 4; reg_++) {                                     \
                    const float v_ = acc[nq_][reg_];                                                         \
                    const uint32_t k_ = l2 ? ord_from_l2(l2_expanded(xnj[nq_], yv_[reg_], v_)) : ord_from_ip(v_); \
                    const unsigned long long c_ = ((unsigned long long)k_ << 32) | iv_[reg_];                \
                    const bool lt_ = (row0_ + reg_ < P.nrows) & (c_ < best[nq_]);                            \
                    best[nq_] = lt_ ? c_ : best[nq_];                                                        \
                }                                                                                            \
            }                                                                                                \
            yn_cur = yn_next;                                                                                \
            ia_cur = ia_next;                                                                                \
            ib_cur = ib_next;                                                                                \
            tile++;                                                                                          \
        }                                                                                                    \
    }
        AM_LOAD(a0, 0);
        yn_cur = yn_next;
        ia_cur = ia_next;
        ib_cur = ib_next;
        int s = 0;
        while (s < nsteps) {
            if (s + 1 < nsteps) AM_LOAD(a1, s + 1);
            AM_STEP(a0);
            s++;
            if (s >= nsteps) break;
            if (s + 1 < nsteps) AM_LOAD(a0, s + 1);
            AM_STEP(a1);
            s++;
        }
#undef AM_LOAD
#undef AM_STEP
    }
    // lane -> (key << 32 | id); min over the 4 row groups of the wave, the 4 waves, then the workgroups
#pragma unroll
    for (int nq = 0; nq < NQ; nq++) {
        unsigned long long v = best[nq];
        unsigned long long o = __shfl_xor(v, 16);
        v = o < v ? o : v;
        o = __shfl_xor(v, 32);
        v = o < v ? o : v;
        if (g == 0) red[wave * (NQ * 16) + nq * 16 + j] = v;
    }
    __syncthreads();
    if (tid < NQ * 16) {
        unsigned long long v = red[tid];
#pragma unroll
        for (int w = 1; w < 4; w++) {
            const unsigned long long o = red[w * (NQ * 16) + tid];
            v = o < v ? o : v;
        }
        const int64_t q = q_base + tid;
        if (q < P.Q && v != ~0ull) atomicMin(&P.best64[q], v);
    }
}

__global__ void k_argmin_finish(const unsigned long long *best64, int64_t Q, int metric, int sqrt_l2, int64_t *out_ids, float *out_dist) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const unsigned long long v = best64[q];
    int64_t oid = -1;
    float od = metric == QK_METRIC_IP ? -INFINITY : INFINITY;
    if (v != ~0ull) {
        oid = (int64_t)(v & 0xFFFFFFFFull);
        const uint32_t o = (uint32_t)(v >> 32);
        if (metric == QK_METRIC_L2) {
            const float d2 = __uint_as_float(o);
            od = sqrt_l2 ? sqrtf(d2) : d2;
        } else {
            od = ip_from_ord(o);
        }
    }
    out_ids[q] = oid;
    if (out_dist) out_dist[q] = od;
}

template <int DB, int NQ, bool L2>
static int launch_argmin_m(hipStream_t st, dim3 grid, size_t lds, const ArgminParams &ap) {
    if (ap.x) {  // the batch's query preparation rides along
        QK_HIP(hipFuncSetAttribute((const void *)k_dense_argmin<DB, NQ, L2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_dense_argmin<DB, NQ, L2, true>), grid, dim3(256), lds, st, ap);
        return QK_OK;
    }
    QK_HIP(hipFuncSetAttribute((const void *)k_dense_argmin<DB, NQ, L2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_dense_argmin<DB, NQ, L2>), grid, dim3(256), lds, st, ap);
    return QK_OK;
}
template <int DB, int NQ>
static int launch_argmin_t(hipStream_t st, dim3 grid, size_t lds, const ArgminParams &ap) {
    return ap.metric == QK_METRIC_L2 ? launch_argmin_m<DB, NQ, true>(st, grid, lds, ap) : launch_argmin_m<DB, NQ, false>(st, grid, lds, ap);
}

template <int DB, int NQ, bool L2>
static int launch_dense_m(hipStream_t st, dim3 grid, size_t lds, const DenseParams &dp) {
    QK_HIP(hipFuncSetAttribute((const void *)k_dense_ord<DB, NQ, L2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_dense_ord<DB, NQ, L2>), grid, dim3(256), lds, st, dp);
    return QK_OK;
}
template <int DB, int NQ>
static int launch_dense_t(hipStream_t st, dim3 grid, size_t lds, const DenseParams &dp) {
    return dp.metric == QK_METRIC_L2 ? launch_dense_m<DB, NQ, true>(st, grid, lds, dp) : launch_dense_m<DB, NQ, false>(st, grid, lds, dp);
}

// k_merge_slices over the [Q][slices][k] candidates of a sliced selection (k_select_rows with slices > 1, k_dense_fused)
int qk_launch_merge_slices(qk_ctx *ctx, const int64_t *sl_ids, const uint32_t *sl_ord, int64_t nq, int slices, int k, int metric,
                           bool sqrt_l2, int64_t *out_ids, float *out_dist) {
    hipStream_t st = ctx->stream;
    const int n_in = slices * k;
    if (n_in > 1024) QK_FAIL(QK_ERR_UNSUPPORTED, "slice merge: %d candidates per query", n_in);
#define MS_LAUNCH(M_) \
    hipLaunchKernelGGL((k_merge_slices<M_>), dim3((unsigned)nq), dim3(64), 0, st, sl_ids, sl_ord, slices, k, metric, sqrt_l2 ? 1 : 0, out_ids, out_dist)
    if (n_in <= 128) MS_LAUNCH(2);
    else if (n_in <= 256) MS_LAUNCH(4);
    else if (n_in <= 512) MS_LAUNCH(8);
    else MS_LAUNCH(16);
#undef MS_LAUNCH
    QK_HIP(hipGetLastError());
    return QK_OK;
}

int qk_dense_device(qk_ctx *ctx, qk_store *s, int64_t list_no, const qk_scan_args &a, qk_timing *timing, int ev_base) {
    ctx->last_scan_kernel = "k_dense";
    const int64_t Q = a.Q;
    const int k = a.k;
    const qk_part &pt = s->parts[list_no];
    const int nrows = (int)pt.size;
    hipStream_t st = ctx->stream;
    const bool tm = ctx->timing && (timing || a.record_events);
    qk_phase_events pe;
    QK_TRY(pe.begin(ctx, tm, ev_base));
    const int nblk = s->nblk;
    const int DB = (nblk % 8 == 0) ? 8 : (nblk % 4 == 0) ? 4 : (nblk % 2 == 0) ? 2 : 1;
    int NQ = 4;
    while (NQ > 1 && (size_t)NQ * nblk * 1024 > 64 * 1024) NQ >>= 1;
    const size_t lds = (size_t)NQ * nblk * 1024 + (size_t)NQ * 16 * 4 + 64;
    if (lds > 160 * 1024) QK_FAIL(QK_ERR_UNSUPPORTED, "dense scan: d=%d too large for the LDS query tile", s->d);
    // (from 32768 rows on the prefiltered form below is the faster nearest-centroid search too -- 1024 queries: 32768 rows 84 -> 71 us,
    //  65536 rows 154 -> 96 us; at 16384 rows 50 against 54 us the fused fp32 argmin stays)
    const bool pf_k1 = k == 1 && nrows >= 32768 && Q <= 16384 && a.x && a.out_ids && !qk_env_set("QK_NO_DENSE_PF") && qk_dense_pf_supported(ctx, s, Q, nrows, 1);
    const bool to_argmin = k == 1 && !pf_k1 && nrows > 0 && s->max_id_seen < ((int64_t)1 << 32) && s->min_id_seen >= 0 && !qk_env_set("QK_NO_ARGMIN");
    const bool argmin_apf = to_argmin && a.x && ((uintptr_t)a.x & 15) == 0 && qk_assign_pf_supported(Q, nrows, s->d, a.metric) &&
                            !qk_env_set("QK_NO_DENSE_APF");
    // a query preparation left pending by the caller (qk_prep_queries(.., defer)) is folded into the fp32 nearest-centroid kernel;
    // every other form below wants the prepared queries in place
    const bool fuse_prep = ctx->prep_pending && to_argmin && !argmin_apf && a.x == ctx->prep_x && Q == ctx->prep_Q &&
                           ((uintptr_t)a.x & 15) == 0 && s->d % 4 == 0 && a.xq4 == (const float4 *)ctx->qprep &&
                           !qk_env_set("QK_NO_FUSED_PREP");
    if (!fuse_prep) QK_TRY(qk_prep_flush(ctx));
    if (to_argmin) {
        // nprobe = 1 / nearest centroid: fused argmin, no key matrix
        const size_t lds_a = lds + (size_t)4 * NQ * 16 * 8;
        // many rows (PartitionManager::add's parent search, a nearest-list search of a huge batch): the k-means assign's bf16
        // prefilter (qk_assign_pf.hip) with the list's ids as the tie order -- the same packed words, ~4x the rate
        const float *rm = nullptr;
        bool apf = argmin_apf;
        if (apf) QK_TRY(qk_store_rowmajor(s, pt.row_off, nrows, &rm));
        if (!rm) apf = false;  // (a list beyond the row-major cap, a table not yet synchronised)
        const size_t apf_bytes = apf ? qk_assign_pf_scratch_bytes(nrows, s->d) : 0;
        QK_TRY(qk_ws_reserve(ctx, (size_t)Q * 8 + apf_bytes + 8192));
        unsigned long long *best64 = (unsigned long long *)qk_ws_alloc(ctx, (size_t)Q * 8);
        void *apf_scratch = apf ? qk_ws_alloc(ctx, apf_bytes) : nullptr;
        if (!best64 || (apf && !apf_scratch)) QK_FAIL(QK_ERR_OOM, "dense argmin: workspace exhausted");
        // the prep kernel of this batch (or the fused kernel of the previous one) left a ready-made "nothing yet" array (first use only)
        bool preinit = ctx->qprep_best64 && ctx->qprep_best64_n == Q;
        if (fuse_prep && !preinit) {  // this batch's array is not known to be clean (first batch of a shape): clear it here
            QK_HIP(hipMemsetAsync(ctx->qprep_best64, 0xFF, (size_t)Q * 8, st));
            preinit = true;
        }
        if (preinit) {
            best64 = ctx->qprep_best64;
            ctx->qprep_best64_n = 0;
            ctx->best64_clean[ctx->best64_cur] = 0;
        }
        const int num_cus_a = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        QK_TRY(pe.mark(0));
        QK_TRY(pe.mark(1));
        if (apf) {
            QK_TRY(qk_assign_pf_launch(ctx, a.x, Q, rm, nrows, s->d, a.metric, s->norms + pt.row_off, s->ids + pt.row_off, nullptr, nullptr,
                                       best64, a.out_dist != nullptr, apf_scratch));
            ctx->last_scan_kernel = "k_assign_pf";
        } else {
        if (!preinit) QK_HIP(hipMemsetAsync(best64, 0xFF, (size_t)Q * 8, st));
        ArgminParams ap;
        ap.vecs = (const float4 *)s->vecs;
        ap.norms = s->norms;
        ap.ids = s->ids + pt.row_off;
        ap.row_off = pt.row_off;
        ap.nrows = nrows;
        ap.nblk = nblk;
        ap.xq4 = a.xq4;
        ap.xn = a.xn;
        ap.Q = Q;
        ap.metric = a.metric;
        ap.best64 = best64;
        ap.x = nullptr;
        if (fuse_prep) {
            ap.x = a.x;
            ap.d = s->d;
            ap.xq4_out = (float4 *)a.xq4;
            ap.xn_out = (float *)a.xn;
            ap.xp4_out = (float4 *)ctx->qprep_xp4;
            ap.zero16 = (uint4 *)ctx->qprep_zero;
            ap.n_zero16 = ctx->qprep_zero ? ctx->prep_zero16 : 0;
            ap.best64_next = ctx->best64_buf[ctx->best64_cur ^ 1];
            ap.n_next = Q;
            ctx->prep_pending = false;
            ctx->best64_clean[ctx->best64_cur ^ 1] = Q;
        }
        const int64_t qgroups = (Q + NQ * 16 - 1) / (NQ * 16);
        const int ntile = (nrows + 15) / 16;
        // two workgroups per CU, at least 8 row tiles each (the query tile staging costs about as much as 8 tiles)
        static const int am_wgs = std::max(1, qk_env_int("QK_ARGMIN_WG_PER_CU", 2));
        static const int am_min_tiles = std::max(4, qk_env_int("QK_ARGMIN_MIN_TILES", 8));
        const int64_t want_chunks = std::max<int64_t>(1, ((int64_t)am_wgs * num_cus_a + qgroups - 1) / qgroups);
        int tiles_per_wg = (int)std::max<int64_t>(am_min_tiles, (ntile + want_chunks - 1) / want_chunks);
        tiles_per_wg = qk_round_up(tiles_per_wg, 4);
        ap.tiles_per_wg = tiles_per_wg;
        const int rchunks = std::max(1, (ntile + tiles_per_wg - 1) / tiles_per_wg);
        dim3 grid((unsigned)qgroups, (unsigned)rchunks);
#define AM_CASE(D_, N_) \
    if (DB == D_ && NQ == N_) QK_TRY((launch_argmin_t<D_, N_>(st, grid, lds_a, ap)));
        AM_CASE(8, 4) AM_CASE(8, 2) AM_CASE(8, 1) AM_CASE(4, 4) AM_CASE(4, 2) AM_CASE(4, 1)
        AM_CASE(2, 4) AM_CASE(2, 2) AM_CASE(2, 1) AM_CASE(1, 4) AM_CASE(1, 2) AM_CASE(1, 1)
#undef AM_CASE
        }
        QK_TRY(pe.mark(2));
        if (a.packed_out && preinit) {
            *a.packed_out = best64;  // lives in the query prep buffer: valid until the next batch is prepared
        } else {
            hipLaunchKernelGGL(k_argmin_finish, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, st, best64, Q, a.metric,
                               a.sqrt_l2 ? 1 : 0, a.out_ids, a.out_dist);
        }
        QK_HIP(hipGetLastError());
        QK_TRY(pe.mark(3));
        if (timing) {
            QK_TRY(qk_pinned_reserve(ctx, 64));
            int32_t *hs = (int32_t *)ctx->pinned;
            QK_HIP(hipStreamSynchronize(st));
            hs[0] = 1;
            hs[1] = 0;
            hs[7] = (int32_t)std::min<int64_t>(Q, INT32_MAX);  // every query scans the one list
            int64_t rows = nrows;
            memcpy(hs + 2, &rows, sizeof(rows));
        }
        return QK_OK;
    }
    if (a.x && a.out_ids && s->min_id_seen >= 0 && qk_coarse_small_supported(s, Q, nrows, k)) {
        // a few dozen queries against a few thousand rows: keys + selection in one launch (k_coarse_small, qk_small.hip)
        QK_TRY(pe.mark(0));
        QK_TRY(pe.mark(1));
        QK_TRY(qk_launch_coarse_small(ctx, s, pt.row_off, nrows, a.x, Q, k, a.metric, a.sqrt_l2, a.out_ids, a.out_dist));
        QK_TRY(pe.mark(2));
        QK_TRY(pe.mark(3));
        if (timing) {
            QK_TRY(qk_pinned_reserve(ctx, 64));
            int32_t *hs = (int32_t *)ctx->pinned;
            QK_HIP(hipStreamSynchronize(st));
            hs[0] = 1;
            hs[1] = 0;
            hs[7] = (int32_t)std::min<int64_t>(Q, INT32_MAX);
            int64_t rows = nrows;
            memcpy(hs + 2, &rows, sizeof(rows));
        }
        return QK_OK;
    }
    static const bool no_fused = qk_env_set("QK_NO_DENSE_FUSED");
    if (a.xq4 && a.out_ids && !no_fused && qk_dense_fused_supported(ctx, s, Q, nrows, k)) {
        // 2 <= k <= 64, d <= 128, a few thousand rows: keys and their selection in one launch, no key matrix (qk_dense_fused.hip)
        QK_TRY(pe.mark(0));
        QK_TRY(pe.mark(1));
        ctx->last_scan_kernel = "k_dense_fused";
        QK_TRY(qk_dense_fused_device(ctx, s, pt.row_off, nrows, a));
        QK_TRY(pe.mark(2));
        QK_TRY(pe.mark(3));
        if (timing) {
            QK_TRY(qk_pinned_reserve(ctx, 64));
            int32_t *hs = (int32_t *)ctx->pinned;
            QK_HIP(hipStreamSynchronize(st));
            hs[0] = 1;
            hs[1] = 0;
            hs[7] = (int32_t)std::min<int64_t>(Q, INT32_MAX);
            int64_t rows = nrows;
            memcpy(hs + 2, &rows, sizeof(rows));
        }
        return QK_OK;
    }
    static const bool no_pf = qk_env_set("QK_NO_DENSE_PF");
    if (a.x && a.out_ids && !no_pf && qk_dense_pf_supported(ctx, s, Q, nrows, k)) {
        // 2 <= k <= 64, d <= 128, thousands of rows: no key matrix -- approximate keys on bf16 MFMA settle which rows can matter,
        // the exact keys of those candidates the answer (qk_dense_pf.hip)
        QK_TRY(pe.mark(0));
        QK_TRY(pe.mark(1));
        ctx->last_scan_kernel = "k_dense_pf";
        QK_TRY(qk_dense_pf_device(ctx, s, pt.row_off, nrows, a));
        QK_TRY(pe.mark(2));
        QK_TRY(pe.mark(3));
        if (timing) {
            QK_TRY(qk_pinned_reserve(ctx, 64));
            int32_t *hs = (int32_t *)ctx->pinned;
            QK_HIP(hipStreamSynchronize(st));
            hs[0] = 1;
            hs[1] = 0;
            hs[7] = (int32_t)std::min<int64_t>(Q, INT32_MAX);
            int64_t rows = nrows;
            memcpy(hs + 2, &rows, sizeof(rows));
        }
        return QK_OK;
    }
    const int64_t ld = qk_round_up64(std::max(nrows, 1), 16);
    const int Cm = qk_round_up(k + 64, 64);
    // beyond the LDS pool machinery: bisection select + sort (k_select_rows_large).  Also where it is simply faster: its cost is ~32
    // counting passes over the key row whatever k is, the pool selection's grows with k (1024 queries, us per coarse call, pool
    // form -> bisection; 4096 rows: k = 100 92 -> 90, 150 124 -> 91, 204 206 -> 90, 256 436 -> 91, 819 1927 -> 117; 16384 rows:
    // 204 292 -> 489, 256 973 -> 491, 819 2720 -> 1039; scripts/coarse_probe.py)
    // (its tie cut orders ids as unsigned: taken early only for stores without negative ids)
    const bool large_k = Cm > 1024 || (s->min_id_seen >= 0 && ((k >= 96 && nrows <= 8192) || (k >= 224 && nrows <= 32768)));
    if (k > QK_MAX_NPROBE) QK_FAIL(QK_ERR_UNSUPPORTED, "dense scan: k=%d exceeds QK_MAX_NPROBE=%d", k, QK_MAX_NPROBE);
    const int mc = Cm <= 128 ? 2 : Cm <= 256 ? 4 : Cm <= 512 ? 8 : 16;
    // query batching keeps the key matrix under 1 GiB
    int64_t qb = std::max<int64_t>(NQ * 16, ((int64_t)1 << 30) / (ld * 4));
    qb = std::min<int64_t>(Q, (qb / (NQ * 16)) * (NQ * 16));
    // sliced selection of long rows (see below): scratch for the slices' candidates
    static const int sel_slices = qk_env_int("QK_SELECT_SLICES", 1);  // probe: 0 = never
    int slices = 1, slice_len = 0;
    if (sel_slices && !large_k && k <= 64 && nrows >= 8192) {
        slice_len = 4096;
        slices = (nrows + slice_len - 1) / slice_len;
        while (slices * k > 1024) {  // the merge wave holds all candidates of a query
            slice_len *= 2;
            slices = (nrows + slice_len - 1) / slice_len;
        }
        if (slices < 2) slices = 1;
    }
    const size_t sl_bytes = slices > 1 ? (size_t)qb * slices * k * 12 + 512 : 0;
    QK_TRY(qk_ws_reserve(ctx, (size_t)qb * ld * 4 + sl_bytes + 4096));
    uint32_t *D = (uint32_t *)qk_ws_alloc(ctx, (size_t)qb * ld * 4);
    if (!D) QK_FAIL(QK_ERR_OOM, "dense scan: workspace exhausted");
    int64_t *sl_ids = nullptr;
    uint32_t *sl_ord = nullptr;
    if (slices > 1) {
        sl_ids = (int64_t *)qk_ws_alloc(ctx, (size_t)qb * slices * k * 8);
        sl_ord = (uint32_t *)qk_ws_alloc(ctx, (size_t)qb * slices * k * 4);
        if (!sl_ids || !sl_ord) QK_FAIL(QK_ERR_OOM, "dense scan: workspace exhausted");
    }
    const int num_cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    QK_TRY(pe.mark(0));
    QK_TRY(pe.mark(1));
    for (int64_t q0 = 0; q0 < Q; q0 += qb) {
        const int64_t nq = std::min(qb, Q - q0);
        DenseParams dp;
        dp.vecs = (const float4 *)s->vecs;
        dp.norms = s->norms;
        dp.row_off = pt.row_off;
        dp.nrows = nrows;
        dp.nblk = nblk;
        dp.xq4 = a.xq4 + q0 * nblk * 4;
        dp.xn = a.xn + q0;
        dp.Q = nq;
        dp.d = s->d;
        dp.metric = a.metric;
        dp.D = D;
        dp.ld = ld;
        const int64_t qgroups = (nq + NQ * 16 - 1) / (NQ * 16);
        const int ntile = (nrows + 15) / 16;
        // enough workgroups to fill the chip, at least 4 tiles (one per wave) each
        // staging the query tile costs as much as ~8 row tiles per wave: aim at one workgroup per CU, >= 16 tiles each
        // (two workgroups per CU from 65536 rows on: the key epilogue of one wave runs under the MFMAs of the other -- 1024 queries,
        //  nprobe 32: 430 -> 371 us; below that the choice made no difference worth having: 16384 rows 106 -> 103)
        int64_t want_chunks = std::max<int64_t>(1, ((int64_t)(ntile >= 4096 ? 2 : 1) * num_cus + qgroups - 1) / qgroups);
        int tiles_per_wg = (int)std::max<int64_t>(16, (ntile + want_chunks - 1) / want_chunks);
        tiles_per_wg = qk_round_up(tiles_per_wg, 4);
        dp.tiles_per_wg = tiles_per_wg;
        const int rchunks = std::max(1, (ntile + tiles_per_wg - 1) / tiles_per_wg);
        if (nrows > 0) {
            dim3 grid((unsigned)qgroups, (unsigned)rchunks);
#define DN_CASE(D_, N_) \
    if (DB == D_ && NQ == N_) QK_TRY((launch_dense_t<D_, N_>(st, grid, lds, dp)));
            DN_CASE(8, 4) DN_CASE(8, 2) DN_CASE(8, 1) DN_CASE(4, 4) DN_CASE(4, 2) DN_CASE(4, 1)
            DN_CASE(2, 4) DN_CASE(2, 2) DN_CASE(2, 1) DN_CASE(1, 4) DN_CASE(1, 2) DN_CASE(1, 1)
#undef DN_CASE
        }
        if (q0 + qb >= Q) QK_TRY(pe.mark(2));
        SelectParams sp;
        sp.D = D;
        sp.ld = ld;
        sp.nrows = nrows;
        sp.ids = s->ids + pt.row_off;
        sp.k = k;
        sp.Cm = Cm;
        sp.metric = a.metric;
        sp.sqrt_l2 = a.sqrt_l2 ? 1 : 0;
        sp.out_ids = a.out_ids + q0 * k;
        sp.out_dist = a.out_dist ? a.out_dist + q0 * k : nullptr;
        if (large_k) {
            SelectLargeParams lp;
            lp.D = D;
            lp.ld = ld;
            lp.nrows = nrows;
            lp.ids = s->ids + pt.row_off;
            lp.k = k;
            int kp = 1;
            while (kp < k) kp <<= 1;
            lp.kp = kp;
            lp.metric = a.metric;
            lp.sqrt_l2 = a.sqrt_l2 ? 1 : 0;
            lp.out_ids = a.out_ids + q0 * k;
            lp.out_dist = a.out_dist ? a.out_dist + q0 * k : nullptr;
            const size_t lds_l = (size_t)kp * 12;
            QK_HIP(hipFuncSetAttribute((const void *)k_select_rows_large, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l));
            hipLaunchKernelGGL(k_select_rows_large, dim3((unsigned)nq), dim3(256), lds_l, st, lp);
            continue;
        }
        const size_t lds_s = (size_t)Cm * 12;
        // long rows (>= 8192 keys, k <= 64): one wave per 4096-key slice instead of one wave per row -- a single wave walking
        // 65536 keys twice was 2/3 of the coarse call at 65536 centroids -- and a one-wave merge of the slices' candidates
        sp.slices = 1;
        sp.slice_len = 0;
        sp.out_ord = nullptr;
        unsigned nwaves = (unsigned)nq;
        if (sl_ids) {
            sp.slices = slices;
            sp.slice_len = slice_len;
            sp.out_ids = sl_ids;
            sp.out_ord = sl_ord;
            sp.out_dist = nullptr;
            nwaves = (unsigned)(nq * slices);
        }
        switch (mc) {
            case 2: hipLaunchKernelGGL((k_select_rows<2>), dim3(nwaves), dim3(64), lds_s, st, sp); break;
            case 4: hipLaunchKernelGGL((k_select_rows<4>), dim3(nwaves), dim3(64), lds_s, st, sp); break;
            case 8: hipLaunchKernelGGL((k_select_rows<8>), dim3(nwaves), dim3(64), lds_s, st, sp); break;
            default: hipLaunchKernelGGL((k_select_rows<16>), dim3(nwaves), dim3(64), lds_s, st, sp); break;
        }
        if (sl_ids)
            QK_TRY(qk_launch_merge_slices(ctx, sl_ids, sl_ord, nq, slices, k, a.metric, a.sqrt_l2, a.out_ids + q0 * k,
                                          a.out_dist ? a.out_dist + q0 * k : nullptr));
    }
    QK_HIP(hipGetLastError());
    QK_TRY(pe.mark(3));
    if (timing) {
        // same scalar block the generic path reports: n_items, -, rows scanned
        QK_TRY(qk_pinned_reserve(ctx, 64));
        int32_t *hs = (int32_t *)ctx->pinned;
        QK_HIP(hipStreamSynchronize(st));
        hs[0] = 1;
        hs[1] = 0;
        hs[7] = (int32_t)std::min<int64_t>(Q, INT32_MAX);  // every query scans the one list
        int64_t rows = nrows;
        memcpy(hs + 2, &rows, sizeof(rows));
    }
    return QK_OK;
}

// out[i] = in[0] + ... + in[i-1], one workgroup: thread t owns a contiguous slice, the 1024 slice sums are scanned through LDS
// (the pair offsets of the wide-k pipeline: Q x nprobe values, a step that is followed by a full scan of those lists)
__global__ __launch_bounds__(1024) void k_exclusive_scan_i64(const int64_t *__restrict__ in, int64_t *__restrict__ out, int64_t len) {
    __shared__ int64_t part[1024];
    const int64_t per = (len + 1023) / 1024;
    const int64_t b = min(len, (int64_t)threadIdx.x * per), e = min(len, b + per);
    int64_t s = 0;
    for (int64_t i = b; i < e; i++) s += in[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int64_t run = part[threadIdx.x] - s;
    for (int64_t i = b; i < e; i++) {
        const int64_t v = in[i];
        out[i] = run;
        run += v;
    }
}

int qk_widek_device(qk_ctx *ctx, qk_store *s, const qk_scan_args &a, qk_timing *timing, int ev_base) {
    const int64_t Q = a.Q;
    const int k = a.k;
    if (k > QK_MAX_WIDE_K) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_scan: k=%d exceeds %d", k, QK_MAX_WIDE_K);
    QK_TRY(qk_store_sync_table(s));
    const int npids = (int)s->parts.size();
    const int P = a.all_lists ? npids : a.P;
    if (P <= 0 || npids <= 0) QK_FAIL(QK_ERR_INVALID, "qk_scan: no lists to scan");
    hipStream_t st = ctx->stream;
    int kp = 1;
    while (kp < k) kp <<= 1;
    // queries per pass: the keys of one pass stay under 2^29 (2 GiB)
    const int64_t per_query_ub = std::max<int64_t>(1, (int64_t)P * std::max<int64_t>(1, s->max_size));
    const int64_t qc = std::max<int64_t>(1, std::min<int64_t>(Q, ((int64_t)1 << 29) / per_query_ub));
    if (per_query_ub > ((int64_t)1 << 30)) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_scan: k=%d with %d lists per query is too large", k, P);
    const int nblk = s->nblk;
    // per-call phase events of THIS pipeline (the inner qk_scan_device runs without a qk_timing, so it records none):
    // [0,1) pair sizes + offsets of the first pass, [1,2) key emission passes, [2,3) the last selection.  Deferred modes
    // are served by the inner call's own scan-kernel events.
    qk_phase_events pe;
    pe.ctx = ctx;
    pe.tm = ctx->timing && (timing || a.record_events);
    pe.dtm = false;
    pe.ev_base = ev_base;
    QK_TRY(pe.mark(0));
    for (int64_t q0 = 0; q0 < Q; q0 += qc) {
        const int64_t nq = std::min(qc, Q - q0);
        const int64_t npairs = nq * P;
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_sizes = 0, o_base = al((size_t)(npairs + 1) * 8);
        const size_t o_keys = o_base + al((size_t)(npairs + 1) * 8);
        const size_t need = o_keys + (size_t)nq * per_query_ub * 4 + 256;
        QK_TRY(qk_aps_reserve(ctx, need));  // (the scan below recycles ctx->ws; this buffer survives it)
        char *B = ctx->aps;
        int64_t *sizes = (int64_t *)(B + o_sizes), *pair_base = (int64_t *)(B + o_base);
        uint32_t *keys = (uint32_t *)(B + o_keys);
        const int64_t *pids = a.pids ? a.pids + q0 * P : nullptr;
        hipLaunchKernelGGL(k_pair_sizes, dim3((unsigned)((npairs + 256) / 256)), dim3(256), 0, st, pids, npairs, P, s->d_size, npids, sizes);
        hipLaunchKernelGGL(k_exclusive_scan_i64, dim3(1), dim3(1024), 0, st, sizes, pair_base, npairs + 1);
        if (q0 == 0) QK_TRY(pe.mark(1));
        qk_scan_args e = a;
        e.x = a.x + q0 * s->d;
        e.xq4 = a.xq4 + q0 * nblk * 4;
        e.xn = a.xn + q0;
        e.Q = nq;
        e.pids = pids;
        e.key_out = keys;
        e.pair_base = pair_base;
        e.out_ids = nullptr;
        e.out_dist = nullptr;
        e.record_events = false;
        QK_TRY(qk_scan_device(ctx, s, e, nullptr, ev_base));
        if (q0 + qc >= Q) QK_TRY(pe.mark(2));
        WideKParams w;
        w.keys = keys;
        w.pair_base = pair_base;
        w.pids = pids;
        w.pt_off = s->d_off;
        w.ids = s->ids;
        w.P = P;
        w.k = k;
        w.kp = kp;
        w.metric = a.metric;
        w.sqrt_l2 = a.sqrt_l2 ? 1 : 0;
        w.out_ids = a.out_ids + q0 * k;
        w.out_dist = a.out_dist ? a.out_dist + q0 * k : nullptr;
        const size_t lds = (size_t)kp * 12;
        QK_HIP(hipFuncSetAttribute((const void *)k_select_pairs_large, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_select_pairs_large, dim3((unsigned)nq), dim3(256), lds, st, w);
        QK_HIP(hipGetLastError());
    }
    QK_TRY(pe.mark(3));
    if (timing) {
        QK_TRY(qk_pinned_reserve(ctx, 64));
        QK_HIP(hipStreamSynchronize(st));
        memset(ctx->pinned, 0, 32);
    }
    return QK_OK;
}

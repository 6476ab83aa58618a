// qk_scan.hip -- the partition scan: distance (fp32 MFMA) + fused top-k, and its grouping / merge stages.
//
// Replaces, for one batch of queries (citations relative to the reference checkout):
//   QueryCoordinator::scan_partitions  src/cpp/src/query_coordinator.cpp:659-673
//     serial_scan         :471-611   (per query: scan_list over its nprobe partitions, TopkBuffer)
//     batched_serial_scan :675-799   (group queries by partition :707-721, batched_scan_list per group,
//                                     merge into per-query buffers :752-758, pad :764-788)
//   scan_list / batched_scan_list      src/cpp/include/list_scanning.h:241-366
//   TypedTopKBuffer                    src/cpp/include/list_scanning.h:41-204 (append + flush -> "pool" below)
//
// Pipeline (all on one stream, no host round trip):
//   k_prep_queries   x[Q][d] -> fragment-ordered copy + squared norms
//   k_group_count / k_group_scan / k_group_scatter   (q,p) pairs -> per-partition query groups, work-item table
//   k_scan<DB,MAXCH> persistent workgroups pull work items (partition, 16-query tile, row chunk); each wave streams
//                    its rows as contiguous 1 KiB float4 loads straight into MFMA A operands, queries come from LDS,
//                    v_mfma_f32_16x16x4_f32 accumulates the dot products in natural k order; candidates that beat
//                    the running k-th best are appended to a per-(wave,query) LDS pool that is compacted by rank
//   k_merge<MAXCH>   one wave per query merges its candidate lists, applies sqrt / padding, writes [Q][k]
//
// Roofline: HBM.  Algorithmic bytes per batch = sum over unique probed partitions of n_p*d*4.
#include "qk_internal.h"

#include <algorithm>
#include <climits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- orderable keys: smaller = better --------------------------------------------------------------
__device__ __forceinline__ uint32_t ord_from_l2(float d2) { return __float_as_uint(d2); }  // d2 >= +0
__device__ __forceinline__ uint32_t ord_from_ip(float ip) {
    uint32_t b = __float_as_uint(ip);
    uint32_t asc = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    return ~asc;
}
__device__ __forceinline__ float ip_from_ord(uint32_t o) {
    uint32_t asc = ~o;
    uint32_t b = (asc & 0x80000000u) ? (asc ^ 0x80000000u) : ~asc;
    return __uint_as_float(b);
}
// faiss::knn_L2sqr expansion, clamped at 0 (oracle: l2sqr_expanded)
__device__ __forceinline__ float l2_expanded(float xn, float yn, float ip) {
    float r = __fmaf_rn(-2.0f, ip, xn + yn);
    return r < 0.0f ? 0.0f : r;
}

// ---- query preparation -----------------------------------------------------------------------------
// xq4[(q*nblk + c)*4 + g] = {x[q][16c+g], x[q][16c+g+4], x[q][16c+g+8], x[q][16c+g+12]}  (B-operand order)
__global__ void k_prep_queries(const float *__restrict__ x, int64_t Q, int d, int nblk, float4 *__restrict__ xq4,
                               float *__restrict__ xn) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t per_q = (int64_t)nblk * 4;
    if (idx < Q * per_q) {
        int64_t q = idx / per_q;
        int rem = (int)(idx - q * per_q);
        int c = rem >> 2, g = rem & 3;
        const float *s = x + q * d;
        int col = 16 * c + g;
        float4 v;
        v.x = col < d ? s[col] : 0.0f;
        v.y = col + 4 < d ? s[col + 4] : 0.0f;
        v.z = col + 8 < d ? s[col + 8] : 0.0f;
        v.w = col + 12 < d ? s[col + 12] : 0.0f;
        xq4[idx] = v;
    }
    if (idx < Q) {
        const float *s = x + idx * d;
        float acc = 0.0f;
        for (int k = 0; k < d; k++) acc = __fmaf_rn(s[k], s[k], acc);
        xn[idx] = acc;
    }
}

// ---- grouping ---------------------------------------------------------------------------------------
struct GroupParams {
    const int64_t *pids;  // [Q*P] or nullptr (all_lists: pair i -> list i % P)
    int64_t npairs;
    int P;
    const int32_t *pt_size;
    int npids;
    int chunk_rows;
    int32_t *g_cnt;     // [npids]
    int32_t *g_cursor;  // [npids]
    int32_t *g_qoff;    // [npids+1]
    int32_t *g_ioff;    // [npids+1]
    int32_t *n_items;   // [1]
    int32_t *grouped_q; // [npairs]
    int32_t *pair_pos;  // [npairs]
    int64_t *n_rows_unique;  // [1] sum of sizes of partitions with >=1 query (algorithmic bytes / (d*4))
};

__device__ __forceinline__ int pair_pid(const GroupParams &G, int64_t i) {
    int64_t p = G.pids ? G.pids[i] : (i % G.P);
    if (p < 0 || p >= G.npids) return -1;
    return G.pt_size[p] > 0 ? (int)p : -1;
}

__global__ void k_group_count(GroupParams G) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.npairs) return;
    int p = pair_pid(G, i);
    if (p >= 0) atomicAdd(&G.g_cnt[p], 1);
}

// single workgroup of 1024 threads: exclusive scans of per-partition query counts and work-item counts
__global__ __launch_bounds__(1024) void k_group_scan(GroupParams G) {
    __shared__ int s_q[1024];
    __shared__ int s_i[1024];
    __shared__ long long s_r[1024];
    const int tid = threadIdx.x;
    const int per = (G.npids + 1023) / 1024;
    const int b = tid * per, e = min(G.npids, b + per);
    int sq = 0, si = 0;
    long long sr = 0;
    for (int p = b; p < e; p++) {
        int c = G.g_cnt[p];
        if (c > 0) {
            int sz = G.pt_size[p];
            sq += c;
            si += ((c + 15) >> 4) * ((sz + G.chunk_rows - 1) / G.chunk_rows);
            sr += sz;
        }
    }
    s_q[tid] = sq;
    s_i[tid] = si;
    s_r[tid] = sr;
    __syncthreads();
    if (tid == 0) {
        int aq = 0, ai = 0;
        long long ar = 0;
        for (int t = 0; t < 1024; t++) {
            int vq = s_q[t], vi = s_i[t];
            s_q[t] = aq;
            s_i[t] = ai;
            aq += vq;
            ai += vi;
            ar += s_r[t];
        }
        G.g_qoff[G.npids] = aq;
        G.g_ioff[G.npids] = ai;
        *G.n_items = ai;
        *G.n_rows_unique = ar;
    }
    __syncthreads();
    int aq = s_q[tid], ai = s_i[tid];
    for (int p = b; p < e; p++) {
        G.g_qoff[p] = aq;
        G.g_ioff[p] = ai;
        int c = G.g_cnt[p];
        if (c > 0) {
            aq += c;
            ai += ((c + 15) >> 4) * ((G.pt_size[p] + G.chunk_rows - 1) / G.chunk_rows);
        }
    }
}

__global__ void k_group_scatter(GroupParams G) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.npairs) return;
    int p = pair_pid(G, i);
    int pos = -1;
    if (p >= 0) {
        pos = atomicAdd(&G.g_cursor[p], 1);
        G.grouped_q[G.g_qoff[p] + pos] = (int32_t)(i / G.P);
    }
    G.pair_pos[i] = pos;
}

// ---- LDS pool compaction (the TopkBuffer::flush of this design) ------------------------------------
// Keeps the k best of n entries under the total order (ord, id, position) and leaves them sorted in [0,k).
template <int MAXCH>
__device__ __forceinline__ int compact_pool(uint32_t *ord, int64_t *id, int n, int k, int lane) {
    uint32_t o[MAXCH];
    int64_t d[MAXCH];
    int rk[MAXCH];
#pragma unroll
    for (int i = 0; i < MAXCH; i++) {
        int e = lane + 64 * i;
        bool has = e < n;
        o[i] = has ? ord[e] : 0xFFFFFFFFu;
        d[i] = has ? id[e] : LLONG_MAX;
        rk[i] = 0;
    }
    for (int t = 0; t < n; t++) {
        uint32_t ot = ord[t];
        int64_t it = id[t];
#pragma unroll
        for (int i = 0; i < MAXCH; i++) {
            int e = lane + 64 * i;
            bool less = (ot < o[i]) || (ot == o[i] && (it < d[i] || (it == d[i] && t < e)));
            rk[i] += less ? 1 : 0;
        }
    }
#pragma unroll
    for (int i = 0; i < MAXCH; i++) {
        int e = lane + 64 * i;
        if (e < n && rk[i] < k) {
            ord[rk[i]] = o[i];
            id[rk[i]] = d[i];
        }
    }
    return n < k ? n : k;
}

// ---- the scan kernel ---------------------------------------------------------------------------------
struct ScanParams {
    const float4 *vecs;
    const float *norms;
    const int64_t *ids;
    const int64_t *pt_off;
    const int32_t *pt_size;
    int npids;
    int nblk;
    const float4 *xq4;
    const float *xn;
    const int32_t *grouped_q;
    const int32_t *g_cnt;
    const int32_t *g_qoff;
    const int32_t *g_ioff;
    const int32_t *n_items;
    int32_t *item_counter;
    uint32_t *gtau;  // [Q] shared running bound per query, or nullptr
    int chunk_rows;
    int k;
    int C;  // pool capacity per (wave, query); k <= C - 4
    int metric;
    uint32_t *cand_ord;  // [slot][k]
    int64_t *cand_id;    // [slot][k]
    int32_t *cand_cnt;   // [slot]
};

template <int DB, int MAXCH>
__global__ __launch_bounds__(256) void k_scan(ScanParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int nblk = P.nblk, C = P.C, k = P.k;
    const bool l2 = P.metric == QK_METRIC_L2;
    float4 *qs = (float4 *)smem;
    const size_t wave_bytes = (size_t)16 * C * 12;
    unsigned char *wbase = smem + (size_t)nblk * 1024 + wave * wave_bytes;
    int64_t *pool_id = (int64_t *)wbase;                      // [16][C]
    uint32_t *pool_ord = (uint32_t *)(wbase + (size_t)16 * C * 8);  // [16][C]
    volatile int *s_item = (volatile int *)(smem + (size_t)nblk * 1024 + QK_WAVES * wave_bytes);
    const int n_items = *P.n_items;
    constexpr int NCD_UNUSED = 0;
    (void)NCD_UNUSED;
    const int ncd = nblk / DB;  // d-chunks per tile

    for (;;) {
        if (tid == 0) *s_item = atomicAdd(P.item_counter, 1);
        __syncthreads();
        const int item = *s_item;
        if (item >= n_items) break;
        // ---- decode the work item --------------------------------------------------------------
        int lo = 0, hi = P.npids;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (P.g_ioff[mid] <= item)
                lo = mid;
            else
                hi = mid;
        }
        const int p = lo;
        const int local = item - P.g_ioff[p];
        const int size_p = P.pt_size[p];
        const int64_t row_off = P.pt_off[p];
        const int nchunk = (size_p + P.chunk_rows - 1) / P.chunk_rows;
        const int qt = local / nchunk, ch = local - qt * nchunk;
        const int nq = min(16, P.g_cnt[p] - 16 * qt);
        const int myq = (j < nq) ? P.grouped_q[P.g_qoff[p] + 16 * qt + j] : -1;
        // ---- query tile -> LDS in B-operand lane order --------------------------------------------
        for (int cb = wave; cb < nblk; cb += QK_WAVES) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (myq >= 0) v = P.xq4[((int64_t)myq * nblk + cb) * 4 + g];
            qs[cb * 64 + lane] = v;
        }
        __syncthreads();
        // ---- this wave's rows ---------------------------------------------------------------------
        const int row_c0 = ch * P.chunk_rows;
        const int nrows = min(size_p - row_c0, P.chunk_rows);
        const int ntile = (nrows + 15) >> 4;
        const int tpw = (ntile + QK_WAVES - 1) / QK_WAVES;
        const int t0 = wave * tpw, t1 = min(ntile, t0 + tpw);
        uint32_t tau = 0xFFFFFFFFu;
        if (myq >= 0 && P.gtau) tau = __hip_atomic_load(&P.gtau[myq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int cnt = 0;
        const float xnj = (myq >= 0 && l2) ? P.xn[myq] : 0.0f;
        uint32_t *my_ord = pool_ord + j * C;
        int64_t *my_id = pool_id + j * C;

        if (t1 > t0) {
            const int64_t tile_abs0 = (row_off >> 4) + (row_c0 >> 4) + t0;
            const float4 *src = P.vecs + tile_abs0 * nblk * 64 + lane;
            const float4 *nsrc = (const float4 *)(P.norms + (tile_abs0 << 4)) + g;  // +4 float4 per tile
            const int nsteps = (t1 - t0) * ncd;
            float4 a0[DB], a1[DB];
            float4 yn_cur = make_float4(0.f, 0.f, 0.f, 0.f), yn_next = yn_cur;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            int dch = 0;       // d-chunk of the step being computed
            int tile = t0;     // tile of the step being computed
            int ldch = 0;      // d-chunk of the step being loaded
            int ltile = 0;     // tiles loaded so far (relative)

#define QK_LOAD(A, S)                                                 \
    {                                                                 \
        const float4 *pp_ = src + (int64_t)(S) * (DB * 64);           \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) A[b_] = pp_[b_ * 64]; \
        if (ldch == 0) {                                              \
            if (l2) yn_next = nsrc[(int64_t)ltile * 4];               \
            ltile++;                                                  \
        }                                                             \
        if (++ldch == ncd) ldch = 0;                                  \
    }

#define QK_STEP(A)                                                                                         \
    {                                                                                                      \
        if (dch == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};                                                   \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) {                                                \
            const float4 bq_ = qs[(dch * DB + b_) * 64 + lane];                                            \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].x, bq_.x, acc, 0, 0, 0);                      \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].y, bq_.y, acc, 0, 0, 0);                      \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].z, bq_.z, acc, 0, 0, 0);                      \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].w, bq_.w, acc, 0, 0, 0);                      \
        }                                                                                                  \
        if (++dch == ncd) {                                                                                \
            dch = 0;                                                                                       \
            epilogue(tile);                                                                                \
            tile++;                                                                                        \
        }                                                                                                  \
    }

            auto epilogue = [&](int tl) {
                const int row0 = row_c0 + (tl << 4);
                const int64_t arow = row_off + row0 + 4 * g;
                const float yv[4] = {yn_cur.x, yn_cur.y, yn_cur.z, yn_cur.w};
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int row = row0 + 4 * g + reg;
                    const bool valid = (myq >= 0) && (row < size_p);
                    const float v = acc[reg];
                    const uint32_t ord = l2 ? ord_from_l2(l2_expanded(xnj, yv[reg], v)) : ord_from_ip(v);
                    const bool pass = valid && ord <= tau;
                    const uint64_t m = __ballot(pass);
                    if (m) {
                        const uint64_t gm = m & (0x0001000100010001ull << j);
                        if (pass) {
                            const int slot = cnt + __popcll(gm & ((1ull << lane) - 1ull));
                            my_ord[slot] = ord;
                            my_id[slot] = P.ids[arow + reg];
                        }
                        cnt += __popcll(gm);
                        uint64_t need = __ballot(cnt > C - 4) & 0xFFFFull;
                        while (need) {
                            const int jq = __ffsll((unsigned long long)need) - 1;
                            need &= need - 1;
                            const int n = __builtin_amdgcn_readlane(cnt, jq);
                            const int nn = compact_pool<MAXCH>(pool_ord + jq * C, pool_id + jq * C, n, k, lane);
                            if (j == jq) {
                                cnt = nn;
                                if (nn >= k) tau = min(tau, pool_ord[jq * C + k - 1]);
                            }
                        }
                    }
                }
                yn_cur = yn_next;
            };

            QK_LOAD(a0, 0);
            yn_cur = yn_next;
            int s = 0;
            while (s < nsteps) {
                if (s + 1 < nsteps) QK_LOAD(a1, s + 1);
                QK_STEP(a0);
                s++;
                if (s >= nsteps) break;
                if (s + 1 < nsteps) QK_LOAD(a0, s + 1);
                QK_STEP(a1);
                s++;
            }
#undef QK_LOAD
#undef QK_STEP
        }
        // ---- final compaction (sorts, caps at k) + emit ---------------------------------------------
        {
            uint64_t need = __ballot(cnt > 0) & 0xFFFFull;
            while (need) {
                const int jq = __ffsll((unsigned long long)need) - 1;
                need &= need - 1;
                const int n = __builtin_amdgcn_readlane(cnt, jq);
                const int nn = compact_pool<MAXCH>(pool_ord + jq * C, pool_id + jq * C, n, k, lane);
                if (j == jq) cnt = nn;
            }
            const int64_t slot0 = ((int64_t)item * QK_WAVES + wave) * 16;
            if (lane < 16) {
                P.cand_cnt[slot0 + lane] = cnt;
                if (P.gtau && myq >= 0 && cnt >= k) atomicMin(&P.gtau[myq], my_ord[k - 1]);
            }
            for (int jq = 0; jq < 16; jq++) {
                const int n = __builtin_amdgcn_readlane(cnt, jq);
                for (int e = lane; e < n; e += 64) {
                    P.cand_ord[(slot0 + jq) * k + e] = pool_ord[jq * C + e];
                    P.cand_id[(slot0 + jq) * k + e] = pool_id[jq * C + e];
                }
            }
        }
    }
}

// ---- merge kernel: one wave per query ---------------------------------------------------------------------
struct MergeParams {
    const int64_t *pids;  // [Q*P] or nullptr
    int P;
    int npids;
    const int32_t *pt_size;
    const int32_t *pair_pos;
    const int32_t *g_ioff;
    int chunk_rows;
    int k;
    int Cm;  // pool capacity, k <= Cm - 64
    int metric;
    const uint32_t *cand_ord;
    const int64_t *cand_id;
    const int32_t *cand_cnt;
    int64_t *out_ids;   // [Q][k]
    float *out_dist;    // [Q][k] or nullptr
    int sqrt_l2;        // 1: output sqrt(d2) (search results); 0: squared (k-means internals)
};

template <int MAXCH>
__global__ __launch_bounds__(64) void k_merge(MergeParams M) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int k = M.k, Cm = M.Cm;
    int64_t *pool_id = (int64_t *)smem;
    uint32_t *pool_ord = (uint32_t *)(smem + (size_t)Cm * 8);
    uint32_t tau = 0xFFFFFFFFu;
    int cnt = 0;
    for (int r = 0; r < M.P; r++) {
        const int64_t pi = q * M.P + r;
        const int pos = M.pair_pos[pi];
        if (pos < 0) continue;
        const int p = M.pids ? (int)M.pids[pi] : r;
        const int nchunk = (M.pt_size[p] + M.chunk_rows - 1) / M.chunk_rows;
        const int64_t item0 = (int64_t)M.g_ioff[p] + (int64_t)(pos >> 4) * nchunk;
        for (int64_t lst = 0; lst < (int64_t)nchunk * QK_WAVES; lst++) {
            const int64_t slot = (item0 * QK_WAVES + lst) * 16 + (pos & 15);
            const int n = M.cand_cnt[slot];
            for (int base = 0; base < n; base += 64) {
                const int e = base + lane;
                const bool has = e < n;
                const uint32_t o = has ? M.cand_ord[slot * k + e] : 0xFFFFFFFFu;
                const bool pass = has && o <= tau;
                const uint64_t m = __ballot(pass);
                if (m) {
                    if (pass) {
                        const int sl = cnt + __popcll(m & ((1ull << lane) - 1ull));
                        pool_ord[sl] = o;
                        pool_id[sl] = M.cand_id[slot * k + e];
                    }
                    cnt += __popcll(m);
                    if (cnt > Cm - 64) {
                        cnt = compact_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane);
                        if (cnt >= k) tau = min(tau, pool_ord[k - 1]);
                    }
                }
                // lists are sorted ascending: once a valid lane fails the bound, the rest of the list fails too
                if (__popcll(m) < min(64, n - base)) break;
            }
        }
    }
    cnt = compact_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane);
    for (int e = lane; e < k; e += 64) {
        int64_t oid = -1;
        float od = M.metric == QK_METRIC_IP ? -INFINITY : INFINITY;
        if (e < cnt) {
            oid = pool_id[e];
            const uint32_t o = pool_ord[e];
            if (M.metric == QK_METRIC_L2) {
                const float d2 = __uint_as_float(o);
                od = M.sqrt_l2 ? sqrtf(d2) : d2;
            } else {
                od = ip_from_ord(o);
            }
        }
        M.out_ids[q * k + e] = oid;
        if (M.out_dist) M.out_dist[q * k + e] = od;
    }
}

// ---- cross-rank merge: [G][Q][k] per-rank results -> [Q][k] ---------------------------------------
// in_key are SQUARED L2 distances / inner products (what qk_search returns with qk_ctx_set_squared_l2), so the
// merge runs on the same (key, id) order as the single-GPU path; sqrt is applied to the output.
template <int MAXCH>
__global__ __launch_bounds__(64) void k_merge_ranks(const int64_t *__restrict__ in_ids, const float *__restrict__ in_key, int G,
                                                    int64_t Q, int k, int Cm, int metric, int sqrt_l2, int64_t *out_ids,
                                                    float *out_dist) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    int64_t *pool_id = (int64_t *)smem;
    uint32_t *pool_ord = (uint32_t *)(smem + (size_t)Cm * 8);
    uint32_t tau = 0xFFFFFFFFu;
    int cnt = 0;
    for (int r = 0; r < G; r++) {
        const int64_t base0 = ((int64_t)r * Q + q) * k;
        for (int base = 0; base < k; base += 64) {
            const int e = base + lane;
            int64_t id = -1;
            uint32_t o = 0xFFFFFFFFu;
            if (e < k) {
                id = in_ids[base0 + e];
                const float v = in_key[base0 + e];
                o = metric == QK_METRIC_L2 ? ord_from_l2(v) : ord_from_ip(v);
            }
            const bool pass = id >= 0 && o <= tau;
            const uint64_t m = __ballot(pass);
            if (m) {
                if (pass) {
                    const int sl = cnt + __popcll(m & ((1ull << lane) - 1ull));
                    pool_ord[sl] = o;
                    pool_id[sl] = id;
                }
                cnt += __popcll(m);
                if (cnt > Cm - 64) {
                    cnt = compact_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane);
                    if (cnt >= k) tau = min(tau, pool_ord[k - 1]);
                }
            }
        }
    }
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

static int pick_maxch(int cap);

int qk_merge_topk_device(qk_ctx *ctx, const int64_t *in_ids, const float *in_key, int G, int64_t Q, int k, int metric,
                         int64_t *out_ids, float *out_dist, bool sqrt_l2) {
    if (Q <= 0) return QK_OK;
    const int Cm = qk_round_up(k + 64, 64);
    if (Cm > 1024) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_merge_topk: k=%d too large", k);
    const size_t lds = (size_t)Cm * 12;
    const int mc = Cm <= 128 ? 2 : Cm <= 256 ? 4 : Cm <= 512 ? 8 : 16;
    hipStream_t st = ctx->stream;
    switch (mc) {
        case 2: hipLaunchKernelGGL((k_merge_ranks<2>), dim3((unsigned)Q), dim3(64), lds, st, in_ids, in_key, G, Q, k, Cm, metric, sqrt_l2 ? 1 : 0, out_ids, out_dist); break;
        case 4: hipLaunchKernelGGL((k_merge_ranks<4>), dim3((unsigned)Q), dim3(64), lds, st, in_ids, in_key, G, Q, k, Cm, metric, sqrt_l2 ? 1 : 0, out_ids, out_dist); break;
        case 8: hipLaunchKernelGGL((k_merge_ranks<8>), dim3((unsigned)Q), dim3(64), lds, st, in_ids, in_key, G, Q, k, Cm, metric, sqrt_l2 ? 1 : 0, out_ids, out_dist); break;
        default: hipLaunchKernelGGL((k_merge_ranks<16>), dim3((unsigned)Q), dim3(64), lds, st, in_ids, in_key, G, Q, k, Cm, metric, sqrt_l2 ? 1 : 0, out_ids, out_dist); break;
    }
    QK_HIP(hipGetLastError());
    return QK_OK;
}

// ---- host orchestration -------------------------------------------------------------------------------------
template <int DB>
static void launch_scan_db(int maxch, dim3 grid, size_t lds, hipStream_t st, const ScanParams &sp) {
    switch (maxch) {
        case 1: hipLaunchKernelGGL((k_scan<DB, 1>), grid, dim3(256), lds, st, sp); break;
        case 2: hipLaunchKernelGGL((k_scan<DB, 2>), grid, dim3(256), lds, st, sp); break;
        case 4: hipLaunchKernelGGL((k_scan<DB, 4>), grid, dim3(256), lds, st, sp); break;
        default: hipLaunchKernelGGL((k_scan<DB, 8>), grid, dim3(256), lds, st, sp); break;
    }
}

template <int DB, int MAXCH>
static int set_scan_lds(size_t lds) {
    QK_HIP(hipFuncSetAttribute((const void *)k_scan<DB, MAXCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return QK_OK;
}

static int set_scan_lds_attr(int db, int maxch, size_t lds) {
#define QK_CASE(D, M) \
    if (db == D && maxch == M) return set_scan_lds<D, M>(lds);
    QK_CASE(1, 1) QK_CASE(1, 2) QK_CASE(1, 4) QK_CASE(1, 8) QK_CASE(2, 1) QK_CASE(2, 2) QK_CASE(2, 4) QK_CASE(2, 8)
    QK_CASE(4, 1) QK_CASE(4, 2) QK_CASE(4, 4) QK_CASE(4, 8) QK_CASE(8, 1) QK_CASE(8, 2) QK_CASE(8, 4) QK_CASE(8, 8)
#undef QK_CASE
    return QK_OK;
}

static int pick_maxch(int cap) { return cap <= 64 ? 1 : cap <= 128 ? 2 : cap <= 256 ? 4 : 8; }

int qk_scan_device(qk_ctx *ctx, qk_store *s, const qk_scan_args &a, qk_timing *timing, int ev_base) {
    const int64_t Q = a.Q;
    const int k = a.k;
    if (Q <= 0) return QK_OK;
    if (k <= 0) QK_FAIL(QK_ERR_INVALID, "qk_scan: k must be positive");
    if (k > QK_MAX_K) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_scan: k=%d exceeds QK_MAX_K=%d", k, QK_MAX_K);
    QK_TRY(qk_store_sync_table(s));
    const int npids = (int)s->parts.size();
    const int P = a.all_lists ? npids : a.P;
    hipStream_t st = ctx->stream;
    const bool tm = ctx->timing && (timing || a.record_events);
    // deferred timing: events are parked in the context and read by qk_ctx_read_timing (no sync here)
    hipEvent_t dev[4] = {nullptr, nullptr, nullptr, nullptr};
    const bool dtm = ctx->timing_mode == 2;
    if (dtm) {
        for (int i = 0; i < 4; i++) {
            if (!ctx->ev_free.empty()) {
                dev[i] = ctx->ev_free.back();
                ctx->ev_free.pop_back();
            } else {
                QK_HIP(hipEventCreate(&dev[i]));
            }
        }
    }

    // nothing to scan: pure padding (query_coordinator.cpp:459-497 zero-partitions case)
    const int64_t npairs = Q * (int64_t)P;

    // ---- geometry ----------------------------------------------------------------------------------
    const int nblk = s->nblk;
    const int DB = (nblk % 8 == 0) ? 8 : (nblk % 4 == 0) ? 4 : (nblk % 2 == 0) ? 2 : 1;
    // pool capacity per (wave, query): k + slack, limited by LDS (160 KiB per workgroup)
    const size_t lds_budget = 160 * 1024 - 64;
    const size_t q_bytes = (size_t)nblk * 1024;
    int C = k + std::max(28, std::min(k, 64));
    C = qk_round_up(C, 4);
    while ((size_t)QK_WAVES * 16 * C * 12 + q_bytes + 16 > lds_budget && C > k + 4) C -= 4;
    if ((size_t)QK_WAVES * 16 * C * 12 + q_bytes + 16 > lds_budget || C < k + 4)
        QK_FAIL(QK_ERR_UNSUPPORTED, "qk_scan: k=%d with d=%d does not fit the LDS top-k pools", k, s->d);
    if (C > 512) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_scan: pool capacity %d > 512", C);
    const int maxch = pick_maxch(C);
    const size_t lds_scan = q_bytes + (size_t)QK_WAVES * 16 * C * 12 + 16;
    const int Cm = qk_round_up(k + 64, 64);
    const int maxch_m = Cm <= 128 ? 2 : Cm <= 256 ? 4 : Cm <= 512 ? 8 : 16;
    const size_t lds_merge = (size_t)Cm * 12;

    const int num_cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    int64_t npresent = std::max<int64_t>(1, s->nlist);
    int64_t max_size = std::max<int64_t>(1, s->max_size);
    int64_t tiles_est = std::max<int64_t>(1, std::min<int64_t>(npairs, npresent + npairs / 16));
    int64_t desired_items = (int64_t)8 * num_cus;
    int64_t nchunk_target = std::max<int64_t>(1, (desired_items + tiles_est - 1) / tiles_est);
    int64_t chunk_rows = std::max<int64_t>(256, qk_round_up64((max_size + nchunk_target - 1) / nchunk_target, 64));
    int64_t max_nchunk = (max_size + chunk_rows - 1) / chunk_rows;
    int64_t tiles_bound = std::max<int64_t>(1, npairs / 16 + std::min<int64_t>(npresent, npairs));
    int64_t items_bound = tiles_bound * max_nchunk;
    int64_t slots_bound = items_bound * QK_WAVES * 16;

    // ---- workspace ---------------------------------------------------------------------------------
    size_t need = 0;
    auto add = [&](size_t b) { need += (b + 255) & ~(size_t)255; };
    add((size_t)Q * s->dpad * 4);            // xq4
    add((size_t)Q * 4);                      // xn
    add((size_t)npids * 4 * 2);              // g_cnt, g_cursor
    add((size_t)(npids + 1) * 4 * 2 + 512);  // g_qoff, g_ioff
    add(256);                                // n_items, item_counter, n_rows_unique
    add((size_t)std::max<int64_t>(npairs, 1) * 4 * 2);  // grouped_q, pair_pos
    add((size_t)Q * 4);                      // gtau
    add((size_t)slots_bound * k * 12 + (size_t)slots_bound * 4);
    need += 4096;
    QK_TRY(qk_ws_reserve(ctx, need));
    float4 *xq4 = (float4 *)qk_ws_alloc(ctx, (size_t)Q * s->dpad * 4);
    float *xn = (float *)qk_ws_alloc(ctx, (size_t)Q * 4);
    int32_t *g_cnt = (int32_t *)qk_ws_alloc(ctx, (size_t)npids * 4 * 2);
    int32_t *g_cursor = g_cnt + npids;
    int32_t *g_qoff = (int32_t *)qk_ws_alloc(ctx, (size_t)(npids + 1) * 4 * 2 + 512);
    int32_t *g_ioff = g_qoff + npids + 1;
    int32_t *scal = (int32_t *)qk_ws_alloc(ctx, 256);
    int32_t *n_items = scal, *item_counter = scal + 1;
    int64_t *n_rows_unique = (int64_t *)(scal + 2);
    int32_t *grouped_q = (int32_t *)qk_ws_alloc(ctx, (size_t)std::max<int64_t>(npairs, 1) * 4 * 2);
    int32_t *pair_pos = grouped_q + std::max<int64_t>(npairs, 1);
    uint32_t *gtau = (uint32_t *)qk_ws_alloc(ctx, (size_t)Q * 4);
    uint32_t *cand_ord = (uint32_t *)qk_ws_alloc(ctx, (size_t)slots_bound * k * 4);
    int64_t *cand_id = (int64_t *)qk_ws_alloc(ctx, (size_t)slots_bound * k * 8);
    int32_t *cand_cnt = (int32_t *)qk_ws_alloc(ctx, (size_t)slots_bound * 4);
    if (!xq4 || !xn || !g_cnt || !g_qoff || !scal || !grouped_q || !gtau || !cand_ord || !cand_id || !cand_cnt)
        QK_FAIL(QK_ERR_OOM, "qk_scan: workspace exhausted");

    if (tm) QK_HIP(hipEventRecord(ctx->ev[ev_base + 0], st));
    if (dtm) QK_HIP(hipEventRecord(dev[0], st));
    // ---- prep + grouping -------------------------------------------------------------------------------
    {
        int64_t total = std::max<int64_t>(Q * nblk * 4, Q);
        hipLaunchKernelGGL(k_prep_queries, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.x, Q, s->d, nblk, xq4, xn);
    }
    QK_HIP(hipMemsetAsync(g_cnt, 0, (size_t)npids * 4 * 2, st));
    QK_HIP(hipMemsetAsync(scal, 0, 256, st));
    QK_HIP(hipMemsetAsync(gtau, 0xFF, (size_t)Q * 4, st));
    GroupParams G;
    G.pids = a.all_lists ? nullptr : a.pids;
    G.npairs = npairs;
    G.P = std::max(P, 1);
    G.pt_size = s->d_size;
    G.npids = npids;
    G.chunk_rows = (int)chunk_rows;
    G.g_cnt = g_cnt;
    G.g_cursor = g_cursor;
    G.g_qoff = g_qoff;
    G.g_ioff = g_ioff;
    G.n_items = n_items;
    G.grouped_q = grouped_q;
    G.pair_pos = pair_pos;
    G.n_rows_unique = n_rows_unique;
    if (npairs > 0) hipLaunchKernelGGL(k_group_count, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, G);
    hipLaunchKernelGGL(k_group_scan, dim3(1), dim3(1024), 0, st, G);
    if (npairs > 0) hipLaunchKernelGGL(k_group_scatter, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, G);
    if (tm) QK_HIP(hipEventRecord(ctx->ev[ev_base + 1], st));
    if (dtm) QK_HIP(hipEventRecord(dev[1], st));

    // ---- scan ------------------------------------------------------------------------------------------------
    ScanParams sp;
    sp.vecs = (const float4 *)s->vecs;
    sp.norms = s->norms;
    sp.ids = s->ids;
    sp.pt_off = s->d_off;
    sp.pt_size = s->d_size;
    sp.npids = npids;
    sp.nblk = nblk;
    sp.xq4 = xq4;
    sp.xn = xn;
    sp.grouped_q = grouped_q;
    sp.g_cnt = g_cnt;
    sp.g_qoff = g_qoff;
    sp.g_ioff = g_ioff;
    sp.n_items = n_items;
    sp.item_counter = item_counter;
    sp.gtau = a.share_tau ? gtau : nullptr;
    sp.chunk_rows = (int)chunk_rows;
    sp.k = k;
    sp.C = C;
    sp.metric = a.metric;
    sp.cand_ord = cand_ord;
    sp.cand_id = cand_id;
    sp.cand_cnt = cand_cnt;
    if (npairs > 0 && npids > 0) {
        QK_TRY(set_scan_lds_attr(DB, maxch, lds_scan));
        int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds_scan));
        int64_t grid = std::min<int64_t>(items_bound, (int64_t)num_cus * wg_per_cu);
        grid = std::max<int64_t>(grid, 1);
        dim3 gd((unsigned)grid);
        switch (DB) {
            case 8: launch_scan_db<8>(maxch, gd, lds_scan, st, sp); break;
            case 4: launch_scan_db<4>(maxch, gd, lds_scan, st, sp); break;
            case 2: launch_scan_db<2>(maxch, gd, lds_scan, st, sp); break;
            default: launch_scan_db<1>(maxch, gd, lds_scan, st, sp); break;
        }
    }
    if (tm) QK_HIP(hipEventRecord(ctx->ev[ev_base + 2], st));
    if (dtm) QK_HIP(hipEventRecord(dev[2], st));

    // ---- merge ---------------------------------------------------------------------------------------------------
    MergeParams mp;
    mp.pids = G.pids;
    mp.P = P;
    mp.npids = npids;
    mp.pt_size = s->d_size;
    mp.pair_pos = pair_pos;
    mp.g_ioff = g_ioff;
    mp.chunk_rows = (int)chunk_rows;
    mp.k = k;
    mp.Cm = Cm;
    mp.metric = a.metric;
    mp.cand_ord = cand_ord;
    mp.cand_id = cand_id;
    mp.cand_cnt = cand_cnt;
    mp.out_ids = a.out_ids;
    mp.out_dist = a.out_dist;
    mp.sqrt_l2 = a.sqrt_l2 ? 1 : 0;
    switch (maxch_m) {
        case 2: hipLaunchKernelGGL((k_merge<2>), dim3((unsigned)Q), dim3(64), lds_merge, st, mp); break;
        case 4: hipLaunchKernelGGL((k_merge<4>), dim3((unsigned)Q), dim3(64), lds_merge, st, mp); break;
        case 8: hipLaunchKernelGGL((k_merge<8>), dim3((unsigned)Q), dim3(64), lds_merge, st, mp); break;
        default: hipLaunchKernelGGL((k_merge<16>), dim3((unsigned)Q), dim3(64), lds_merge, st, mp); break;
    }
    QK_HIP(hipGetLastError());
    if (tm) QK_HIP(hipEventRecord(ctx->ev[ev_base + 3], st));
    if (dtm) {
        QK_HIP(hipEventRecord(dev[3], st));
        if (ev_base == 4) {
            for (int i = 0; i < 4; i++) ctx->ev_pending.push_back(dev[i]);
        } else {  // coarse stage: keep (start, end)
            ctx->ev_pending_coarse.push_back(dev[0]);
            ctx->ev_pending_coarse.push_back(dev[3]);
            ctx->ev_free.push_back(dev[1]);
            ctx->ev_free.push_back(dev[2]);
        }
    }
    if (timing) {
        // device scalars come back through pinned memory; the caller synchronises before reading them
        QK_TRY(qk_pinned_reserve(ctx, 64));
        QK_HIP(hipMemcpyAsync(ctx->pinned, scal, 16, hipMemcpyDeviceToHost, st));
    }
    return QK_OK;
}

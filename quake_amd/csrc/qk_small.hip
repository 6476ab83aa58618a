// qk_small.hip -- QueryCoordinator::search for SMALL batches in ONE launch.
//
// A 1-query search through the batch pipeline is a chain of ~12 dependent launches (prep, coarse keys, select, convert,
// group, seed, scan, merge): ~100 us of launch boundaries around ~5 us of work (profiles/r01_latency_probe.json).  Here the
// whole path -- src/cpp/src/query_coordinator.cpp:612-657 (parent search -> scan_partitions), serial_scan :471-611,
// scan_list + TopkBuffer src/cpp/include/list_scanning.h:41-204,241-311 -- is one kernel:
//
//   grid = Q x W workgroups; the W workgroups of a query each
//     A. compute the coarse step for the query themselves (every centroid, canonical chains, one row per thread; the
//        centroid arena is L2 / Infinity-Cache resident) and select the nprobe nearest under the (key, id) order by
//        bisection on the key bits (and on the id bits among ties on the cut), so that no grid-wide barrier separates the
//        coarse step from the scan;
//     B. take the w-th slice of the rows of the probed partitions laid end to end, one row per thread and round, keep the
//        slice's k best in LDS (select_pool / compact_pool: the TopkBuffer append + flush of this design);
//     C. leave a sorted record; the LAST workgroup of the query to arrive (one atomic ticket; agent-scope release before,
//        acquire after) merges the W records, applies sqrt / padding and writes [k] ids + distances.
//
// Arithmetic and order are the canonical ones: every distance is one k-ordered fmaf chain, L2 in the
// expanded form with the stored row norms, selection under (key, id) -- so the answers are bit-identical to the batch
// pipeline's and to the oracle's.
#include "qk_internal.h"

#include <map>
#include "qk_device.h"

#include <algorithm>

struct SmallParams {
    // parent (flat) index: one list of centroids
    const float4 *c_vecs;
    const float *c_norms;
    const int64_t *c_ids;
    int64_t c_row0;
    int c_n;
    // partitions
    const float4 *vecs;
    const float *norms;
    const int64_t *ids;
    const int64_t *pt_off;
    const int32_t *pt_size;
    int npids;
    int nblk;
    int d;
    const float *x;  // [Q][d] row-major
    int nprobe, k, metric, sqrt_l2;
    int W;       // workgroups per query
    int cap;     // LDS candidate pool capacity (multiple of 64, > k)
    int64_t *out_ids;
    float *out_dist;
    // workspace: records [Q][W][k] (ord, id), counts [Q][W], tickets [Q] (zero between calls: the last arriver resets its own)
    uint32_t *rec_ord;
    int64_t *rec_id;
    int32_t *rec_cnt;
    unsigned int *ticket;
    // split coarse step (batches of several queries): workgroup (q, w) computes the keys of its 1/W of the centroids, the W
    // slices meet in ckeys [Q][c_n] behind a per-query arrival counter (all workgroups of the launch are co-resident: the host
    // sizes the grid by the occupancy of this kernel)
    int split;
    uint32_t *ckeys;
    unsigned int *arrive;  // [Q], zero between calls (reset with the ticket)
    long long *clock;  // probe (QK_SMALL_CLOCK): phase stamps of workgroup 0 in 100 MHz ticks, or nullptr
};

// canonical key of arena row `row` against the query staged in LDS (sq, zero-padded to 16 columns): one k-ordered fmaf chain
// over the tile-major layout (qk_internal.h): float4 (tile*nblk + c)*64 + g*16 + r holds columns 16c + {g, 4+g, 8+g, 12+g}
__device__ __forceinline__ uint32_t small_row_key(const float4 *vecs, const float *norms, int nblk, int64_t row, const float *sq,
                                                  float xn, bool l2) {
    const int64_t tile = row >> 4;
    const int r = (int)(row & 15);
    const float4 *base = vecs + tile * nblk * 64 + r;
    const float yn = l2 ? norms[row] : 0.0f;  // requested with the row data, not after the chain
    float acc = 0.0f;
    for (int c0 = 0; c0 < nblk; c0 += 8) {
        float4 v[8][4];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int cc = min(c0 + c, nblk - 1);
#pragma unroll
            for (int g = 0; g < 4; g++) v[c][g] = base[cc * 64 + g * 16];
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
            if (c0 + c < nblk) {
                const float4 *q4 = (const float4 *)(sq + (c0 + c) * 16);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float4 qq = q4[t];  // columns 16c + 4t .. 4t+3 (same address in every lane: LDS broadcast)
                    const float e0 = t == 0 ? v[c][0].x : t == 1 ? v[c][0].y : t == 2 ? v[c][0].z : v[c][0].w;
                    const float e1 = t == 0 ? v[c][1].x : t == 1 ? v[c][1].y : t == 2 ? v[c][1].z : v[c][1].w;
                    const float e2 = t == 0 ? v[c][2].x : t == 1 ? v[c][2].y : t == 2 ? v[c][2].z : v[c][2].w;
                    const float e3 = t == 0 ? v[c][3].x : t == 1 ? v[c][3].y : t == 2 ? v[c][3].z : v[c][3].w;
                    acc = __fmaf_rn(e0, qq.x, acc);
                    acc = __fmaf_rn(e1, qq.y, acc);
                    acc = __fmaf_rn(e2, qq.z, acc);
                    acc = __fmaf_rn(e3, qq.w, acc);
                }
            }
        }
    }
    return l2 ? ord_from_l2(l2_expanded(xn, yn, acc)) : ord_from_ip(acc);
}

// two rows at once (d <= 128 per pass: both rows' 32 float4 are requested before either chain starts)
__device__ __forceinline__ void small_row_key2(const float4 *vecs, const float *norms, int nblk, int64_t rowa, int64_t rowb, const float *sq,
                                               float xn, bool l2, uint32_t &ka, uint32_t &kb) {
    const float4 *ba = vecs + (rowa >> 4) * nblk * 64 + (int)(rowa & 15);
    const float4 *bb = vecs + (rowb >> 4) * nblk * 64 + (int)(rowb & 15);
    const float yna = l2 ? norms[rowa] : 0.0f, ynb = l2 ? norms[rowb] : 0.0f;
    float acca = 0.0f, accb = 0.0f;
    for (int c0 = 0; c0 < nblk; c0 += 4) {
        float4 va[4][4], vb[4][4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int cc = min(c0 + c, nblk - 1);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                va[c][g] = ba[cc * 64 + g * 16];
                vb[c][g] = bb[cc * 64 + g * 16];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (c0 + c < nblk) {
                const float4 *q4 = (const float4 *)(sq + (c0 + c) * 16);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float4 qq = q4[t];
                    const float qv[4] = {qq.x, qq.y, qq.z, qq.w};
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const float ea = t == 0 ? va[c][g].x : t == 1 ? va[c][g].y : t == 2 ? va[c][g].z : va[c][g].w;
                        const float eb = t == 0 ? vb[c][g].x : t == 1 ? vb[c][g].y : t == 2 ? vb[c][g].z : vb[c][g].w;
                        acca = __fmaf_rn(ea, qv[g], acca);
                        accb = __fmaf_rn(eb, qv[g], accb);
                    }
                }
            }
        }
    }
    ka = l2 ? ord_from_l2(l2_expanded(xn, yna, acca)) : ord_from_ip(acca);
    kb = l2 ? ord_from_l2(l2_expanded(xn, ynb, accb)) : ord_from_ip(accb);
}

// block-wide sum of one int per thread (256 threads)
__device__ __forceinline__ int small_block_sum(int v, int *s_red /*[4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

constexpr int QK_SMALL_THREADS = 512;  // 8 waves: two per SIMD
constexpr int QK_SMALL_NW = QK_SMALL_THREADS / 64;

template <int MAXCH>
__device__ __forceinline__ int small_select_t(uint32_t *ord, int64_t *id, int n, int k, int lane) {
    uint32_t kth;
    return select_pool<MAXCH>(ord, id, n, k, lane, kth);
}
__device__ __forceinline__ int small_select(uint32_t *ord, int64_t *id, int n, int k, int lane, int span) {
    if (span <= 64) return small_select_t<1>(ord, id, n, k, lane);
    if (span <= 128) return small_select_t<2>(ord, id, n, k, lane);
    if (span <= 256) return small_select_t<4>(ord, id, n, k, lane);
    if (span <= 512) return small_select_t<8>(ord, id, n, k, lane);
    return small_select_t<16>(ord, id, n, k, lane);
}
__device__ __forceinline__ int small_sort(uint32_t *ord, int64_t *id, int n, int k, int lane) {
    if (n <= 64) return compact_pool<1>(ord, id, n, k, lane);
    if (n <= 128) return compact_pool<2>(ord, id, n, k, lane);
    if (n <= 256) return compact_pool<4>(ord, id, n, k, lane);
    return compact_pool<8>(ord, id, n, k, lane);
}

// The k best of the n entries (ord, id) in LDS, sorted under (key, id), left in [0, min(n, k)).  A single wave selecting among
// hundreds of entries is a long serial chain of ballots (measured 6-10 us per use at 2.4 GHz); here every wave of the
// workgroup selects inside its own chunk (<= 64 entries where possible: 32 ballots) and wave 0 merges the <= 8 k survivors.
// Exact: selection and sort are select_pool / compact_pool (ties on the cut by id).  n <= 8192, k <= 64.  Uniform return value;
// contains barriers.
__device__ __forceinline__ int block_topk(uint32_t *ord, int64_t *id, int n, int k, uint32_t *tmp_ord, int64_t *tmp_id, int *s_got) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = max(1, min(QK_SMALL_NW, (n + 63) >> 6));
    const int per = (n + nchunks - 1) / nchunks;
    int got = 0;
    if (wave < nchunks) {
        const int b0 = wave * per, nw = max(0, min(per, n - b0));
        if (nw > 0) got = small_select(ord + b0, id + b0, nw, k, lane, per);
    }
    if (lane == 0) s_got[wave] = got;
    __syncthreads();
    if (wave == 0) {
        int m = 0;
        for (int c = 0; c < nchunks; c++) {
            const int g = s_got[c];
            for (int e = lane; e < g; e += 64) {
                tmp_ord[m + e] = ord[c * per + e];
                tmp_id[m + e] = id[c * per + e];
            }
            m += g;
        }
        const int nn = small_sort(tmp_ord, tmp_id, m, k, lane);  // m <= 8 k <= 512
        for (int e = lane; e < nn; e += 64) {
            ord[e] = tmp_ord[e];
            id[e] = tmp_id[e];
        }
        if (lane == 0) s_got[0] = nn;
    }
    __syncthreads();
    const int res = s_got[0];
    __syncthreads();
    return res;
}

constexpr int QK_SMALL_MAXQ = 256;   // queries (workspace); the default envelope is QK_SMALL_MAXQ_DEFAULT
constexpr int QK_SMALL_MAXQ_DEFAULT = 32;  // measured (1M x 128, nlist 1024, nprobe 10; device us, this kernel vs the batch pipeline):
                                           // 1: 40 / -, 8: 51 / 123, 16: 66 / 126, 32: 98 / 134, 64: 153 / 144
constexpr int QK_SMALL_MAXQW = 4096; // workgroups of one launch (records in the workspace)
constexpr int QK_SMALL_MAXP = 64;    // nprobe
constexpr int QK_SMALL_CPT = 16;     // centroids per thread (256 threads): nlist <= 4096
constexpr int QK_SMALL_MAXK = 32;

__global__ __launch_bounds__(QK_SMALL_THREADS) void k_search_small(SmallParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_red[4];
    __shared__ int64_t s_pid[QK_SMALL_MAXP];      // the probe list (rank order) ...
    __shared__ uint32_t s_pord[QK_SMALL_MAXP];
    __shared__ int s_np;
    __shared__ long long s_pre[QK_SMALL_MAXP + 1];  // ... and the prefix sums of the probed partitions' sizes
    __shared__ int64_t s_off[QK_SMALL_MAXP];
    __shared__ int s_last;
    __shared__ uint32_t s_kth;
    __shared__ int s_rcnt[65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x / P.W, w = blockIdx.x % P.W;
    const bool l2 = P.metric == QK_METRIC_L2;
    const int dpad = P.nblk * 16;
    float *sq = (float *)smem;                                  // [dpad] query, zero padded
    int64_t *pool_id = (int64_t *)(smem + (size_t)dpad * 4);    // [cap]
    uint32_t *pool_ord = (uint32_t *)(pool_id + P.cap);         // [cap]
    uint32_t *s_ck = pool_ord + P.cap;                          // [c_n] coarse keys ...
    int64_t *s_cid = (int64_t *)(s_ck + ((P.c_n + 7) & ~7));    // [c_n] ... and the partition ids that go with them
    __shared__ uint32_t s_mord[QK_SMALL_NW * QK_SMALL_MAXP];  // block_topk scratch
    __shared__ int64_t s_mid[QK_SMALL_NW * QK_SMALL_MAXP];
    __shared__ int s_got[QK_SMALL_NW];

    const long long cy0 = P.clock ? clock64() : 0;
    if (P.clock && tid == 0 && blockIdx.x == 0) P.clock[0] = wall_clock64();
    // ---- the query -----------------------------------------------------------------------------------------------------
    for (int c = tid; c < dpad; c += QK_SMALL_THREADS) sq[c] = c < P.d ? P.x[(int64_t)q * P.d + c] : 0.0f;
    __syncthreads();
    float xn = 0.0f;
    if (l2) {  // canonical |x|^2, every thread for itself: broadcast ds_read_b128, four columns per read (padding columns are 0)
        const float4 *sq4 = (const float4 *)sq;
#pragma unroll 8
        for (int c = 0; c < dpad / 4; c++) {
            const float4 v = sq4[c];
            xn = __fmaf_rn(v.x, v.x, xn);
            xn = __fmaf_rn(v.y, v.y, xn);
            xn = __fmaf_rn(v.z, v.z, xn);
            xn = __fmaf_rn(v.w, v.w, xn);
        }
    }

    // ---- A. coarse: keys of this thread's centroids (kept in LDS), then the nprobe smallest under (key, id) -------------------
    // (two centroid rows per thread in flight: the chain of one hides the load latency of the other)
    if (!P.split) {
        // every workgroup of the query computes all the keys: no exchange, W x the L2 reads (fine for one or two queries)
        for (int r = tid; r < P.c_n; r += 2 * QK_SMALL_THREADS) {
            const int r2 = r + QK_SMALL_THREADS;
            uint32_t k1, k2 = 0;
            if (r2 < P.c_n) {
                small_row_key2(P.c_vecs, P.c_norms, P.nblk, P.c_row0 + r, P.c_row0 + r2, sq, xn, l2, k1, k2);
                s_ck[r2] = k2;
            } else {
                k1 = small_row_key(P.c_vecs, P.c_norms, P.nblk, P.c_row0 + r, sq, xn, l2);
            }
            s_ck[r] = k1;
        }
        __syncthreads();
    } else {
        // this workgroup's 1/W of the centroids -> ckeys[q]; arrival counter; then everybody reads the whole key row.
        // (redundant keys cost Q x W x c_n x d x 4 bytes of L2 reads: 8 queries x 52 workgroups x 512 KB = 213 MB, 72 us)
        const int r0 = (int)(((long long)P.c_n * w) / P.W), r1 = (int)(((long long)P.c_n * (w + 1)) / P.W);
        uint32_t *krow = P.ckeys + (int64_t)q * P.c_n;
        for (int r = r0 + tid; r < r1; r += 2 * QK_SMALL_THREADS) {
            const int r2 = r + QK_SMALL_THREADS;
            uint32_t k1, k2 = 0;
            if (r2 < r1) {
                small_row_key2(P.c_vecs, P.c_norms, P.nblk, P.c_row0 + r, P.c_row0 + r2, sq, xn, l2, k1, k2);
                __hip_atomic_store(&krow[r2], k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                k1 = small_row_key(P.c_vecs, P.c_norms, P.nblk, P.c_row0 + r, sq, xn, l2);
            }
            __hip_atomic_store(&krow[r], k1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the exchange runs on agent-scope atomics end to end (keys, counter, reads): no L2 write-back / invalidate.  (A release
        // fence here, executed by all 8 waves of every workgroup, is a buffer_wbl2 each: the wait grew to 30-40 us.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            atomicAdd(&P.arrive[q], 1u);
            while (__hip_atomic_load(&P.arrive[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)P.W) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        for (int r = tid; r < P.c_n; r += QK_SMALL_THREADS)
            s_ck[r] = __hip_atomic_load(&krow[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    if (P.clock && tid == 0 && blockIdx.x == 0) P.clock[1] = wall_clock64();
    const int want = min(P.nprobe, P.c_n);
    // the `want` nearest under (key, partition id), in rank order
    for (int r = tid; r < P.c_n; r += QK_SMALL_THREADS) s_cid[r] = P.c_ids[P.c_row0 + r];
    __syncthreads();
    {
        const int nn = block_topk(s_ck, s_cid, P.c_n, want, s_mord, s_mid, s_got);
        for (int e = tid; e < nn; e += QK_SMALL_THREADS) {
            s_pord[e] = s_ck[e];
            s_pid[e] = s_cid[e];
        }
        if (tid == 0) s_np = nn;
    }
    if (P.clock && tid == 0 && blockIdx.x == 0) P.clock[2] = wall_clock64();
    __syncthreads();
    const int np = min(s_np, QK_SMALL_MAXP);
    // ---- B. this workgroup's slice of the probed rows -----------------------------------------------------------------------
    if (tid < 64) {  // wave 0: sizes and offsets of the np <= 64 probed partitions in ONE round trip, then a wave prefix sum
        int sz = 0;
        int64_t off = 0;
        if (tid < np) {
            const int64_t p = s_pid[tid];
            if (p >= 0 && p < P.npids) {
                sz = max(P.pt_size[p], 0);  // absent lists have size -1
                off = P.pt_off[p];
            }
        }
        long long inc = sz;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        if (tid < np) {
            s_pre[tid] = inc - sz;
            s_off[tid] = sz > 0 ? off : 0;
        }
        if (tid == max(np, 1) - 1) s_pre[np] = np > 0 ? inc : 0;
    }
    __syncthreads();
    const long long R = s_pre[np];
    const long long lo = (R * w) / P.W, hi = (R * (w + 1)) / P.W;
    const int k = P.k, cap = P.cap;
    int cnt = 0;              // entries held in the pool (uniform)
    uint32_t tau = 0xFFFFFFFFu;
    const int per_round = cap - k;  // multiple of 64 is not required: rows are appended with a ballot prefix per wave
    __shared__ int s_fill;
    for (long long base = lo; base < hi; base += per_round) {
        const long long end = min(hi, base + per_round);
        if (tid == 0) s_fill = cnt;
        __syncthreads();
        const int iters = (int)((end - base + QK_SMALL_THREADS - 1) / QK_SMALL_THREADS);
        for (int it = 0; it < iters; it++) {
            const long long g = base + tid + QK_SMALL_THREADS * (long long)it;
            bool pass = false;
            uint32_t key = 0xFFFFFFFFu;
            int64_t id = -1;
            if (g < end) {
                int pi = 0;
                while (pi + 1 < np && s_pre[pi + 1] <= g) pi++;
                const int64_t row = s_off[pi] + (g - s_pre[pi]);
                id = P.ids[row];  // (requested before the row data: one round trip, not two)
                key = small_row_key(P.vecs, P.norms, P.nblk, row, sq, xn, l2);
                pass = key != 0xFFFFFFFFu && key <= tau;
            }
            const uint64_t m = __ballot(pass);
            if (m) {
                int at = 0;
                if (lane == 0) at = atomicAdd(&s_fill, __popcll(m));
                at = __builtin_amdgcn_readfirstlane(at);
                if (pass) {
                    const int sl = at + __popcll(m & ((1ull << lane) - 1ull));
                    pool_ord[sl] = key;
                    pool_id[sl] = id;
                }
            }
        }
        __syncthreads();
        cnt = s_fill;
        __syncthreads();
        if (cnt > k || end >= hi) {  // keep the k best, sorted; their k-th key bounds the next round
            cnt = block_topk(pool_ord, pool_id, cnt, k, s_mord, s_mid, s_got);
            if (cnt >= k) tau = min(tau, pool_ord[k - 1]);
            __syncthreads();
        }
    }
    if (lo >= hi) cnt = 0;
    if (P.clock && tid == 0 && blockIdx.x == 0) P.clock[3] = wall_clock64();
    // ---- C. record, ticket, merge by the last arriver -----------------------------------------------------------------------
    const int64_t rbase = ((int64_t)q * P.W + w) * k;
    for (int e = tid; e < cnt; e += QK_SMALL_THREADS) {
        P.rec_ord[rbase + e] = pool_ord[e];
        P.rec_id[rbase + e] = pool_id[e];
    }
    if (tid == 0) P.rec_cnt[q * P.W + w] = cnt;
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = atomicAdd(&P.ticket[q], 1u);
        s_last = t == (unsigned)(P.W - 1) ? 1 : 0;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            P.ticket[q] = 0;  // ready for the next call on this stream
            if (P.split) P.arrive[q] = 0;  // (every workgroup of the query is past the arrival wait: it took a ticket)
        }
    }
    __syncthreads();
    if (P.clock && tid == 0 && blockIdx.x == 0) {
        P.clock[4] = wall_clock64();
        P.clock[7] = clock64() - cy0;
    }
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (tid == 0) s_fill = 0;
    __syncthreads();
    // the W records: their counts in one round trip (threads 0..W-1), then every entry of every record in a second one
    if (tid < P.W) s_rcnt[tid] = __hip_atomic_load(&P.rec_cnt[q * P.W + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int ww = 0; ww < P.W; ww++) {
            const int c = s_rcnt[ww];
            s_rcnt[ww] = run;
            run += c;
        }
        s_rcnt[P.W] = run;
        s_fill = run;
    }
    __syncthreads();
    for (int e = tid; e < s_rcnt[P.W]; e += QK_SMALL_THREADS) {
        int ww = 0;
        while (ww + 1 < P.W && s_rcnt[ww + 1] <= e) ww++;
        const int64_t src = ((int64_t)q * P.W + ww) * k + (e - s_rcnt[ww]);
        pool_ord[e] = __hip_atomic_load(&P.rec_ord[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pool_id[e] = __hip_atomic_load(&P.rec_id[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int total = s_fill;  // <= W * k <= cap
    const int nn = block_topk(pool_ord, pool_id, total, k, s_mord, s_mid, s_got);
    if (wave == 0) {
        for (int e = lane; e < k; e += 64) {
            int64_t oid = -1;
            float od = l2 ? INFINITY : -INFINITY;
            if (e < nn) {
                oid = pool_id[e];
                const uint32_t o = pool_ord[e];
                if (l2) {
                    const float d2 = __uint_as_float(o);
                    od = P.sqrt_l2 ? sqrtf(d2) : d2;
                } else {
                    od = ip_from_ord(o);
                }
            }
            P.out_ids[(int64_t)q * k + e] = oid;
            if (P.out_dist) P.out_dist[(int64_t)q * k + e] = od;
        }
        if (P.clock && lane == 0) {
            P.clock[5] = wall_clock64();
            P.clock[6] = blockIdx.x;
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
// Supported envelope (qk_search routes everything else through the batch pipeline): flat parent with one list of at most
// 4096 centroids whose ids are >= 0, nprobe <= 64, k <= 32, Q <= 64 (QK_SMALL_MAXQ_DEFAULT).
// ---- coarse step of a mid-sized batch (32 < Q <= 256 against <= 4096 centroids) in ONE launch ------------------------------------
// The batch pipeline's coarse step is k_dense_ord + k_select_rows (key matrix written and read back, two launches sized for
// thousands of queries); for a few dozen queries it is ~25 us of mostly launch latency.  Here workgroup q computes the keys of
// ALL centroids for its query (the same canonical chains as k_search_small: the centroid arena is L2-resident) and selects the
// nprobe nearest with block_topk -- same keys, same (key, id) order, same output as the two-kernel form.
struct CoarseSmallParams {
    const float4 *c_vecs;
    const float *c_norms;
    const int64_t *c_ids;  // arena ids
    int64_t c_row0;
    int c_n;
    int nblk, d;
    const float *x;  // [Q][d] row-major
    int k;
    int metric, sqrt_l2;
    int64_t *out_ids;  // [Q][k]
    float *out_dist;   // [Q][k] or nullptr
};

__global__ __launch_bounds__(QK_SMALL_THREADS) void k_coarse_small(CoarseSmallParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_mord[QK_SMALL_NW * QK_SMALL_MAXP];
    __shared__ int64_t s_mid[QK_SMALL_NW * QK_SMALL_MAXP];
    __shared__ int s_got[QK_SMALL_NW];
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x;
    const bool l2 = P.metric == QK_METRIC_L2;
    const int dpad = P.nblk * 16;
    float *sq = (float *)smem;                                                  // [dpad]
    int64_t *s_cid = (int64_t *)(smem + (size_t)dpad * 4);                      // [c_n]
    uint32_t *s_ck = (uint32_t *)(s_cid + P.c_n);                               // [c_n]
    for (int c = tid; c < dpad; c += QK_SMALL_THREADS) sq[c] = c < P.d ? P.x[q * P.d + c] : 0.0f;
    for (int r = tid; r < P.c_n; r += QK_SMALL_THREADS) s_cid[r] = P.c_ids[P.c_row0 + r];
    __syncthreads();
    float xn = 0.0f;
    if (l2) {
        const float4 *sq4 = (const float4 *)sq;
#pragma unroll 8
        for (int c = 0; c < dpad / 4; c++) {
            const float4 v = sq4[c];
            xn = __fmaf_rn(v.x, v.x, xn);
            xn = __fmaf_rn(v.y, v.y, xn);
            xn = __fmaf_rn(v.z, v.z, xn);
            xn = __fmaf_rn(v.w, v.w, xn);
        }
    }
    for (int r = tid; r < P.c_n; r += 2 * QK_SMALL_THREADS) {
        const int r2 = r + QK_SMALL_THREADS;
        uint32_t k1, k2 = 0;
        if (r2 < P.c_n) {
            small_row_key2(P.c_vecs, P.c_norms, P.nblk, P.c_row0 + r, P.c_row0 + r2, sq, xn, l2, k1, k2);
            s_ck[r2] = k2;
        } else {
            k1 = small_row_key(P.c_vecs, P.c_norms, P.nblk, P.c_row0 + r, sq, xn, l2);
        }
        s_ck[r] = k1;
    }
    __syncthreads();
    const int nn = block_topk(s_ck, s_cid, P.c_n, min(P.k, P.c_n), s_mord, s_mid, s_got);
    for (int e = tid; e < P.k; e += QK_SMALL_THREADS) {
        int64_t oid = -1;
        float od = l2 ? INFINITY : -INFINITY;
        if (e < nn) {
            oid = s_cid[e];
            const uint32_t o = s_ck[e];
            if (l2) {
                const float d2 = __uint_as_float(o);
                od = P.sqrt_l2 ? sqrtf(d2) : d2;
            } else {
                od = ip_from_ord(o);
            }
        }
        P.out_ids[q * P.k + e] = oid;
        if (P.out_dist) P.out_dist[q * P.k + e] = od;
    }
}

bool qk_coarse_small_supported(const qk_store *s, int64_t Q, int nrows, int k) {
    static const int enabled = qk_env_int("QK_COARSE_SMALL", 1);
    if (!enabled || Q <= 0 || Q > 256 || k < 2 || k > QK_SMALL_MAXP || nrows <= 0 || nrows > 256 * QK_SMALL_CPT) return false;
    if (Q * (int64_t)nrows > 256 * 1024) return false;  // every workgroup reads the whole centroid arena from L2
    if ((size_t)s->dpad * 4 + (size_t)nrows * 12 + 8192 > 64 * 1024) return false;
    return true;
}

int qk_launch_coarse_small(qk_ctx *ctx, qk_store *s, int64_t row_off, int nrows, const float *x, int64_t Q, int k, int metric,
                           bool sqrt_l2, int64_t *out_ids, float *out_dist) {
    CoarseSmallParams P;
    P.c_vecs = (const float4 *)s->vecs;
    P.c_norms = s->norms;
    P.c_ids = s->ids;
    P.c_row0 = row_off;
    P.c_n = nrows;
    P.nblk = s->nblk;
    P.d = s->d;
    P.x = x;
    P.k = k;
    P.metric = metric;
    P.sqrt_l2 = sqrt_l2 ? 1 : 0;
    P.out_ids = out_ids;
    P.out_dist = out_dist;
    const size_t lds = (size_t)s->dpad * 4 + (size_t)nrows * 12 + 16;
    hipLaunchKernelGGL(k_coarse_small, dim3((unsigned)Q), dim3(QK_SMALL_THREADS), lds, ctx->stream, P);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

bool qk_small_supported(qk_ctx *ctx, qk_store *parent, qk_store *s, int64_t Q, int nprobe, int k) {
    static const int enabled = qk_env_int("QK_SMALL", 1);
    static const int max_q = qk_env_int("QK_SMALL_MAX_Q", QK_SMALL_MAXQ_DEFAULT);
    if (!enabled || !parent || Q <= 0 || Q > std::min(QK_SMALL_MAXQ, max_q)) return false;
    if (k > QK_SMALL_MAXK || nprobe > QK_SMALL_MAXP || nprobe <= 0) return false;
    if (parent->nlist != 1 || parent->ntotal <= 0 || parent->ntotal > 256 * QK_SMALL_CPT) return false;  // <= 4096 centroids
    if (parent->min_id_seen < 0 || parent->d != s->d) return false;
    if ((size_t)s->dpad * 4 + 1024 * 12 + (size_t)parent->ntotal * 12 + 8192 > 160 * 1024) return false;
    // How much there is to scan decides too: a query's slices are at most 64 workgroups and every workgroup ranks all the centroids
    // itself.  10M x 128 in 4096 lists (2441 rows each), ms per call, this kernel / the batch pipeline: 1 query nprobe 10 0.065 /
    // 0.089, 32 0.115 / 0.088, 64 0.208 / 0.107; 4 queries 0.074 / 0.088, 0.131 / 0.124, 0.228 / 0.168; 16 queries 0.137 / 0.131,
    // 0.329 / 0.207, 0.664 / 0.269; 32 queries 0.223 / 0.167, 0.598 / 0.251, 1.258 / 0.329.  On the configs[0] shape (977 rows per
    // list, nprobe 10) it wins up to 32 queries (98 / 134 us).  So: at most 32k rows per query and 320k rows per call.
    static const int64_t rows_q_max = qk_env_int("QK_SMALL_ROWS_PER_QUERY", 32768), rows_max = qk_env_int("QK_SMALL_ROWS", 327680);
    const int64_t mean_rows = std::max<int64_t>(1, s->ntotal / std::max<int64_t>(1, s->n_nonempty > 0 ? s->n_nonempty : s->nlist));
    const int64_t rows_q = std::min<int64_t>(nprobe, parent->ntotal) * mean_rows;
    if (rows_q > rows_q_max || rows_q * Q > rows_max) return false;
    return true;
}

int qk_search_small_device(qk_ctx *ctx, qk_store *parent, qk_store *s, const float *x, int64_t Q, int nprobe, int k, int metric,
                           int64_t *out_ids, float *out_dist, bool sqrt_l2) {
    QK_TRY(qk_store_sync_table(s));
    int64_t c_row0 = 0, c_n = 0;
    for (const qk_part &pt : parent->parts)
        if (pt.present) {
            c_row0 = pt.row_off;
            c_n = pt.size;
            break;
        }
    // workgroups per query: enough that a slice is a few hundred rows, no more than fill the chip a few times over
    const int64_t rows_est = std::max<int64_t>(1, (int64_t)std::min<int64_t>(nprobe, c_n) * std::max<int64_t>(1, s->ntotal / std::max<int64_t>(1, s->nlist)));
    static const int w_rows = std::max(16, qk_env_int("QK_SMALL_W_ROWS", 192)), w_max = std::max(1, qk_env_int("QK_SMALL_W_MAX", 64));  // (probe)
    int W = (int)std::min<int64_t>(w_max, std::max<int64_t>(1, rows_est / w_rows));
    W = std::min(W, std::max(1, 1024 / k));
    W = std::min<int64_t>(W, std::max<int64_t>(1, 2048 / Q));
    const int cap = 1024;
    const size_t lds = (size_t)s->dpad * 4 + (size_t)cap * 12 + (size_t)((c_n + 7) & ~7) * 12 + 64;
    if (lds > 48 * 1024) QK_HIP(hipFuncSetAttribute((const void *)k_search_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // split coarse step: the workgroups of a query wait for each other, so the whole grid has to be resident at once
    static const int split_env = qk_env_int("QK_SMALL_SPLIT", -1);  // -1 auto, 0 never, 1 always
    static const int split_min_wgs = qk_env_int("QK_SMALL_SPLIT_MIN", 96);
    int resident = 0;
    {
        // workgroups per CU by LDS size (the only launch parameter occupancy depends on), cached in the CONTEXT: one store may
        // be searched through several contexts at once (include/quake_hip.h), on devices with different CU counts
        auto it = ctx->small_occ.find(lds);
        if (it == ctx->small_occ.end()) {
            int per_cu = 0;
            QK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_search_small, QK_SMALL_THREADS, lds));
            it = ctx->small_occ.emplace(lds, per_cu).first;
        }
        resident = it->second * std::max(1, ctx->prop.multiProcessorCount);
    }
    bool split = split_env == 1 || (split_env < 0 && Q * (int64_t)W >= split_min_wgs);
    if (split) {
        if (resident < Q) {
            split = false;  // cannot even hold one workgroup per query: redundant keys
        } else {
            W = (int)std::min<int64_t>(W, resident / Q);
        }
    }
    if (!split) W = std::min<int64_t>(W, std::max<int64_t>(1, QK_SMALL_MAXQW / Q));
    // persistent workspace (tickets and arrival counters must stay zero between calls)
    const size_t sz_rec_id = (size_t)QK_SMALL_MAXQW * QK_SMALL_MAXK * 8, sz_rec_ord = (size_t)QK_SMALL_MAXQW * QK_SMALL_MAXK * 4;
    const size_t sz_cnt = (size_t)QK_SMALL_MAXQW * 4, sz_tick = (size_t)QK_SMALL_MAXQ * 4;
    const size_t sz_keys = (size_t)QK_SMALL_MAXQ * 256 * QK_SMALL_CPT * 4;
    const size_t need = sz_rec_id + sz_rec_ord + sz_cnt + 2 * sz_tick + sz_keys + 1024;
    if (!ctx->small_ws) {
        QK_HIP(hipMalloc((void **)&ctx->small_ws, need));
        QK_HIP(hipMemsetAsync(ctx->small_ws, 0, need, ctx->stream));
    }
    SmallParams P;
    P.c_vecs = (const float4 *)parent->vecs;
    P.c_norms = parent->norms;
    P.c_ids = parent->ids;
    P.c_row0 = c_row0;
    P.c_n = (int)c_n;
    P.vecs = (const float4 *)s->vecs;
    P.norms = s->norms;
    P.ids = s->ids;
    P.pt_off = s->d_off;
    P.pt_size = s->d_size;
    P.npids = (int)s->parts.size();
    P.nblk = s->nblk;
    P.d = s->d;
    P.x = x;
    P.nprobe = nprobe;
    P.k = k;
    P.metric = metric;
    P.sqrt_l2 = sqrt_l2 ? 1 : 0;
    P.W = W;
    P.cap = cap;
    P.out_ids = out_ids;
    P.out_dist = out_dist;
    char *b = ctx->small_ws;
    P.rec_id = (int64_t *)b;
    b += sz_rec_id;
    P.rec_ord = (uint32_t *)b;
    b += sz_rec_ord;
    P.rec_cnt = (int32_t *)b;
    b += sz_cnt;
    P.ticket = (unsigned int *)b;
    b += sz_tick;
    P.arrive = (unsigned int *)b;
    b += sz_tick;
    P.ckeys = (uint32_t *)b;
    P.split = split ? 1 : 0;
    static const bool probe_clock = qk_env_set("QK_SMALL_CLOCK");
    static long long *d_clock = nullptr;
    P.clock = nullptr;
    if (probe_clock) {
        if (!d_clock) QK_HIP(hipMalloc((void **)&d_clock, 64));
        QK_HIP(hipMemsetAsync(d_clock, 0, 64, ctx->stream));
        P.clock = d_clock;
    }
    hipLaunchKernelGGL(k_search_small, dim3((unsigned)(Q * W)), dim3(QK_SMALL_THREADS), lds, ctx->stream, P);
    QK_HIP(hipGetLastError());
    if (probe_clock) {
        long long h[8];
        QK_HIP(hipMemcpyAsync(h, d_clock, 64, hipMemcpyDeviceToHost, ctx->stream));
        QK_HIP(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[k_search_small] Q=%lld W=%d: workgroup 0 (10 ns ticks): query+keys %lld, select %lld, scan %lld, record+ticket %lld; "
                "last workgroup %lld ended %lld after workgroup 0 started; shader clock %.2f GHz\n", (long long)Q, W, h[1] - h[0], h[2] - h[1], h[3] - h[2],
                h[4] - h[3], h[6], h[5] - h[0], h[4] > h[0] ? (double)h[7] / (double)(h[4] - h[0]) * 0.1 : 0.0);
    }
    return QK_OK;
}

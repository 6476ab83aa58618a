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
//   k_group_count / k_group_scan / k_group_scatter
//                    (q,p) pairs -> per-partition query groups; the work is the sequence of "items"
//                    (partition p, 16-query tile qt), each ntiles(p) row tiles long, laid end to end in tile units
//   k_scan<DB,MAXCH> persistent, ONE WAVE PER WORKGROUP.  The tile sequence is cut into equal contiguous ranges, one
//                    per wave (perfect byte balance whatever the partition sizes).  A wave walks its range segment by
//                    segment (a segment = part of one item): stages the 16 queries in LDS in B-operand lane order,
//                    streams the rows as contiguous 1 KiB float4 loads straight into MFMA A operands (register
//                    double-buffered), v_mfma_f32_16x16x4_f32 accumulates dot products in natural k order; candidates
//                    that beat the running k-th best go to a per-query LDS pool compacted by rank (the TopkBuffer
//                    append + flush); at the segment end each non-empty pool becomes a record chained to its
//                    (query, partition) pair.  A per-query bound shared through global memory (gtau) lets later
//                    segments skip almost everything.
//   k_merge<MAXCH>   one wave per query walks its pairs' record chains, merges, applies sqrt / padding -> [Q][k]
//
// Roofline: HBM.  Algorithmic bytes per batch = sum over unique probed partitions of n_p*d*4.
#include "qk_internal.h"
#include <vector>
#include <climits>
#include "qk_device.h"
#include "qk_scan_types.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <map>

// ---- query preparation -----------------------------------------------------------------------------
// xq4[(q*nblk + c)*4 + g] = {x[q][16c+g], x[q][16c+g+4], x[q][16c+g+8], x[q][16c+g+12]}  (B-operand order)
// xn[q] = canonical squared norm (k-ordered fmaf chain).  One workgroup = 16 queries staged through LDS so that the
// global reads are coalesced and the 16 serial norm chains run out of LDS.
__global__ __launch_bounds__(256) void k_prep_queries(const float *__restrict__ x, int64_t Q, int d, int nblk,
                                                      float4 *__restrict__ xq4, float *__restrict__ xn,
                                                      unsigned long long *__restrict__ best64, int qpw,
                                                      float4 *__restrict__ xp4, uint4 *__restrict__ zero16, int64_t n_zero16) {
    extern __shared__ float sq[];  // [qpw][d+1]
    // the scan's per-call state (counters, cursors, bounds) starts from zero: cleared here, by everybody, instead of a memset
    // launch between the coarse step and the scan (6 us of the 0.31 ms bench step)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_zero16; i += (int64_t)gridDim.x * 256)
        zero16[i] = make_uint4(0u, 0u, 0u, 0u);
    const int ldq = d + 1;
    const int64_t q0 = (int64_t)blockIdx.x * qpw;
    const int nq = (int)min((int64_t)qpw, Q - q0);
    for (int i = threadIdx.x; i < nq * d; i += 256) {
        int r = i / d, c = i - r * d;
        sq[r * ldq + c] = x[(q0 + r) * d + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nq * nblk * 4; i += 256) {
        int r = i / (nblk * 4), rem = i - r * nblk * 4;
        int c = rem >> 2, g = rem & 3;
        int col = 16 * c + g;
        const float *s = sq + r * ldq;
        float4 v;
        v.x = col < d ? s[col] : 0.0f;
        v.y = col + 4 < d ? s[col + 4] : 0.0f;
        v.z = col + 8 < d ? s[col + 8] : 0.0f;
        v.w = col + 12 < d ? s[col + 12] : 0.0f;
        xq4[(q0 + r) * nblk * 4 + rem] = v;
        // row-major copy padded to 16 columns (natural column order, four columns per float4): the B operand source of the
        // row-per-lane scan (qk_scan_rl.hip)
        const int c4 = 4 * rem;
        float4 w;
        w.x = c4 < d ? s[c4] : 0.0f;
        w.y = c4 + 1 < d ? s[c4 + 1] : 0.0f;
        w.z = c4 + 2 < d ? s[c4 + 2] : 0.0f;
        w.w = c4 + 3 < d ? s[c4 + 3] : 0.0f;
        xp4[(q0 + r) * nblk * 4 + rem] = w;
    }
    if (threadIdx.x < nq) {
        const float *s = sq + threadIdx.x * ldq;
        float acc = 0.0f;
        for (int k = 0; k < d; k++) acc = __fmaf_rn(s[k], s[k], acc);
        xn[q0 + threadIdx.x] = acc;
        best64[q0 + threadIdx.x] = ~0ull;  // "nothing yet" for the nearest-centroid kernel (k_dense_argmin): saves its memset
    }
}

size_t qk_scan_zero_bytes(int64_t npids, int64_t Q) { return (size_t)npids * 4 * 2 + 256 + (size_t)Q * 4; }

// Lays the batch's prep buffer out (fragment-ordered copy | norms | two "nothing yet" arrays of the nearest-centroid kernel, used
// in turn | row-major padded copy | region cleared for the scan) and either launches k_prep_queries or -- defer -- leaves the launch
// PENDING (ctx->prep_pending): the nearest-centroid kernel of the same batch then does the preparation itself while it stages its
// queries (k_dense_argmin<.., FUSE>, qk_dense.hip: one launch and one kernel boundary less in front of a search), and every other
// consumer of the prepared queries calls qk_prep_flush first.
int qk_prep_queries(qk_ctx *ctx, const float *x, int64_t Q, int d, const float4 **xq4, const float **xn, size_t zero_bytes, bool defer) {
    const int dpad = qk_round_up(d, 16), nblk = dpad / 16;
    const size_t off_n = ((size_t)Q * dpad * 4 + 255) & ~(size_t)255;
    const size_t off_b = (off_n + (size_t)Q * 4 + 255) & ~(size_t)255;
    const size_t off_b1 = (off_b + (size_t)Q * 8 + 255) & ~(size_t)255;
    const size_t off_p = (off_b1 + (size_t)Q * 8 + 255) & ~(size_t)255;
    const size_t off_z = (off_p + (size_t)Q * dpad * 4 + 255) & ~(size_t)255;
    const size_t zero_pad = (zero_bytes + 15) & ~(size_t)15;
    size_t need = off_z + zero_pad + 256;
    if (need > ctx->qprep_cap) {
        QK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->qprep) QK_HIP(hipFree(ctx->qprep));
        ctx->qprep = nullptr;
        ctx->qprep_cap = 0;
        ctx->scratch_reallocs++;
        if (hipMalloc((void **)&ctx->qprep, need + need / 4) != hipSuccess) QK_FAIL(QK_ERR_OOM, "query prep buffer allocation failed");
        ctx->qprep_cap = need + need / 4;
    }
    // the two "nothing yet" arrays sit where THIS batch size puts them: entries known to hold ~0 survive only an unchanged layout
    if (ctx->qprep_layout_Q != Q || ctx->qprep_layout_d != d || ctx->qprep_layout_base != ctx->qprep) {
        ctx->best64_clean[0] = ctx->best64_clean[1] = 0;
        ctx->qprep_layout_Q = Q;
        ctx->qprep_layout_d = d;
        ctx->qprep_layout_base = ctx->qprep;
    }
    ctx->best64_cur ^= 1;  // this batch's array; the other one was this array's predecessor
    ctx->best64_buf[0] = (unsigned long long *)(ctx->qprep + off_b);
    ctx->best64_buf[1] = (unsigned long long *)(ctx->qprep + off_b1);
    float4 *q4 = (float4 *)ctx->qprep;
    float *n = (float *)(ctx->qprep + off_n);
    ctx->qprep_best64 = ctx->best64_buf[ctx->best64_cur];
    ctx->qprep_xp4 = (const float4 *)(ctx->qprep + off_p);
    ctx->qprep_zero = zero_bytes ? ctx->qprep + off_z : nullptr;
    ctx->qprep_zero_bytes = zero_bytes;  // the first qk_scan_device of this batch that needs exactly this much consumes it
    ctx->prep_pending = false;
    ctx->prep_x = x;
    ctx->prep_Q = Q;
    ctx->prep_d = d;
    ctx->prep_zero16 = (int64_t)(zero_pad / 16);
    *xq4 = q4;
    *xn = n;
    if (defer) {
        ctx->prep_pending = true;
        ctx->qprep_best64_n = ctx->best64_clean[ctx->best64_cur] >= Q ? Q : 0;  // (a fused launch clears what is not clean itself)
        return QK_OK;
    }
    return qk_prep_flush(ctx, true);
}

// launches the plain prep kernel of the batch laid out last (force: whether or not it is pending)
int qk_prep_flush(qk_ctx *ctx, bool force) {
    if (!ctx->prep_pending && !force) return QK_OK;
    ctx->prep_pending = false;
    const int64_t Q = ctx->prep_Q;
    const int d = ctx->prep_d;
    const int dpad = qk_round_up(d, 16), nblk = dpad / 16;
    const size_t off_n = ((size_t)Q * dpad * 4 + 255) & ~(size_t)255;
    // queries per workgroup: 16, fewer when that would leave most of the chip idle (1024 x 768: 64 workgroups took 16 us)
    int qpw = 16;
    while (qpw > 2 && (Q + qpw - 1) / qpw < 2 * (int64_t)std::max(1, ctx->prop.multiProcessorCount) && (int64_t)qpw * d > 1024) qpw >>= 1;
    const size_t lds = (size_t)qpw * (d + 1) * 4;
    if (lds > 160 * 1024) QK_FAIL(QK_ERR_UNSUPPORTED, "d=%d too large for the query prep kernel", d);
    if (lds > 48 * 1024)
        QK_HIP(hipFuncSetAttribute((const void *)k_prep_queries, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_prep_queries, dim3((unsigned)((Q + qpw - 1) / qpw)), dim3(256), lds, ctx->stream, ctx->prep_x, Q, d, nblk,
                       (float4 *)ctx->qprep, (float *)(ctx->qprep + off_n), ctx->qprep_best64, qpw, (float4 *)ctx->qprep_xp4,
                       (uint4 *)ctx->qprep_zero, ctx->qprep_zero ? ctx->prep_zero16 : (int64_t)0);
    QK_HIP(hipGetLastError());
    ctx->best64_clean[ctx->best64_cur] = Q;  // the kernel writes ~0 for every query of the batch
    ctx->qprep_best64_n = Q;                 // initialised for Q queries; the first nearest-centroid launch consumes it
    return QK_OK;
}

struct GroupParams {
    const int64_t *pids;  // [Q*P] or nullptr (all_lists: pair i -> list i % P)
    const unsigned long long *pids_packed;  // [Q] (P = 1) or nullptr
    int64_t npairs;
    int P;
    const int32_t *pt_size;
    int npids;
    int32_t *g_cnt;       // [npids] queries probing each partition
    int32_t *g_cursor;    // [npids]
    int32_t *g_qoff;      // [npids+1] offsets into grouped_*
    int32_t *n_active;    // [1]
    ActiveInfo *active;   // [npids+1] partitions with >= 1 query, ascending; entry n_active is a sentinel (toff = total)
    const int64_t *pt_off;
    int64_t *n_tiles;     // [1] total tiles = sum over active p of ntiles(p) * ceil(cnt_p/16)
    int64_t *n_rows_unique;  // [1] sum of sizes of active partitions (algorithmic bytes / (d*4))
    int32_t *grouped_q;   // [npairs] query of each grouped entry
    int32_t *grouped_pair;// [npairs] pair index (q*P + r) of each grouped entry
    int32_t *act_list;    // [min(npids, npairs)] partitions with at least one probing query, in first-hit order (k_group_count)
    int32_t *n_act;       // its length (zeroed per call)
    int seg_ovh;          // sequence units charged for the start of a pass (see seq_weight)
    int qgroup;           // query tiles that share one pass over a partition (k_scan's query-sharing workgroups), >= 1
    int32_t *pair_head;   // [npairs] head of the record chain of each pair (-1 = none)
    int32_t *pair_slots;  // [npairs][32]: {record count, first 31 record ids} -- what the merge reads in ONE load; later
                          // records of the pair go to the chain
    uint32_t *gtau;       // [Q] per-query shared bound, reset here
    int rl;               // 1: the sequence is measured in the units of the row-per-lane scan (rl_part_len), else seq_weight
    RlCost rlc;
    int32_t *n_pairs_live; // [1] (query, partition) pairs that reach a present, non-empty partition (qk_timing::partitions_scanned)
    // mixed sequence (rl only): lists with cnt >= hot.min weigh nothing in the per-wave sequence and are cut into items instead
    HotCost hot;
    int32_t *act_hoff;     // [npids+1] hot items in front of every active list; entry n_active is the total
    int32_t *n_hot;        // [1]
    long long *hot_units;  // [1]
};

// length of a list in the per-wave sequence, and the hot items / hot cost it contributes instead when it is hot
__device__ __forceinline__ long long group_seq_len(const GroupParams &G, int c, int sz, int *items, long long *hunits) {
    *items = 0;
    *hunits = 0;
    if (!G.rl) return 0;  // (seq_weight: the caller)
    if (hot_list(c, sz, G.hot)) {
        const HotShape hs = hot_shape(c, sz, G.hot);
        *items = hs.nqblk * hs.nrr;
        *hunits = hot_units_of(c, sz, G.hot);
        return 0;
    }
    return rl_part_len(c, sz, G.rlc);
}

__device__ __forceinline__ int pair_pid(const GroupParams &G, int64_t i) {
    int64_t p;
    if (G.pids_packed) {  // nearest-list result of k_dense_argmin (qk_scan_args::pids_packed)
        const unsigned long long v = G.pids_packed[i];
        p = v == ~0ull ? -1 : (int64_t)(v & 0xFFFFFFFFull);
    } else {
        p = G.pids ? G.pids[i] : (i % G.P);
    }
    if (p < 0 || p >= G.npids) return -1;
    return G.pt_size[p] > 0 ? (int)p : -1;
}

// returns the partition of the pair (-1: none) and, in *pos, how many queries reached it before this one
__device__ __forceinline__ int group_count_one(const GroupParams &G, int64_t i, int *pos = nullptr) {
    G.pair_head[i] = -1;
    G.pair_slots[i * QK_SLOTS] = 0;
    int p = pair_pid(G, i);
    // the first query to reach a partition lists it: the scan below walks the probed partitions only, not all of them
    // (a rank of an N-GPU index sees N x 4096 list numbers, 1/N of the batch's queries land on its own)
    int old = 0;
    if (p >= 0) {
        old = atomicAdd(&G.g_cnt[p], 1);
        if (old == 0) G.act_list[atomicAdd(G.n_act, 1)] = p;
        if (pos) *pos = old;
    }
    // the pair's list and arrival rank wait in the (still unused) record slots of its slot line for the scatter pass, which then
    // needs neither the list numbers again nor a second round of atomics (three dependent round trips less)
    G.pair_slots[i * QK_SLOTS + 1] = p;
    G.pair_slots[i * QK_SLOTS + 2] = old;
    return p;
}

__global__ void k_group_count(GroupParams G) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G.npairs) group_count_one(G, i);
}

// block-wide exclusive scan of one value per thread (1024 threads = 16 waves); returns the exclusive prefix, *total
// receives the block sum.  wave-level shuffles + one LDS hop.
template <typename T>
__device__ __forceinline__ T block_exscan_1024(T v, T *s_wave /*[16]*/, T *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    T wpre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        T sv = s_wave[w];
        if (w < wave) wpre += sv;
        tot += sv;
    }
    __syncthreads();
    *total = tot;
    return wpre + inc - v;
}

// length of a partition in the work sequence.  G = 1: (query tiles) x (row tiles).  G > 1 (query-sharing workgroups): a pass
// over the row tiles serves up to G query tiles; a pass with q' = 1, 2 or 4 (rounded up) query tiles keeps q' of the G waves
// busy per tile cut, so it weighs q' units per row tile -- the sequence is measured in workgroup time, which is what the
// static cut has to balance.
__device__ __forceinline__ long long seq_weight(int cnt_q, int size_p, int G, int ovh) {
    const int nqt = (cnt_q + 15) >> 4, ntl = (size_p + 15) >> 4;
    const int nfull = nqt / G, rem = nqt % G;
    const int rp = rem == 0 ? 0 : rem <= 1 ? 1 : rem <= 2 ? 2 : 4;
    // + ovh units in front of every pass: what starting a segment costs (query staging, cold start of the pools, record
    // emission).  Every wave of the workgroup pays it at the same time whatever the width of the pass, so it is a constant
    // number of units (steps x waves) -- a range that holds more pass starts gets fewer row tiles
    return (long long)ntl * ((long long)nfull * G + rp) + (long long)ovh * (nfull + (rem > 0 ? 1 : 0));
}

// single workgroup of 1024 threads: exclusive scans over the probed partitions
// SAME_KERNEL: the counters were written with atomics by this very workgroup (single-kernel form) and are read back with agent-
// scope loads, which go past the L2; behind a kernel boundary (three-kernel form) plain loads do, and hit in L2 -- the scan
// kernel is a chain of dependent loads and nothing else (4096 active lists: 16 round trips per thread)
// HASHED (the one-workgroup form, group_small_body): act_list / g_cnt / g_qoff are LDS tables indexed by hash slot, key_of[slot]
// is the list number
template <bool SAME_KERNEL, bool HASHED = false>
__device__ __forceinline__ void group_scan_body(const GroupParams &G, long long *s_w /*[64] LDS*/, const int *key_of = nullptr) {
    const int tid = threadIdx.x;
    // (agent-scope loads: in the single-kernel form the counters were just written with atomics by this workgroup)
    const int n_act = SAME_KERNEL ? __hip_atomic_load(G.n_act, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *G.n_act;
    const int per = (n_act + 1023) / 1024;
    const int b = tid * per, e = min(n_act, b + per);
    int sq = 0, sa = 0;
    long long stl = 0, sr = 0;
    unsigned long long sh = 0;  // hot items (low 32 bits) | hot units (high 32 bits)
    for (int i = b; i < e; i++) {
        const int slot = G.act_list[i];
        const int p = HASHED ? key_of[slot] : slot;
        const int c = SAME_KERNEL ? __hip_atomic_load(&G.g_cnt[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : G.g_cnt[slot];
        const int sz = G.pt_size[p];
        sq += c;
        sa += 1;
        int hi_;
        long long hu_;
        const long long rl_len = group_seq_len(G, c, sz, &hi_, &hu_);
        stl += G.rl ? rl_len : seq_weight(c, sz, G.qgroup, G.seg_ovh);
        sh += (unsigned long long)(unsigned)hi_ | ((unsigned long long)(hu_ > 0x3FFFFFFFll ? 0x3FFFFFFFll : hu_) << 32);
        sr += sz;
    }
    // three scans behind ONE pair of barriers: (active count | grouped-query count) packed in 64 bits (both < 2^31, so
    // the halves never carry into each other), tiles, rows (total only)
    long long tq, ta, tt, tr;
    long long aq, aa, at;
    unsigned long long ah, th;  // hot items | hot units: exclusive prefix of this thread, total
    {
        const int lane = tid & 63, wave = tid >> 6;
        unsigned long long i0 = ((unsigned long long)(unsigned)sa << 32) | (unsigned)sq;
        long long i1 = stl, i2 = sr;
        unsigned long long i3 = sh;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o0 = __shfl_up(i0, off);
            const long long o1 = __shfl_up(i1, off), o2 = __shfl_up(i2, off);
            const unsigned long long o3 = __shfl_up(i3, off);
            if (lane >= off) {
                i0 += o0;
                i1 += o1;
                i2 += o2;
                i3 += o3;
            }
        }
        if (lane == 63) {
            s_w[wave] = (long long)i0;
            s_w[16 + wave] = i1;
            s_w[32 + wave] = i2;
            s_w[48 + wave] = (long long)i3;
        }
        __syncthreads();
        unsigned long long p0 = 0, t0 = 0, p3 = 0, t3 = 0;
        long long p1 = 0, t1 = 0, t2 = 0;
        // (not fully unrolled: the compiler then requests all 64 eight-byte partials before the first add -- 128 VGPRs, which at
        //  1024 threads per workgroup is the whole budget: 20-30 registers went to scratch in the middle of this latency chain)
#pragma unroll 2
        for (int w = 0; w < 16; w++) {
            const unsigned long long v0 = (unsigned long long)s_w[w];
            const long long v1 = s_w[16 + w];
            const unsigned long long v3 = (unsigned long long)s_w[48 + w];
            if (w < wave) {
                p0 += v0;
                p1 += v1;
                p3 += v3;
            }
            t0 += v0;
            t1 += v1;
            t2 += s_w[32 + w];
            t3 += v3;
        }
        ah = p3 + i3 - sh;
        th = t3;
        const unsigned long long e0 = p0 + i0 - (((unsigned long long)(unsigned)sa << 32) | (unsigned)sq);
        aq = (long long)(e0 & 0xFFFFFFFFull);
        aa = (long long)(e0 >> 32);
        at = p1 + i1 - stl;
        tq = (long long)(t0 & 0xFFFFFFFFull);
        ta = (long long)(t0 >> 32);
        tt = t1;
        tr = t2;
    }
    if (tid == 0) {
        if (!HASHED) G.g_qoff[G.npids] = (int)tq;
        *G.n_active = (int)ta;
        ActiveInfo sent;
        sent.toff = tt;
        sent.row_off = 0;
        sent.p = -1;
        sent.size = sent.cnt = sent.qoff = 0;
        G.active[ta] = sent;
        *G.n_tiles = tt;
        *G.n_rows_unique = tr;
        *G.n_pairs_live = (int)tq;
        if (G.act_hoff) {
            G.act_hoff[ta] = (int)(th & 0xFFFFFFFFull);
            *G.n_hot = (int)(th & 0xFFFFFFFFull);
            *G.hot_units = (long long)(th >> 32);
        }
    }
    int ahi = (int)(ah & 0xFFFFFFFFull);
    for (int i = b; i < e; i++) {
        const int slot = G.act_list[i];
        const int p = HASHED ? key_of[slot] : slot;
        const int c = SAME_KERNEL ? __hip_atomic_load(&G.g_cnt[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : G.g_cnt[slot];
        G.g_qoff[slot] = (int)aq;
        ActiveInfo inf;
        inf.toff = at;
        inf.row_off = G.pt_off[p];
        inf.p = p;
        inf.size = G.pt_size[p];
        inf.cnt = c;
        inf.qoff = (int)aq;
        G.active[aa] = inf;
        if (G.act_hoff) G.act_hoff[aa] = ahi;
        aq += c;
        aa += 1;
        int hi_;
        long long hu_;
        const long long rl_len = group_seq_len(G, c, G.pt_size[p], &hi_, &hu_);
        at += G.rl ? rl_len : seq_weight(c, G.pt_size[p], G.qgroup, G.seg_ovh);
        ahi += hi_;
    }
}

__global__ __launch_bounds__(1024) void k_group_scan(GroupParams G) {
    __shared__ long long s_w[64];
    group_scan_body<false>(G, s_w);
}

__device__ __forceinline__ void group_scatter_one(const GroupParams &G, int64_t i) {
    // {list, arrival rank} left by group_count_one
    const int p = G.pair_slots[i * QK_SLOTS + 1];
    if (p >= 0) {
        const int pos = G.pair_slots[i * QK_SLOTS + 2];
        G.grouped_q[G.g_qoff[p] + pos] = (int32_t)(i / G.P);
        G.grouped_pair[G.g_qoff[p] + pos] = (int32_t)i;
    }
}

__global__ void k_group_scatter(GroupParams G) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G.npairs) group_scatter_one(G, i);
}

// count + scan + scatter in ONE workgroup for small batches (<= QK_GROUP_SMALL pairs): two launches and two dependent kernel
// boundaries less.  Round 2 kept the counters in global memory (atomics, agent-scope read-backs: ~10 dependent round trips, 13 us
// for 640 pairs, 19 us for 1024) and lost to the three kernels beyond 1024 pairs.  Round 3: the counters live in LDS -- an open-
// addressing table of 2 slots per pair keyed by list number (the batch touches <= npairs lists however many the index has: a rank
// of an N-GPU index sees N x 4096 list numbers), first-hit list, per-list offsets, all in LDS; global memory is read for the pair's
// list and its size and written with the results.
constexpr int QK_GROUP_SMALL_PP = 4;                        // pairs per thread
constexpr int QK_GROUP_SMALL = 1024 * QK_GROUP_SMALL_PP;
constexpr int QK_GROUP_SLOTS = 2 * QK_GROUP_SMALL;          // a power of two
constexpr size_t QK_GROUP_SMALL_LDS = (size_t)(3 * QK_GROUP_SLOTS + QK_GROUP_SMALL) * 4;
__device__ __forceinline__ void group_small_body(const GroupParams &G) {
    extern __shared__ __align__(16) int g_smem[];
    __shared__ long long s_w[64];
    __shared__ int s_nact;
    int *s_key = g_smem, *s_cnt = s_key + QK_GROUP_SLOTS, *s_qoff = s_cnt + QK_GROUP_SLOTS, *s_act = s_qoff + QK_GROUP_SLOTS;
    // (only the slots a batch of this size can reach are cleared: the table is 2 x the pairs rounded up to a power of two)
    int slots = 2048;
    while (slots < 2 * G.npairs) slots <<= 1;
    const int shift = 32 - (31 - __clz(slots));
    for (int i = threadIdx.x; i < slots; i += 1024) {
        s_key[i] = -1;
        s_cnt[i] = 0;
    }
    if (threadIdx.x == 0) s_nact = 0;
    __syncthreads();
    int sl[QK_GROUP_SMALL_PP], pos[QK_GROUP_SMALL_PP];
#pragma unroll
    for (int jj = 0; jj < QK_GROUP_SMALL_PP; jj++) {
        const int64_t i = threadIdx.x + 1024 * (int64_t)jj;
        sl[jj] = -1;
        pos[jj] = 0;
        if (i < G.npairs) {
            G.pair_head[i] = -1;
            G.pair_slots[i * QK_SLOTS] = 0;
            const int p = pair_pid(G, i);
            if (p >= 0) {
                int h = (int)(((uint32_t)p * 2654435761u) >> shift);
                for (;;) {
                    const int old = atomicCAS(&s_key[h], -1, p);
                    if (old == -1 || old == p) break;
                    h = (h + 1) & (slots - 1);
                }
                pos[jj] = atomicAdd(&s_cnt[h], 1);
                if (pos[jj] == 0) s_act[atomicAdd(&s_nact, 1)] = h;  // first-hit order, as k_group_count lists them
                sl[jj] = h;
            }
        }
    }
    __syncthreads();
    {
        GroupParams H = G;
        H.n_act = &s_nact;
        H.act_list = s_act;
        H.g_cnt = s_cnt;
        H.g_qoff = s_qoff;
        group_scan_body<false, true>(H, s_w, s_key);
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < QK_GROUP_SMALL_PP; jj++) {
        const int64_t i = threadIdx.x + 1024 * (int64_t)jj;
        if (sl[jj] >= 0) {
            const int at = s_qoff[sl[jj]] + pos[jj];
            G.grouped_q[at] = (int32_t)(i / G.P);
            G.grouped_pair[at] = (int32_t)i;
        }
    }
}

__global__ __launch_bounds__(1024) void k_group_small(GroupParams G) { group_small_body(G); }
static inline size_t group_small_lds(int64_t) { return QK_GROUP_SMALL_LDS; }

// ---- bound seeding ---------------------------------------------------------------------------------------
// One wave per (query, partition) pair: exact distances (same canonical chain as the MFMA path) from the query to the
// first min(64, n_p) rows of the partition, one row per lane; the k-th smallest of them bounds the query's final k-th
// best, and goes into gtau[q] with atomicMin.  k_scan then starts every segment with a bound instead of +inf, which
// removes most of the cold-start appends/compactions (the slow path of the scan epilogue).
struct SeedParams {
    const int64_t *pids;  // [Q*P] or nullptr (pair i -> list i % P)
    const unsigned long long *pids_packed;  // [Q] (P = 1) or nullptr
    int64_t npairs;
    int P;
    const int32_t *pt_size;
    const int64_t *pt_off;
    int npids;
    const float4 *vecs;
    const float *norms;
    int nblk;
    int d;
    const float *x;   // [Q][d] row-major queries
    const float *xn;  // [Q]
    int k;
    int metric;
    uint32_t *gtau;
    int seed_ranks;      // pairs with r < seed_ranks are sampled (the nearest partitions give the tight bound)
    int strict_first;    // 1: the sample comes from the query's FIRST list or there is no bound (qk_scan_args::seed_first)
};

// grid = Q * seed_ranks waves: wave w -> query w / seed_ranks, rank w % seed_ranks.  M rows per lane: the sample is the
// first min(n_p, 64*M) rows, so that it can bound k <= 64*M.
// CB = 16-column blocks requested per memory round trip (8: one trip for d <= 128, 128 VGPRs of row data; the fused
// seed + group launch runs 1024-thread workgroups -- 128 VGPRs in all -- and takes two trips of 4)
template <int M, int CB>
__device__ __forceinline__ void seed_tau_body(const SeedParams &S, const int64_t w, const int lane) {
    const int64_t qq = w / S.seed_ranks;
    int rr = (int)(w % S.seed_ranks);
    if (rr >= S.P) return;
    int64_t p = -1;
    int size_p = 0;
    // one seeded rank: the nearest list that HOLDS k rows here (a rank of a sharded index owns one list in N: its nearest owned
    // one is what it can learn a bound from; at most 8 places are tried)
    for (int tries = 0; tries < (S.seed_ranks == 1 && !S.strict_first ? 8 : 1) && rr < S.P; tries++, rr++) {
        const int64_t pair = qq * S.P + rr;
        if (S.pids_packed) {
            const unsigned long long v = S.pids_packed[pair];
            p = v == ~0ull ? -1 : (int64_t)(v & 0xFFFFFFFFull);
        } else {
            p = S.pids ? S.pids[pair] : (pair % S.P);
        }
        size_p = (p >= 0 && p < S.npids) ? S.pt_size[p] : 0;
        if (size_p >= S.k) break;
    }
    if (p < 0 || p >= S.npids) return;
    if (size_p < S.k) return;  // fewer than k rows: no bound from this partition
    const int n = min(size_p, 64 * M);
    const int64_t q = qq;
    const float *xq = S.x + q * S.d;
    const float xnq = S.xn[q];
    const int64_t row_base = S.pt_off[p];
    uint32_t key[M];
#pragma unroll
    for (int i = 0; i < M; i++) {
        const int lrow = min(lane + 64 * i, n - 1);  // idle lanes recompute the last row (keeps every load unconditional)
        const int64_t row = row_base + lrow;
        const int64_t tile = row >> 4;
        const int r = (int)(row & 15);
        const float yn = S.norms[row];
        float acc = 0.0f;
        // 8 blocks (128 columns) at a time: the 32 float4 of the lane's row and the 128 query values (2 per lane,
        // broadcast with v_readlane) are requested together, then one k-ordered fmaf chain -- the arithmetic of the MFMA path
        for (int c0 = 0; c0 < S.nblk; c0 += CB) {
            float4 v[CB][4];
#pragma unroll
            for (int c = 0; c < CB; c++) {
                const int cc = min(c0 + c, S.nblk - 1);
                const float4 *blk = S.vecs + (tile * S.nblk + cc) * 64 + r;
#pragma unroll
                for (int g = 0; g < 4; g++) v[c][g] = blk[g * 16];
            }
            const int colA = c0 * 16 + lane, colB = c0 * 16 + 64 + lane;
            const float xa = colA < S.d ? xq[colA] : 0.0f;
            const float xb = colB < S.d ? xq[colB] : 0.0f;
#pragma unroll
            for (int c = 0; c < CB; c++) {
                if (c0 + c < S.nblk) {
                    const float e[16] = {v[c][0].x, v[c][1].x, v[c][2].x, v[c][3].x, v[c][0].y, v[c][1].y, v[c][2].y, v[c][3].y,
                                         v[c][0].z, v[c][1].z, v[c][2].z, v[c][3].z, v[c][0].w, v[c][1].w, v[c][2].w, v[c][3].w};
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        const int cl = c * 16 + t;  // column within this group of 128
                        // (the builtin is typed int -> int: move the bits, not the value)
                        const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cl < 64 ? xa : xb), cl & 63));
                        acc = __fmaf_rn(e[t], xv, acc);  // padded columns: e[t] == 0 and xv == 0 -> acc unchanged
                    }
                }
            }
        }
        key[i] = S.metric == QK_METRIC_L2 ? ord_from_l2(l2_expanded(xnq, yn, acc)) : ord_from_ip(acc);
        if (lane + 64 * i >= n) key[i] = 0xFFFFFFFFu;
    }
    // rank of every sampled key under (key, sample position); the one with rank k-1 is the bound
    int rk[M];
#pragma unroll
    for (int i = 0; i < M; i++) rk[i] = 0;
#pragma unroll
    for (int jj = 0; jj < M; jj++) {
        for (int t = 0; t < 64; t++) {
            const uint32_t ot = __builtin_amdgcn_readlane(key[jj], t);
            const int pos_t = t + 64 * jj;
#pragma unroll
            for (int i = 0; i < M; i++) rk[i] += (ot < key[i] || (ot == key[i] && pos_t < lane + 64 * i)) ? 1 : 0;
        }
    }
    uint32_t bound = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < M; i++) {
        const uint64_t mk = __ballot(rk[i] == S.k - 1);
        if (mk) bound = __builtin_amdgcn_readlane(key[i], __ffsll((unsigned long long)mk) - 1);
    }
    if (lane == 0 && bound != 0xFFFFFFFFu) atomicMax(&S.gtau[q], ~bound);  // gtau holds ~bound: 0 = no bound yet
}

template <int M>
__global__ __launch_bounds__(64) void k_seed_tau(SeedParams S) {
    seed_tau_body<M, 8>(S, blockIdx.x, threadIdx.x);
}

// The same for larger batches: the counting pass of the three-kernel grouping and the seeding kernel in one launch (the
// first n_count_blocks workgroups count, the others carry 4 seeding waves each).
template <int M>
__global__ __launch_bounds__(256) void k_group_count_seed(GroupParams G, SeedParams S, int n_count_blocks, int64_t n_seed_waves) {
    if ((int)blockIdx.x < n_count_blocks) {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i < G.npairs) group_count_one(G, i);
        return;
    }
    const int64_t w = ((int64_t)blockIdx.x - n_count_blocks) * 4 + (threadIdx.x >> 6);
    if (w < n_seed_waves) seed_tau_body<M, 8>(S, w, threadIdx.x & 63);
}

// Grouping and bound seeding of a small batch in ONE launch: both depend only on the probed-partition lists, neither on the
// other.  Workgroup 0 is k_group_small, the others carry 16 seeding waves each (k <= 64: a 64-row sample per wave).  In line
// the two kernels took 12.4 + 10.1 us of the 0.31 ms bench step.
__global__ __launch_bounds__(1024) void k_group_seed(GroupParams G, SeedParams S, int64_t n_seed_waves) {
    if (blockIdx.x == 0) {
        group_small_body(G);
        return;
    }
    const int64_t w = ((int64_t)blockIdx.x - 1) * 16 + (threadIdx.x >> 6);
    if (w < n_seed_waves) seed_tau_body<1, 4>(S, w, threadIdx.x & 63);
}


// exact key of one sampled row (row `lrow` of the partition that starts at arena row `row_base`) for the query xq: the
// arithmetic of the MFMA path -- one k-ordered fmaf chain; the query values are broadcast with v_readlane
__device__ __forceinline__ uint32_t seed_row_key(const SeedParams &S, const float *xq, float xnq, int64_t row_base, int lrow, int lane) {
    const int64_t row = row_base + lrow;
    const int64_t tile = row >> 4;
    const int r = (int)(row & 15);
    const float yn = S.norms[row];
    float acc = 0.0f;
    // 8 blocks (128 columns) at a time: the 32 float4 of the lane's row and the 128 query values (2 per lane,
    // broadcast with v_readlane) are requested together, then one k-ordered fmaf chain -- the arithmetic of the MFMA path
    for (int c0 = 0; c0 < S.nblk; c0 += 8) {
        float4 v[8][4];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int cc = min(c0 + c, S.nblk - 1);
            const float4 *blk = S.vecs + (tile * S.nblk + cc) * 64 + r;
#pragma unroll
            for (int g = 0; g < 4; g++) v[c][g] = blk[g * 16];
        }
        const int colA = c0 * 16 + lane, colB = c0 * 16 + 64 + lane;
        const float xa = colA < S.d ? xq[colA] : 0.0f;
        const float xb = colB < S.d ? xq[colB] : 0.0f;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            if (c0 + c < S.nblk) {
                const float e[16] = {v[c][0].x, v[c][1].x, v[c][2].x, v[c][3].x, v[c][0].y, v[c][1].y, v[c][2].y, v[c][3].y,
                                     v[c][0].z, v[c][1].z, v[c][2].z, v[c][3].z, v[c][0].w, v[c][1].w, v[c][2].w, v[c][3].w};
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const int cl = c * 16 + t;  // column within this group of 128
                    // (the builtin is typed int -> int: move the bits, not the value)
                    const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cl < 64 ? xa : xb), cl & 63));
                    acc = __fmaf_rn(e[t], xv, acc);  // padded columns: e[t] == 0 and xv == 0 -> acc unchanged
                }
            }
        }
    }
    return S.metric == QK_METRIC_L2 ? ord_from_l2(l2_expanded(xnq, yn, acc)) : ord_from_ip(acc);
}

// W waves per (query, partition) pair, 64 W sampled rows, one memory round trip deep like the one-wave kernel: the bound is the
// k-th smallest of 4x as many rows (k = 10: the 4 % quantile of the partition instead of the 16 % one), so k_scan starts
// every segment with fewer appends and compactions.  Keys meet in LDS; every thread ranks its own.
template <int W>
__global__ __launch_bounds__(64 * W) void k_seed_tau_wg(SeedParams S) {
    __shared__ uint32_t s_key[64 * W];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t qq = blockIdx.x / S.seed_ranks;
    const int rr = blockIdx.x % S.seed_ranks;
    if (rr >= S.P) return;
    const int64_t pair = qq * S.P + rr;
    int64_t p;
    if (S.pids_packed) {
        const unsigned long long v = S.pids_packed[pair];
        p = v == ~0ull ? -1 : (int64_t)(v & 0xFFFFFFFFull);
    } else {
        p = S.pids ? S.pids[pair] : (pair % S.P);
    }
    if (p < 0 || p >= S.npids) return;
    const int size_p = S.pt_size[p];
    if (size_p < S.k) return;  // (uniform over the workgroup: nobody reaches the barrier)
    const int n = min(size_p, 64 * W);
    const int mine = wv * 64 + lane;
    uint32_t key = seed_row_key(S, S.x + qq * S.d, S.xn[qq], S.pt_off[p], min(mine, n - 1), lane);
    if (mine >= n) key = 0xFFFFFFFFu;
    s_key[threadIdx.x] = key;
    __syncthreads();
    int rk = 0;
    for (int t = 0; t < 64 * W; t++) {
        const uint32_t ot = s_key[t];
        rk += (ot < key || (ot == key && t < (int)threadIdx.x)) ? 1 : 0;
    }
    if (rk == S.k - 1 && key != 0xFFFFFFFFu) atomicMax(&S.gtau[qq], ~key);  // gtau holds ~bound: 0 = no bound yet
}

#ifndef QK_SEED_M_WIDE
#define QK_SEED_M_WIDE 0   // 64 < k <= 128: rows of the nearest list sampled for the bound = 64 * this (2 / 4 / 8); 0 = by nprobe (4, 8 from 16 on)
#endif

// ---- the scan kernel ---------------------------------------------------------------------------------
// Compile-time experiment switches (scripts/scan_ab.sh builds one library per combination)
#ifndef QK_OPT_EARLY_LOAD
#define QK_OPT_EARLY_LOAD 1   // first tile's loads before the query staging
#endif
#ifndef QK_OPT_EARLY_REC
#define QK_OPT_EARLY_REC 0    // reserve record slots at segment start + deferred rec_next store
#endif
#ifndef QK_OPT_ONE_BALLOT
#define QK_OPT_ONE_BALLOT 1   // one ballot per tile in the steady state
#endif

#ifndef QK_DYN_PCT_DEFAULT
// Dynamic tail.  Measured on the bench configuration (QK_SCAN_DYN_PCT / QK_SCAN_DYN_CHUNK sweeps): 20 % in 16-tile chunks
// takes 5-12 % off k_scan at 6 or 8 waves per CU, but every chunk is a segment and leaves a record per live query, and
// k_merge pays it back (+10 us); at 4 waves per CU the static cut is already balanced.  Off by default.
// Also measured and dropped: tail stealing (the last 6-20 % of every wave's share guarded by claim bits; the owner goes on
// inside its segment, idle waves take unreached tails): parity held and k_merge stayed at 0.010 ms, but k_scan went
// 0.263 -> 0.268 (20 %) .. 0.297 ms (6 %): a stolen piece costs its work plus a segment start, a single bit says nothing
// about how far behind the owner is, and late steals end after the owner would have.  The wave end times (mean 0.227 ms,
// max 0.26 ms) follow the appends a range happens to see (80 vs 190 per wave between the fastest and slowest decile) and
// the XCD (workgroups with blockIdx % 8 in {1, 6, 7} end 3 % later) -- neither is known when the cut is made.
#define QK_DYN_PCT_DEFAULT 0     // share of the tile sequence handed out dynamically (0 = static cut only)
#endif
#ifndef QK_DYN_CHUNK_DEFAULT
#define QK_DYN_CHUNK_DEFAULT 16  // tiles per dynamic chunk
#endif

#ifndef QK_OPT_SELECT1
#define QK_OPT_SELECT1 1      // in-loop compaction of one-wave pools (k <= 36) by bisection select instead of the rank sort
#endif
#ifndef QK_OPT_NT
#define QK_OPT_NT 1           // non-temporal loads for the streamed partition rows (measured: loads-only 5.74 -> 6.44 TB/s)
#endif
#ifndef QK_OPT_STEP_DRAIN
#define QK_OPT_STEP_DRAIN 0   // s_waitcnt vmcnt(0) after every step (limits the bytes in flight per wave)
#endif

__device__ __forceinline__ float4 qk_ld_stream(const float4 *p) {
#if QK_OPT_NT
    const f32x4 t = __builtin_nontemporal_load((const f32x4 *)p);
    return make_float4(t[0], t[1], t[2], t[3]);
#else
    return *p;
#endif
}

// MODE 0 = product; 1 = skip the top-k epilogue; 2 = loads only (probe variants for bandwidth attribution, QK_SCAN_MODE);
// 3 = product with query-sharing workgroups (ScanParams::qshare) -- a compile-time variant: the extra branches cost the
// plain path 8 % when they were decided at run time
// L2: the metric, compile-time as well -- as a runtime flag it left a uniform branch around every result element of the epilogue
template <int DB, int MAXCH, int MODE, bool L2>
__global__ __launch_bounds__(256) void k_scan(ScanParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    // nw waves per workgroup (1, 2 or 4) share ONE LDS query tile and split every segment's tiles between them; each
    // wave keeps its own pools and emits its own records.  nw > 1 is chosen by the host for wide rows, where a
    // wave-private query tile (1 KiB per 16 columns) would leave room for only 2-3 waves per CU.
    const int lane = threadIdx.x & 63;
    // P.pack > 1: the hardware workgroup is a bundle of `pack` INDEPENDENT one-wave workgroups of the cut (own query tile, own
    // pools, own range; no barrier anywhere on that path).  The dispatcher spreads the waves of one workgroup over the SIMDs
    // of a CU, which it does not do for single-wave workgroups: QK_SCAN_WAVE_CLOCK showed 57 SIMDs holding two of the 1024
    // waves (and 57 none) on some launches, and those 114 waves set the kernel time (0.26 -> 0.31 ms).
    const int wv_phys = threadIdx.x >> 6;
    const int pack = P.pack;
    const int wv = pack > 1 ? 0 : wv_phys, nw = pack > 1 ? 1 : (int)(blockDim.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int nblk = P.nblk, C = P.C, k = P.k;
    constexpr bool l2 = L2;
    constexpr bool qshare = MODE == 3;
    constexpr bool PRODUCT = MODE == 0 || MODE == 3;
    constexpr bool EMIT = MODE == 4;  // wide-k path: keys out, selection happens in k_select_rows_large afterwards
    const size_t per_wave = (size_t)nblk * 1024 + (size_t)16 * C * 12;  // qshare: every wave owns a query tile + pools
    unsigned char *smem_w = smem + (pack > 1 ? (size_t)wv_phys * P.pack_lds : 0);
    float4 *qs = (float4 *)(smem_w + (qshare ? wv * per_wave : 0));           // [nblk*64] (shared by the workgroup unless qshare)
    unsigned char *pool_base = qshare ? smem_w + wv * per_wave + (size_t)nblk * 1024
                                      : smem_w + (size_t)nblk * 1024 + (size_t)wv * 16 * C * 12;
    int64_t *pool_id = (int64_t *)pool_base;                                   // [16][C]
    uint32_t *pool_ord = (uint32_t *)(pool_base + (size_t)16 * C * 8);         // [16][C]
    uint32_t *my_ord = pool_ord + j * C;
    int64_t *my_id = pool_id + j * C;
    const int ncd = nblk / DB;  // d-chunks per tile

    // ---- this wave's contiguous share of the global tile sequence ------------------------------------------
    const long long T = *P.n_tiles;
    const long long W = pack > 1 ? (long long)gridDim.x * pack : (long long)gridDim.x;
    const long long vblock = pack > 1 ? (long long)blockIdx.x * pack + wv_phys : (long long)blockIdx.x;
    const bool dyn = P.dyn_counter != nullptr && nw == 1;
    // static share: an equal cut of the first Ts tiles; the rest is claimed chunk by chunk by whoever finishes first
    // (waves do not finish together: HBM channel and XCD placement make equal tile counts take unequal time)
    const long long Ts = dyn ? T - (T * P.dyn_pct) / 100 : T;
    long long T0 = (Ts * vblock) / W, T1 = (Ts * (vblock + 1)) / W;
    if (P.xcd_on && !dyn) {
        // weighted cut: the split points are f(a) = T a / total for cumulative weights a; a wave's end is its successor's start
        const int np = pack > 1 ? pack : 1;
        long long pre[9];
        pre[0] = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) pre[i + 1] = pre[i] + P.xcd_w[i];
        const long long u = blockIdx.x, gu = gridDim.x;
        const long long total = ((gu >> 3) * pre[8] + pre[gu & 7]) * np;
        const long long a0 = ((u >> 3) * pre[8] + pre[u & 7]) * np + (long long)P.xcd_w[u & 7] * (pack > 1 ? wv_phys : 0);
        const long long a1 = a0 + P.xcd_w[u & 7];
        T0 = a0 <= 0 ? 0 : (long long)((double)T * (double)a0 / (double)total);
        T1 = a1 >= total ? T : (long long)((double)T * (double)a1 / (double)total);
    }
    if (!dyn && T1 <= T0) return;
    const long long wc0 = (P.wave_clock || P.xcd_stat) ? wall_clock64() : 0;
    const int n_active = *P.n_active;
    int pend_rec = -1, pend_old = -1, pend_cnt = 0;  // deferred header store of this lane's previous record
    int dbg_comp = 0, dbg_app = 0, dbg_seg = 0;       // probe counters (QK_SCAN_WAVE_CLOCK)
    long long dbg_t_end = 0, dbg_t_stage = 0;         // probe: ticks spent in segment ends / query staging
    for (;;) {
    if (T1 > T0) {
    // 64-ary search for the partition containing tile T0: active[lo].toff <= T0 < active[lo+1].toff
    int lo = 0, hi = n_active;
    while (hi - lo > 1) {
        const int span = hi - lo;
        const int step = (span + 63) >> 6;
        const int probe = min(lo + (lane + 1) * step, hi);          // lanes probe lo+step, lo+2step, ..., hi
        const bool gt = (probe >= hi) || (P.active[probe].toff > T0);  // active[hi].toff > T0 by the invariant
        const uint64_t m = __ballot(gt);
        const int first = __ffsll((unsigned long long)m) - 1;      // first lane whose probe is beyond T0 (always exists)
        const int nlo = min(lo + first * step, hi - 1);
        const int nhi = min(lo + (first + 1) * step, hi);
        lo = nlo;
        hi = nhi;
    }
    int ai = lo;
    long long cur = T0;

    while (cur < T1) {
        // ---- segment = tiles [tl, tend) of item (p, qt) ---------------------------------------------------------
        const ActiveInfo inf = P.active[ai];
        const long long base = inf.toff;
        const int size_p = inf.size;
        const int64_t row_off = inf.row_off;
        const int ntl = (size_p + 15) >> 4;
        const int cnt_p = inf.cnt;
        const int nqt = (cnt_p + 15) >> 4;
        const int G = qshare ? nw : 1;           // query tiles per pass over the partition
        const int ngrp = (nqt + G - 1) / G;      // passes (items) of this partition
        const long long local = cur - base;
        // position inside the partition's weighted sequence (seq_weight): full passes weigh G units per row tile, the last
        // pass q' = 1, 2 or 4; a row tile belongs to the range that holds its first unit
        const int nfull = nqt / G;
        const int ovh = P.seg_ovh;  // units in front of every pass that stand for the cost of starting it (no row tiles)
        const long long wfull = (long long)ntl * G + ovh;
        int grp, wq;
        long long off;
        if (local < nfull * wfull) {
            grp = (int)(local / wfull);
            off = local - grp * wfull;
            wq = G;
        } else {
            grp = nfull;
            off = local - nfull * wfull;
            const int rem = nqt - nfull * G;
            wq = rem <= 1 ? 1 : rem <= 2 ? 2 : 4;
        }
        const long long pass_len = (long long)ntl * wq + ovh;
        const long long off_end = min(pass_len, off + (T1 - cur));
        const int tl_wg = (int)((max(0ll, off - ovh) + wq - 1) / wq);
        const int tend_wg = (int)((max(0ll, off_end - ovh) + wq - 1) / wq);
        cur += off_end - off;
        if (tend_wg <= tl_wg) {  // a range boundary inside one row tile's units or inside the start charge: nothing here
            if (grp == ngrp - 1 && off_end == pass_len) ai++;
            continue;
        }
        // this wave's query tile and its contiguous share of the segment's tiles:
        //   split mode (wide rows) / nw == 1: one query tile, the tiles are cut nw ways;
        //   qshare: the nq_g query tiles of the pass go to waves 0..nq_g-1 (rounded up to a power of two); when the pass
        //   has fewer query tiles than waves, the spare waves take a second / third / fourth cut of the tiles
        int qt = grp, part = wv, parts = nw;
        bool idle = false;
        if (qshare) {
            const int nq_g = min(G, nqt - grp * G);
            const int nq_p = nq_g <= 1 ? 1 : nq_g <= 2 ? 2 : 4;
            parts = max(1, nw / nq_p);
            const int ql = wv % nq_p;
            part = wv / nq_p;
            idle = ql >= nq_g || part >= parts;
            qt = grp * G + min(ql, nq_g - 1);
        }
        const int tl = idle ? tend_wg : tl_wg + (int)(((long long)(tend_wg - tl_wg) * part) / parts);
        const int tend = idle ? tend_wg : tl_wg + (int)(((long long)(tend_wg - tl_wg) * (part + 1)) / parts);
        if (grp == ngrp - 1 && off_end == pass_len) ai++;  // item sequence of this partition exhausted
        const int nq = idle ? 0 : min(16, cnt_p - 16 * qt);
        const int gidx = inf.qoff + 16 * qt + j;
        // grouped entry of this lane's query + record slots for the segment: issued FIRST so that they return first
        // (loads complete in order); the first tile's loads go out right behind them and fly under the query staging
        const int myq = (j < nq) ? P.grouped_q[gidx] : -1;
        const int mypair = (j < nq) ? P.grouped_pair[gidx] : -1;
        int base_rec = 0;
        if (QK_OPT_EARLY_REC && PRODUCT && lane == 0) base_rec = atomicAdd(P.rec_counter, nq);
        uint32_t tau = 0xFFFFFFFFu;
        int cnt = 0;
        dbg_seg++;
        float xnj = 0.0f;
        {
            // (a wave whose share is empty still issues the static loads: keep them inside the segment)
            const int64_t tile_abs0 = (row_off >> 4) + min(tl, tend_wg - 1);
            const float4 *src = P.vecs + tile_abs0 * nblk * 64 + lane;
            const float4 *nsrc = (const float4 *)(P.norms + (tile_abs0 << 4)) + g;  // +4 float4 per tile
            // ids of this lane's 4 rows travel with the tile (static prefetch): an id load inside the append path
            // would force s_waitcnt vmcnt(0) and drain the prefetched tile every time a candidate passes
            const longlong2 *isrc = (const longlong2 *)(P.ids + (tile_abs0 << 4)) + 2 * g;  // +8 longlong2 per tile
            // norms + ids are double-buffered with the tile data (y0/i0* with a0, y1/i1* with a1): no register copies,
            // so the only wait on a buffer is its first use -- after the following step's loads have been issued
            longlong2 i00 = {0, 0}, i01 = {0, 0}, i10 = {0, 0}, i11 = {0, 0};
            const int nsteps = (tend - tl) * ncd;
            float4 a0[DB], a1[DB];
            float4 y0 = make_float4(0.f, 0.f, 0.f, 0.f), y1 = y0;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float probe_sink = 0.f;
            int dch = 0;    // d-chunk of the step being computed
            int tile = tl;  // tile of the step being computed
            int ldch = 0;   // d-chunk of the step being loaded
            int ltile = 0;  // tiles loaded so far (relative)

            // The load stream is STATIC (same loads every iteration, clamped at the end of the segment) so that the
            // compiler can place counted s_waitcnt vmcnt(N) and keep the next step's loads in flight under the MFMAs.
            int lS = 0;  // next step to load (clamped to nsteps-1)
#define QK_LOAD(A, Y, I0, I1)                                         \
    {                                                                 \
        const float4 *pp_ = src + (int64_t)lS * (DB * 64);            \
        if (qshare) { /* the other waves of the workgroup read the same tile: keep it cacheable */ \
            _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) A[b_] = pp_[b_ * 64];      \
        } else {                                                      \
            _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++)         \
                A[b_] = qk_ld_stream(pp_ + b_ * 64);                  \
        }                                                             \
        Y = nsrc[(int64_t)ltile * 4];                                 \
        I0 = isrc[(int64_t)ltile * 8];                                \
        I1 = isrc[(int64_t)ltile * 8 + 1];                            \
        if (lS < nsteps - 1) {                                        \
            lS++;                                                     \
            if (++ldch == ncd) {                                      \
                ldch = 0;                                             \
                ltile++;                                              \
            }                                                         \
        }                                                             \
    }

#define QK_STEP(A, Y, I0, I1, LIVE)                                                                                    \
    {                                                                                                      \
        if (dch == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};                                                   \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) {                                                \
            if (MODE == 2) {                                                                               \
                acc[0] += A[b_].x;                                                                         \
                acc[1] += A[b_].y;                                                                         \
                acc[2] += A[b_].z;                                                                         \
                acc[3] += A[b_].w;                                                                         \
            } else {                                                                                       \
                const float4 bq_ = qs[(dch * DB + b_) * 64 + lane];                                        \
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].x, bq_.x, acc, 0, 0, 0);                  \
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].y, bq_.y, acc, 0, 0, 0);                  \
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].z, bq_.z, acc, 0, 0, 0);                  \
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].w, bq_.w, acc, 0, 0, 0);                  \
            }                                                                                              \
        }                                                                                                  \
        if (++dch == ncd) {                                                                                \
            dch = 0;                                                                                       \
            if (PRODUCT || EMIT) {                                                                         \
                epilogue(tile, LIVE, Y, I0, I1);                                                           \
            } else {                                                                                       \
                probe_sink += acc[0] + acc[1] + acc[2] + acc[3] + Y.x + (float)I0.x + (float)I1.x;         \
            }                                                                                              \
            tile++;                                                                                        \
        }                                                                                                  \
    }

            auto epilogue = [&](int tl_, bool live, const float4 yn, const longlong2 ia, const longlong2 ib) {
                const int row0 = tl_ << 4;
                const float yv[4] = {yn.x, yn.y, yn.z, yn.w};
                const int64_t idv[4] = {ia.x, ia.y, ib.x, ib.y};
                if (EMIT) {
                    if (live && myq >= 0) {
                        uint32_t *dst = P.key_out + P.pair_base[mypair] + row0 + 4 * g;
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) {
                            const float v = acc[reg];
                            if (row0 + 4 * g + reg < size_p)
                                dst[reg] = l2 ? ord_from_l2(l2_expanded(xnj, yv[reg], v)) : ord_from_ip(v);
                        }
                    }
                    return;
                }
                // every 8 tiles pick up bounds published by other waves working on the same query (only when queries
                // probe more than one partition: gtau is null otherwise).  Measured (scan_probe.py, 10M x 128, P=32)
                // against a per-tile plain (L1-stale) load, a per-tile sc1 load in the prefetch stream and an
                // exchange at compaction time: this variant is 5-25 % faster although consuming the load drains
                // the prefetched tile.
                if (P.gtau && P.tau_refresh && (tl_ & 7) == 7 && myq >= 0)
                    tau = min(tau, ~__hip_atomic_load(&P.gtau[myq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                uint32_t ordv[4];
                bool anyp = false;
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int row = row0 + 4 * g + reg;
                    const bool valid = live && (myq >= 0) && (row < size_p);
                    const float v = acc[reg];
                    const uint32_t o = l2 ? ord_from_l2(l2_expanded(xnj, yv[reg], v)) : ord_from_ip(v);
                    ordv[reg] = valid ? o : 0xFFFFFFFFu;  // an invalid row can never pass (tau < 0xFFFFFFFF once set; see below)
                    anyp |= valid && o <= tau;
                }
                // steady state: nothing beats the running k-th best -> one ballot, one branch per tile
                if (!QK_OPT_ONE_BALLOT || __ballot(anyp)) {
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const uint32_t ord = ordv[reg];
                        const bool pass = ord != 0xFFFFFFFFu && ord <= tau;
                        const uint64_t m = __ballot(pass);
                        if (m) {
                            const uint64_t gm = m & (0x0001000100010001ull << j);
                            if (pass) {
                                const int slot = cnt + __popcll(gm & ((1ull << lane) - 1ull));
                                my_ord[slot] = ord;
                                my_id[slot] = idv[reg];
                            }
                            cnt += __popcll(gm);
                            dbg_app += __popcll(m);
                            uint64_t need = __ballot(cnt > C - 4) & 0xFFFFull;
                            while (need) {
                                dbg_comp++;
                                const int jq = __ffsll((unsigned long long)need) - 1;
                                need &= need - 1;
                                const int n = __builtin_amdgcn_readlane(cnt, jq);
                                uint32_t kth;
                                int nn;
                                if (MAXCH > 1 || QK_OPT_SELECT1) {
                                    nn = select_pool<MAXCH>(pool_ord + jq * C, pool_id + jq * C, n, k, lane, kth);
                                } else {
                                    nn = compact_pool<MAXCH>(pool_ord + jq * C, pool_id + jq * C, n, k, lane);
                                    kth = nn >= k ? pool_ord[jq * C + k - 1] : 0xFFFFFFFFu;
                                }
                                if (j == jq) {
                                    cnt = nn;
                                    if (nn >= k) {
                                        tau = min(tau, kth);
                                        // publish (fire and forget: no returned value, no wait)
                                        if (P.gtau && P.tau_publish && lane < 16) atomicMax(&P.gtau[myq], ~tau);
                                    }
                                }
                            }
                        }
                    }
                }
            };

            if (QK_OPT_EARLY_LOAD) QK_LOAD(a0, y0, i00, i01);
            // ---- query tile -> LDS in B-operand lane order (wave-private), while the first tile is in flight --------
            const long long dbg_s0 = P.wave_clock ? wall_clock64() : 0;
            {
                const int qsafe = myq >= 0 ? myq : 0;
                const float4 *qsrc = P.xq4 + (int64_t)qsafe * nblk * 4 + g;
                if (P.gtau) tau = ~__hip_atomic_load(&P.gtau[qsafe], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (l2) xnj = P.xn[qsafe];
                // LDS only (no vmcnt wait: the first tile stays in flight): every wave has left the previous tile
                const bool coop = nw > 1 && !qshare;  // split mode: one query tile staged by all waves, fenced by barriers
                if (coop) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                for (int cb0 = coop ? wv * DB : 0; cb0 < nblk; cb0 += coop ? nw * DB : DB) {
                    float4 qv[DB];
#pragma unroll
                    for (int b = 0; b < DB; b++) qv[b] = qsrc[(cb0 + b) * 4];
#pragma unroll
                    for (int b = 0; b < DB; b++) {
                        if (myq < 0) qv[b] = make_float4(0.f, 0.f, 0.f, 0.f);
                        qs[(cb0 + b) * 64 + lane] = qv[b];
                    }
                }
                if (myq < 0) {
                    tau = 0xFFFFFFFFu;
                    xnj = 0.0f;
                }
                if (coop) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (P.wave_clock) dbg_t_stage += wall_clock64() - dbg_s0;
            if (!QK_OPT_EARLY_LOAD) QK_LOAD(a0, y0, i00, i01);
            // (a third tile buffer was measured twice at 4 waves per CU: 0.265 -> 0.296 ms, slower, with counted vmcnt waits in
            //  the ISA; at 3 waves per CU it makes no difference.  Probe modes on the bench configuration: loads only 0.229 ms,
            //  + MFMA 0.236 ms, + top-k 0.265 ms at any of 4 / 6 / 8 waves per CU -- the gap to the stream is the top-k path
            //  (cold starts after the seed bound, segment-end compaction and record emission), not latency hiding.)
            for (int s = 0; s < nsteps; s += 2) {
                QK_LOAD(a1, y1, i10, i11);
                QK_STEP(a0, y0, i00, i01, true);
                if (QK_OPT_STEP_DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                QK_LOAD(a0, y0, i00, i01);
                QK_STEP(a1, y1, i10, i11, s + 1 < nsteps);
                if (QK_OPT_STEP_DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#undef QK_LOAD
#undef QK_STEP
            if (!PRODUCT && probe_sink == 12345.678f) my_ord[0] = 1;  // keep the probe's loads alive
        }
        // ---- segment end: final compaction (sorts, caps at k), publish bound, emit records ---------------------------
        const long long dbg_e0 = P.wave_clock ? wall_clock64() : 0;
        {
            uint64_t need = __ballot(cnt > 0) & 0xFFFFull;
            while (need) {
                const int jq = __ffsll((unsigned long long)need) - 1;
                need &= need - 1;
                const int n = __builtin_amdgcn_readlane(cnt, jq);
                const int nn = compact_pool<MAXCH>(pool_ord + jq * C, pool_id + jq * C, n, k, lane);
                if (j == jq) cnt = nn;
            }
            const uint64_t have = __ballot(cnt > 0) & 0xFFFFull;
            if (have) {
                // the two atomics of an emission -- a slot in the pair's line, record numbers for the segment -- do not depend
                // on each other: both are issued before either result is awaited (one round trip instead of two)
                int slot = -1;
                if (lane < 16 && cnt > 0) slot = atomicAdd(&P.pair_slots[(int64_t)mypair * QK_SLOTS], 1);
                if (!QK_OPT_EARLY_REC && lane == 0) base_rec = atomicAdd(P.rec_counter, nq);
                const int rec0 = __builtin_amdgcn_readfirstlane(base_rec);
                int myrec = -1;
                if (lane < 16 && cnt > 0) {
                    myrec = rec0 + lane;
                    // the first 31 records of a pair are listed in its slot line (the merge fetches them together, no
                    // pointer chase); further ones are chained through pair_head.  (max_recs is an upper bound of the
                    // records a launch can emit; a slot taken for a record beyond it reads as "none")
                    if (slot < QK_SLOTS - 1) P.pair_slots[(int64_t)mypair * QK_SLOTS + 1 + slot] = myrec < P.max_recs ? myrec : -1;
                    if (myrec >= P.max_recs) *P.overflow = 1;  // never, if the host bound holds: the context reports it
                    if (myrec < P.max_recs) {
                        // the store of the previous head is deferred to the next emit (or kernel end) so that the
                        // wave does not stall on the exchange's round trip
                        if (pend_rec >= 0) P.rec_hdr[pend_rec] = make_int2(pend_old, pend_cnt);
                        pend_old = -1;
                        if (slot >= QK_SLOTS - 1) pend_old = atomicExch(&P.pair_head[mypair], myrec);
                        pend_rec = myrec;
                        pend_cnt = cnt;
                        if (!QK_OPT_EARLY_REC) {
                            P.rec_hdr[pend_rec] = make_int2(pend_old, pend_cnt);
                            pend_rec = -1;
                        }
                        if (P.gtau && P.tau_publish && cnt >= k) atomicMax(&P.gtau[myq], ~my_ord[k - 1]);
                    }
                }
                uint64_t todo = have;
                while (todo) {
                    const int jq = __ffsll((unsigned long long)todo) - 1;
                    todo &= todo - 1;
                    const int n = __builtin_amdgcn_readlane(cnt, jq);
                    const int rec = __builtin_amdgcn_readlane(myrec, jq);
                    if (rec < P.max_recs)
                        for (int e = lane; e < n; e += 64) {
                            P.rec_ord[(int64_t)rec * k + e] = pool_ord[jq * C + e];
                            P.rec_id[(int64_t)rec * k + e] = pool_id[jq * C + e];
                        }
                }
            }
        }
        if (P.wave_clock) dbg_t_end += wall_clock64() - dbg_e0;
    }
    }  // range
        if (!dyn) break;
        unsigned long long c = 0;
        if (lane == 0) c = atomicAdd(P.dyn_counter, (unsigned long long)P.dyn_chunk);
        c = __shfl(c, 0);
        T0 = Ts + (long long)c;
        if (T0 >= T) break;
        T1 = min(T, T0 + P.dyn_chunk);
    }
    if (pend_rec >= 0) P.rec_hdr[pend_rec] = make_int2(pend_old, pend_cnt);
    if (P.xcd_stat && lane == 0 && (nw == 1 || wv == 0)) {
        atomicAdd(&P.xcd_stat[blockIdx.x & 7], (unsigned long long)(wall_clock64() - wc0));
        atomicAdd(&P.xcd_stat[8 + (blockIdx.x & 7)], 1ull);
    }
    if (P.wave_clock && lane == 0) {
        long long *wcp = P.wave_clock + 8 * (pack > 1 ? vblock : (long long)blockIdx.x * nw + wv);
        wcp[0] = wc0;
        wcp[1] = wall_clock64();
        wcp[2] = dbg_comp;
        wcp[3] = dbg_app;
        wcp[4] = dbg_seg;
        wcp[5] = dbg_t_end;
        wcp[6] = dbg_t_stage;
        // where the wave ran: HW_ID (simd [5:4], cu [11:8], sh [12], se [15:13]) and XCC_ID [3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
        wcp[7] = ((long long)xcc << 32) | hw;
    }
}

// ---- host orchestration -------------------------------------------------------------------------------------
// row-per-lane form (qk_scan_rl.hip)
int qk_launch_merge(qk_ctx *ctx, MergeParams mp, dim3 mgrid);  // qk_merge.hip

template <int DB, int MAXCH, int MODE, bool L2>
static int launch_scan_k(dim3 grid, dim3 block, size_t lds, hipStream_t st, const ScanParams &sp) {
    QK_HIP(hipFuncSetAttribute((const void *)k_scan<DB, MAXCH, MODE, L2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_scan<DB, MAXCH, MODE, L2>), grid, block, lds, st, sp);
    return QK_OK;
}
template <int DB, int MAXCH, int MODE>
static int launch_scan_m(dim3 grid, dim3 block, size_t lds, hipStream_t st, const ScanParams &sp) {
    return sp.metric == QK_METRIC_L2 ? launch_scan_k<DB, MAXCH, MODE, true>(grid, block, lds, st, sp)
                                     : launch_scan_k<DB, MAXCH, MODE, false>(grid, block, lds, st, sp);
}

template <int DB, int MAXCH>
static int launch_scan_t(dim3 grid, dim3 block, size_t lds, hipStream_t st, const ScanParams &sp) {
    return launch_scan_m<DB, MAXCH, 0>(grid, block, lds, st, sp);
}

template <int DB, int MAXCH>
static int launch_scan_qs(dim3 grid, dim3 block, size_t lds, hipStream_t st, const ScanParams &sp) {
    return launch_scan_m<DB, MAXCH, 3>(grid, block, lds, st, sp);
}
// the (DB, MAXCH) combinations the query-sharing variant is compiled for (narrow rows, k <= 100)

template <int DB>
static int launch_scan_emit(dim3 grid, dim3 block, size_t lds, hipStream_t st, const ScanParams &sp) {
    return launch_scan_m<DB, 1, 4>(grid, block, lds, st, sp);
}

static int launch_scan(int db, int maxch, dim3 grid, dim3 block, size_t lds, hipStream_t st, const ScanParams &sp) {
    if (sp.key_out) {
        if (db == 1) return launch_scan_emit<1>(grid, block, lds, st, sp);
        if (db == 2) return launch_scan_emit<2>(grid, block, lds, st, sp);
        if (db == 4) return launch_scan_emit<4>(grid, block, lds, st, sp);
        if (db == 8) return launch_scan_emit<8>(grid, block, lds, st, sp);
        return launch_scan_emit<16>(grid, block, lds, st, sp);
    }
    if (sp.qshare) {
#define QK_CASEQ(D, M) \
    if (db == D && maxch == M) return launch_scan_qs<D, M>(grid, block, lds, st, sp);
        QK_CASEQ(2, 1) QK_CASEQ(2, 2) QK_CASEQ(2, 4) QK_CASEQ(4, 1) QK_CASEQ(4, 2) QK_CASEQ(4, 4) QK_CASEQ(8, 1) QK_CASEQ(8, 2) QK_CASEQ(8, 4)
#undef QK_CASEQ
        QK_FAIL(QK_ERR_UNSUPPORTED, "no query-sharing scan kernel for DB=%d MAXCH=%d", db, maxch);
    }
    static const int probe_mode = qk_env_int("QK_SCAN_MODE", 0);
    if (probe_mode == 1 && db == 8 && maxch == 1) {
        return launch_scan_k<8, 1, 1, true>(grid, block, lds, st, sp);
    }
    if (probe_mode == 2 && db == 8 && maxch == 1) {
        return launch_scan_k<8, 1, 2, true>(grid, block, lds, st, sp);
    }
#define QK_CASE(D, M) \
    if (db == D && maxch == M) return launch_scan_t<D, M>(grid, block, lds, st, sp);
    QK_CASE(1, 1) QK_CASE(1, 2) QK_CASE(1, 4) QK_CASE(1, 8) QK_CASE(2, 1) QK_CASE(2, 2) QK_CASE(2, 4) QK_CASE(2, 8)
    QK_CASE(4, 1) QK_CASE(4, 2) QK_CASE(4, 4) QK_CASE(4, 8) QK_CASE(8, 1) QK_CASE(8, 2) QK_CASE(8, 4) QK_CASE(8, 8)
    QK_CASE(16, 1) QK_CASE(16, 2) QK_CASE(16, 4) QK_CASE(16, 8)
#undef QK_CASE
    QK_FAIL(QK_ERR_UNSUPPORTED, "no scan kernel for DB=%d MAXCH=%d", db, maxch);
}


int qk_scan_device(qk_ctx *ctx, qk_store *s, const qk_scan_args &a, qk_timing *timing, int ev_base) {
    const int64_t Q = a.Q;
    const bool emit = a.key_out != nullptr;  // key emission for qk_widek_device: no top-k, k plays no role here
    const int k = emit ? 1 : a.k;
    if (Q <= 0) return QK_OK;
    if (k <= 0) QK_FAIL(QK_ERR_INVALID, "qk_scan: k must be positive");
    // beyond the LDS pools: one list (flat / parent index) -> k_select_rows_large; several lists -> every key is emitted and
    // selected afterwards (qk_widek_device)
    const bool one_list = a.all_lists && s->nlist == 1 && !emit;
    if (k > QK_MAX_K && !one_list) {
        if (a.per_pair) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_scan: k=%d exceeds QK_MAX_K=%d", k, QK_MAX_K);
        QK_TRY(qk_prep_flush(ctx));
        return qk_widek_device(ctx, s, a, timing, ev_base);
    }
    if (k > QK_MAX_NPROBE) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_scan: k=%d exceeds %d", k, QK_MAX_NPROBE);
    QK_TRY(qk_store_sync_table(s));
    const int npids = (int)s->parts.size();
    // dense form: every query against ONE list (the parent / flat index of query_coordinator.cpp:624-626,644)
    // (any batch size: for one query the key-matrix path is 3 launches against 6 of the grouped scan -- 133 -> 95 us per search)
    static const int dense_min_q = qk_env_int("QK_DENSE_MIN_Q", 1);
    if (one_list && (Q >= dense_min_q || k > QK_MAX_K)) {
        for (int64_t p = 0; p < npids; p++)
            if (s->parts[p].present) return qk_dense_device(ctx, s, p, a, timing, ev_base);
    }
    QK_TRY(qk_prep_flush(ctx));  // (a query preparation left pending for the nearest-centroid kernel: nobody else folds it in)
    const int P = a.all_lists ? npids : a.P;
    // per-pair results: a bound learnt in one list must not prune another list's own top-k
    const bool share_tau = a.share_tau && !a.per_pair && !emit;
    hipStream_t st = ctx->stream;
    const bool tm = ctx->timing && (timing || a.record_events);
    qk_phase_events pe;
    QK_TRY(pe.begin(ctx, tm, ev_base));
    const int64_t npairs = Q * (int64_t)P;  // may be 0: pure padding (query_coordinator.cpp:459-497)
    if (npairs > 0x7FFFFFF0LL) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_scan: Q*P too large");

    // ---- geometry and form: qk_scan_plan.hip (the host rule with its measured constants, and the form feedback) -----------------
    ScanPlan plan;
    QK_TRY(qk_scan_plan(ctx, s, a, emit, k, P, npairs, &plan));
    const int nblk = s->nblk;
    const size_t q_bytes = (size_t)nblk * 1024;
    int DB = plan.DB;
    const int C = plan.C, nw = plan.nw, qshare = plan.qshare;
    const bool use_rl = plan.use_rl;
    const int64_t rl_per_list = plan.rl_per_list;
    const RlCost rlc = plan.rlc;
    const int rl_app = plan.rl_app, rl_waves = plan.rl_waves;
    const HotCost hot = plan.hot;
    qk_ctx::form_stat *fmeasure = plan.measure;
    const int maxch = pick_maxch(C);
    const size_t lds_scan = use_rl ? ((qk_scan_rl_lds_per_wave(nblk, C, rlc.qb) + 15) & ~(size_t)15)
                            : qshare ? (size_t)nw * (q_bytes + (size_t)16 * C * 12) : q_bytes + (size_t)nw * 16 * C * 12;
    const int Cm = qk_round_up(k + 64, 64);
    const int maxch_m = Cm <= 128 ? 2 : Cm <= 256 ? 4 : Cm <= 512 ? 8 : 16;
    const size_t lds_merge = (((size_t)Cm * 12 + 15) & ~(size_t)15) + (size_t)64 * QK_SLOTS * 4;  // pool + 64 pair slot lines

    int num_cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    {
        static const int spare = qk_env_int("QK_SCAN_SPARE_CUS", 0);  // probe: leave CUs to the small kernels of other streams
        if (spare > 0 && num_cus > 2 * spare) num_cus -= spare;
    }
    // persistent grid: as many single-wave workgroups as stay resident (LDS-limited; registers allow ~12 per CU); the tile
    // partition is static, so every wave must be resident at once.  Measured (bench.py --nprobe 1/8/32, QK_SCAN_WAVE_CLOCK):
    // long launches: 8 per CU (10+ lose bandwidth).  Short launches (a wave gets < ~160 tiles): 4 per CU = one wave per
    // SIMD.  With 6 or 8 the waves of a CU finish between 65 % and 100 % of the kernel time (those that share a SIMD are
    // slower, and the slowest wave of a static cut sets the time); with 4 they finish within 85-100 % and every query
    // leaves fewer, longer records: bench configuration 0.276-0.297 ms (6 per CU) -> 0.264-0.268 ms (4 per CU).
    int waves_per_cu = nw * (int)std::max<size_t>(1, std::min<size_t>(8 / nw, (160 * 1024) / (lds_scan + 512)));
    int64_t tiles_est;
    {
        const int64_t npresent_e = std::max<int64_t>(1, s->nlist);
        const int64_t mean_tiles = std::max<int64_t>(1, (s->ntotal / npresent_e + 15) / 16);
        tiles_est = std::max<int64_t>(1, npairs / 16 + std::min<int64_t>(npresent_e, npairs)) * mean_tiles;
        if (nw == 1 && waves_per_cu > 4 && tiles_est < (int64_t)8 * num_cus * 160) waves_per_cu = 4;
    }
    {
        static const int wpc = qk_env_int("QK_SCAN_WAVES_PER_CU", 0);  // probe override
        if (wpc > 0) waves_per_cu = std::max(nw, wpc / nw * nw);
    }
    if (use_rl) waves_per_cu = rl_waves;  // one wave per SIMD: 256+ registers of row data per wave
    // wide rows (d >= 256: the LDS query tile leaves room for <= 4 waves per CU): 16 blocks = 16 KB per load step, so
    // that the few resident waves still keep enough bytes in flight to cover the HBM latency
    if (nblk % 16 == 0 && waves_per_cu <= 4 && !qshare && !use_rl && !qk_env_set("QK_SCAN_NO_DB16")) DB = 16;
    const int wgs_per_cu = waves_per_cu / nw;
    const int64_t n_wgs = (int64_t)num_cus * wgs_per_cu;
    const int64_t n_waves = n_wgs * nw;
    // records: every wave-segment emits at most 16; segments <= items + waves
    const int64_t npresent = std::max<int64_t>(1, s->nlist);
    const int64_t items_bound = std::max<int64_t>(1, npairs / 16 + std::min<int64_t>(npresent, npairs));
    // (with nw waves per workgroup every segment is cut nw ways: nw records per pair and segment)
    // ... plus one segment start per dynamic chunk (k_scan's dynamic tail, nw == 1)
    int64_t dyn_ranges = 0;
    {
        static const int pct = qk_env_int("QK_SCAN_DYN_PCT", QK_DYN_PCT_DEFAULT);
        static const int chunk = std::max(1, qk_env_int("QK_SCAN_DYN_CHUNK", QK_DYN_CHUNK_DEFAULT));
        if (nw == 1 && pct > 0) {
            const int64_t tiles_all = items_bound * ((std::max<int64_t>(1, s->max_size) + 15) / 16);
            dyn_ranges = (tiles_all * std::min(90, pct) / 100) / chunk + 2;
        }
    }
    const int64_t seg_starts = n_wgs + dyn_ranges;
    // (row-per-lane form: a segment emits one record per live query of its pass; every range boundary inside a pass adds at
    //  most rlc.qb records)
    // (hot lists: a pair leaves one record per row range of its list, at most QK_HOT_NRR_MAX)
    const int64_t hot_recs = hot.min > 0 ? npairs * std::min<int64_t>(QK_HOT_NRR_MAX, hot_block_cost((int)((std::max<int64_t>(1, s->max_size) + 15) / 16), hot.hq / 16, hot) / hot.unit + 1) : 0;
    const int64_t max_recs = use_rl ? std::min<int64_t>(0x7FFFFFF0LL, npairs + hot_recs + (int64_t)rlc.qb * (n_waves + QK_RL_DYN_MAX + 2))
                                    : std::min<int64_t>(0x7FFFFFF0LL, nw * std::min<int64_t>(16 * (items_bound + seg_starts), npairs + 16 * seg_starts));

    // ---- workspace ---------------------------------------------------------------------------------
    const int64_t np1 = std::max<int64_t>(npairs, 1);
    size_t need = 0;
    auto add = [&](size_t b) { need += (b + 255) & ~(size_t)255; };
    add((size_t)npids * 4 * 2 + 256 + (size_t)Q * 4 + 64);
    add((size_t)(npids + 1) * sizeof(ActiveInfo) + 64);
    add((size_t)(npids + 1) * 4 + 64);
    add((size_t)(npids + 2) * 4 + 64);
    add((size_t)np1 * 4 * 3);
    add((size_t)np1 * QK_SLOTS * 4);
    add((size_t)(std::min<int64_t>(npids, np1) + 1) * 4 + 64);
    add((size_t)Q * 4);
    add((size_t)max_recs * 8);
    add((size_t)max_recs * k * 4);
    add((size_t)max_recs * k * 8);
    need += 8192;
    QK_TRY(qk_ws_reserve(ctx, need));
    const float4 *xq4 = a.xq4;
    const float *xn = a.xn;
    if (!xq4 || !xn) QK_FAIL(QK_ERR_INVALID, "qk_scan: queries were not prepared");
    // zeroed region: g_cnt [npids], g_cursor [npids], scal [64], gtau [Q]  (one memset per call -- or none: the query prep
    // kernel of this batch leaves a zeroed region of the size it was told to, qk_prep_queries(..., zero_bytes))
    const size_t zero_bytes = qk_scan_zero_bytes(npids, Q);
    const bool prezeroed = ctx->qprep_zero && ctx->qprep_zero_bytes == zero_bytes && a.xq4 == (const float4 *)ctx->qprep;
    int32_t *g_cnt = prezeroed ? (int32_t *)ctx->qprep_zero : (int32_t *)qk_ws_alloc(ctx, zero_bytes + 64);
    if (prezeroed) ctx->qprep_zero_bytes = 0;  // consumed
    int32_t *g_cursor = g_cnt + npids;
    int32_t *scal = g_cursor + npids;
    ActiveInfo *active = (ActiveInfo *)qk_ws_alloc(ctx, (size_t)(npids + 1) * sizeof(ActiveInfo) + 64);
    int32_t *g_qoff = (int32_t *)qk_ws_alloc(ctx, (size_t)(npids + 1) * 4 + 64);
    int32_t *act_hoff = (int32_t *)qk_ws_alloc(ctx, (size_t)(npids + 2) * 4 + 64);
    // (+ [8] hot items, [10..11] hot units (i64), [12] next hot item: the mixed sequence of the row-per-lane form)
    // scal layout (int32 units): [0] n_active, [1] rec_counter, [2..3] n_rows_unique (i64), [4..5] n_tiles (i64), [6] n_act, [7] live pairs
    int32_t *n_active = scal, *rec_counter = scal + 1;
    int64_t *n_rows_unique = (int64_t *)(scal + 2);
    int64_t *n_tiles = (int64_t *)(scal + 4);
    int32_t *grouped_q = (int32_t *)qk_ws_alloc(ctx, (size_t)np1 * 4 * 3);
    int32_t *grouped_pair = grouped_q + np1;
    int32_t *pair_head = grouped_pair + np1;
    int32_t *pair_slots = (int32_t *)qk_ws_alloc(ctx, (size_t)np1 * QK_SLOTS * 4);
    int32_t *act_list = (int32_t *)qk_ws_alloc(ctx, (size_t)(std::min<int64_t>(npids, np1) + 1) * 4 + 64);
    uint32_t *gtau = (uint32_t *)(scal + 64);
    int2 *rec_hdr = (int2 *)qk_ws_alloc(ctx, (size_t)max_recs * 8);
    uint32_t *rec_ord = (uint32_t *)qk_ws_alloc(ctx, (size_t)max_recs * k * 4);
    int64_t *rec_id = (int64_t *)qk_ws_alloc(ctx, (size_t)max_recs * k * 8);
    if (!g_cnt || !active || !g_qoff || !act_hoff || !grouped_q || !pair_slots || !act_list || !gtau || !rec_hdr || !rec_ord || !rec_id)
        QK_FAIL(QK_ERR_OOM, "qk_scan: workspace exhausted");

    QK_TRY(pe.mark(0));
    // ---- grouping -----------------------------------------------------------------------------------------
    if (!prezeroed) QK_HIP(hipMemsetAsync(g_cnt, 0, zero_bytes, st));
    GroupParams G;
    G.pids = a.all_lists ? nullptr : a.pids;
    G.pids_packed = a.all_lists ? nullptr : a.pids_packed;
    G.npairs = npairs;
    G.P = std::max(P, 1);
    G.pt_size = s->d_size;
    G.npids = npids;
    G.g_cnt = g_cnt;
    G.g_cursor = g_cursor;
    G.g_qoff = g_qoff;
    G.n_active = n_active;
    G.active = active;
    G.pt_off = s->d_off;
    G.n_tiles = n_tiles;
    G.n_rows_unique = n_rows_unique;
    G.grouped_q = grouped_q;
    G.grouped_pair = grouped_pair;
    G.pair_head = pair_head;
    // start charge of a pass in the static cut: QK_SCAN_SEG_OVH row-tile steps of a d = 128 tile (8 KB), scaled to the
    // tile of this index, times the waves of a workgroup (they all pay it at once).  Measured (10M x 128, batch 10000):
    // nprobe 32 +11 %, nprobe 8 +1 % with 8 steps; nothing beyond noise for one-wave workgroups (nprobe 1) or wide rows,
    // where the charge stays off unless QK_SCAN_SEG_OVH_ALL=1.
    static const int seg_ovh_env = qk_env_int("QK_SCAN_SEG_OVH", 8);
    static const bool seg_ovh_all = qk_env_int("QK_SCAN_SEG_OVH_ALL", 0) != 0;
    const int64_t tile_bytes = 64 * (int64_t)s->dpad;
    const int seg_ovh = use_rl ? rlc.ovh
                        : (seg_ovh_env <= 0 || !(qshare || seg_ovh_all))
                            ? 0
                            : (int)std::max<int64_t>(1, ((int64_t)seg_ovh_env * 8192 + tile_bytes / 2) / tile_bytes) * nw;
    G.qgroup = qshare ? nw : 1;
    G.seg_ovh = seg_ovh;
    G.rl = use_rl ? 1 : 0;
    G.rlc = rlc;
    G.act_list = act_list;
    G.n_act = scal + 6;  // zeroed with the other counters
    G.n_pairs_live = scal + 7;
    G.pair_slots = pair_slots;
    G.gtau = gtau;
    G.hot = hot;
    G.act_hoff = hot.min > 0 ? act_hoff : nullptr;
    G.n_hot = scal + 8;
    G.hot_units = (long long *)(scal + 10);
    static const int no_seed = qk_env_int("QK_NO_SEED", 0);
    // (measured: for k > 64 a sample bound is far looser than the bound the pools reach by themselves -- no gain, and
    //  the 64*M-row sample costs 0.1 ms at d = 768; the wider instantiations stay available for probing)
    static const int seed_max_k = qk_env_int("QK_SEED_MAX_K", 64);
    const bool seed_pairs = a.per_pair && a.seed_first && !emit;  // (not under QK_NO_SEED: the caller dropped its own bound for this one)
    // (narrow rows, 64 < k <= 128: a 128-row sample is cheap and pays at every nprobe -- 10M x 128, k = 100, ms per step without ->
    //  with: nprobe 2 0.648 -> 0.538, 4 0.743 -> 0.637, 8 1.099 -> 0.940, 32 2.04 -> 1.94; k = 70, nprobe 8 1.011 -> 0.880)
    const int seed_cap = nblk <= 8 ? std::max(seed_max_k, 128) : seed_max_k;
    const bool seeded = ((!no_seed && share_tau) || seed_pairs) && k <= std::min(seed_cap, 512) && npairs > 0 && npids > 0;
    bool fused_group = false, fused_count = false;
    if (seeded) {
        // bound seeding: for the first (nearest) partitions of every query, the k-th smallest distance of a 64-row
        // sample goes into gtau[q]
        SeedParams sd;
        sd.pids = G.pids;
        sd.pids_packed = G.pids_packed;
        sd.npairs = npairs;
        sd.P = G.P;
        sd.pt_size = s->d_size;
        sd.pt_off = s->d_off;
        sd.npids = npids;
        sd.vecs = (const float4 *)s->vecs;
        sd.norms = s->norms;
        sd.nblk = nblk;
        sd.d = s->d;
        sd.x = a.x;
        sd.xn = xn;
        sd.k = k;
        sd.metric = a.metric;
        sd.gtau = gtau;
        // which lists are sampled: the NEAREST one only (measured, nprobe 8-32 on both bench corpora: the second-nearest list's
        // sample never tightened the bound -- same scan time, 4 us of seeding less), and 128 of its rows instead of 64 when the
        // batch goes through the larger-batch launch (the scan of the mixed form is 2x slower without any seed and gains
        // 8-16 us from the larger sample; 256 rows cost the seeding what they save the scan: it is bandwidth-bound)
        static const int seed_ranks_env = qk_env_int("QK_SEED_RANKS", 1);
        sd.seed_ranks = seed_pairs ? 1 : std::min(std::max(1, seed_ranks_env), G.P);
        sd.strict_first = seed_pairs ? 1 : 0;
        // (a side stream + fork/join events was measured slower than running it in line: 45 vs 40 us group phase.
        //  Larger samples -- 128 / 256 rows, scalar or on MFMA with all tile loads in flight -- take 6-12 us off k_scan and
        //  add 10-40 us here: the kernel is a chain of five dependent memory round trips, not arithmetic.)
        const dim3 sg((unsigned)(Q * sd.seed_ranks));
        // (QK_SEED_WAVES = 2 / 4: 128 / 256-row samples at the latency of the 64-row one.  Measured on the bench: k_scan 0.254 ->
        //  0.248 / 0.240 ms, this kernel 10 -> 17 / 30 us -- a 256-row sample of 1024 partitions is 134 MB of reads, 9 % of what
        //  the scan streams.  One for one again; the default stays at 64 rows.)
        static const int seed_waves = qk_env_int("QK_SEED_WAVES", 1);
        static const bool no_small_f = qk_env_set("QK_NO_GROUP_SMALL");
        static const bool no_fuse = qk_env_set("QK_NO_GROUP_SEED");
        if (k <= 64 && seed_waves == 1 && npairs <= QK_GROUP_SMALL && !no_small_f && !no_fuse) {
            const int64_t nsw = Q * sd.seed_ranks;
            QK_HIP(hipFuncSetAttribute((const void *)k_group_seed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)QK_GROUP_SMALL_LDS));
            hipLaunchKernelGGL(k_group_seed, dim3((unsigned)(1 + (nsw + 15) / 16)), dim3(1024), group_small_lds(npairs), st, G, sd, nsw);
            fused_group = true;
        } else if (k <= 64 && seed_waves == 1 && npairs > QK_GROUP_SMALL && !no_fuse) {
            const int64_t nsw = Q * sd.seed_ranks;
            const int ncb = (int)((npairs + 255) / 256);
            static const int seed_m2 = qk_env_int("QK_SEED_M2", 1);
            // (128 rows pay at nprobe 2-4 -- scan 0.339 / 0.385 -> 0.318 / 0.370 ms, seeding +4 us; from nprobe 8 on the larger sample costs
            //  the seeding what it saves the scan)
            if (seed_m2 && G.P > 1 && G.P <= 4)
                hipLaunchKernelGGL(k_group_count_seed<2>, dim3((unsigned)(ncb + (nsw + 3) / 4)), dim3(256), 0, st, G, sd, ncb, nsw);
            else
                hipLaunchKernelGGL(k_group_count_seed<1>, dim3((unsigned)(ncb + (nsw + 3) / 4)), dim3(256), 0, st, G, sd, ncb, nsw);
            fused_count = true;
        } else if (k <= 64 && seed_waves == 4)
            hipLaunchKernelGGL((k_seed_tau_wg<4>), sg, dim3(256), 0, st, sd);
        else if (k <= 64 && seed_waves == 2)
            hipLaunchKernelGGL((k_seed_tau_wg<2>), sg, dim3(128), 0, st, sd);
        else if (k <= 64)
            hipLaunchKernelGGL((k_seed_tau<1>), sg, dim3(64), 0, st, sd);
        else if (k <= 128) {
            // (round 6, profiles/r06_ab_k100.jsonl: 10M x 128, k = 100, ms per step with a 128 / 256 / 512-row sample -- nprobe 2: 0.535 /
            //  0.531 / 0.582, nprobe 8: 0.938 / 0.934 / 0.982, nprobe 32: 1.936 / 1.849 / 1.782: 256 rows, 512 from nprobe 16 on)
            if (QK_SEED_M_WIDE == 8 || (QK_SEED_M_WIDE == 0 && G.P >= 16)) hipLaunchKernelGGL((k_seed_tau<8>), sg, dim3(64), 0, st, sd);
            else if (QK_SEED_M_WIDE == 2) hipLaunchKernelGGL((k_seed_tau<2>), sg, dim3(64), 0, st, sd);
            else hipLaunchKernelGGL((k_seed_tau<4>), sg, dim3(64), 0, st, sd);
        } else if (k <= 256)
            hipLaunchKernelGGL((k_seed_tau<4>), sg, dim3(64), 0, st, sd);
        else
            hipLaunchKernelGGL((k_seed_tau<8>), sg, dim3(64), 0, st, sd);
    }
    static const bool no_small = qk_env_set("QK_NO_GROUP_SMALL");
    if (fused_group) {
        // grouped by workgroup 0 of k_group_seed
    } else if (npairs <= QK_GROUP_SMALL && !no_small) {
        QK_HIP(hipFuncSetAttribute((const void *)k_group_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)QK_GROUP_SMALL_LDS));
        hipLaunchKernelGGL(k_group_small, dim3(1), dim3(1024), group_small_lds(npairs), st, G);
    } else {
        if (npairs > 0 && !fused_count) hipLaunchKernelGGL(k_group_count, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, G);
        hipLaunchKernelGGL(k_group_scan, dim3(1), dim3(1024), 0, st, G);
        if (npairs > 0) hipLaunchKernelGGL(k_group_scatter, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, G);
    }
    QK_TRY(pe.mark(1));

    // ---- scan ------------------------------------------------------------------------------------------------
    if (npairs > 0 && npids > 0) {
        ScanParams sp;
        sp.vecs = (const float4 *)s->vecs;
        sp.norms = s->norms;
        sp.ids = s->ids;
        sp.pt_off = s->d_off;
        sp.pt_size = s->d_size;
        sp.nblk = nblk;
        sp.xq4 = xq4;
        sp.xn = xn;
        sp.grouped_q = grouped_q;
        sp.grouped_pair = grouped_pair;
        sp.n_active = n_active;
        sp.active = active;
        sp.n_tiles = n_tiles;
        sp.gtau = share_tau ? gtau : nullptr;
        static const int force_refresh = qk_env_int("QK_SCAN_TAU_REFRESH", -1);
        sp.tau_refresh = P > 1 ? 1 : 0;  // one partition per query: the seed is all there is to share
        if (force_refresh >= 0) sp.tau_refresh = force_refresh;
        sp.tau_publish = 1;
        if (!share_tau && a.tau_init) {  // caller-provided per-query bound (same ~bound format), never updated here
            sp.gtau = const_cast<uint32_t *>(a.tau_init);
            sp.tau_refresh = 0;
            sp.tau_publish = 0;
        }
        if (seed_pairs && seeded) {  // the sample bound of the first list, read once per pool and never updated
            sp.gtau = gtau;
            sp.tau_refresh = 0;
            sp.tau_publish = 0;
        }

        static const int probe_tau0 = qk_env_int("QK_SCAN_TAU0", 0);
        if (probe_tau0) {  // probe: a bound of 0 -> nothing ever passes (isolates the steady-state epilogue cost)
            QK_HIP(hipMemsetAsync(gtau, 0xFF, (size_t)Q * 4, st));
            sp.gtau = gtau;
        }
        sp.k = k;
        sp.C = C;
        sp.metric = a.metric;
        sp.pair_head = pair_head;
        sp.qshare = qshare;
        sp.seg_ovh = seg_ovh;
        sp.xp4 = ctx->qprep_xp4;
        sp.rl_h0 = rlc.h0;
        sp.rl_h1 = rlc.h1;
        sp.rl_e = rlc.e;
        sp.rl_m = rlc.m;
        sp.rl_qb = rlc.qb;
        sp.rl_app = rl_app;
        static const int rl_probe = qk_env_int("QK_SCAN_RL_PROBE", 0);
        sp.rl_probe = rl_probe;
        sp.hot = hot;
        sp.act_hoff = act_hoff;
        sp.n_hot = scal + 8;
        sp.hot_units = (const long long *)(scal + 10);
        sp.hot_counter = scal + 12;  // zeroed with the counters
        static const int hot_first_pct = qk_env_int("QK_SCAN_HOT_FIRST_PCT", 300);
        sp.hot_first_pct = std::max(1, hot_first_pct);
        sp.key_out = a.key_out;
        sp.pair_base = a.pair_base;
        sp.pair_slots = pair_slots;
        sp.rec_counter = rec_counter;
        sp.max_recs = (int32_t)max_recs;
        sp.overflow = ctx->overflow_dev;
        sp.rec_hdr = rec_hdr;
        sp.rec_ord = rec_ord;
        sp.rec_id = rec_id;
        // do not launch (many) more waves than there are tiles to hand out
        int64_t tiles_ub = std::max<int64_t>(1, items_bound * ((std::max<int64_t>(1, s->max_size) + 15) / 16));
        int64_t grid = std::max<int64_t>(1, std::min<int64_t>(n_wgs, tiles_ub));
        // every wave must be resident at once AND evenly spread: pad the LDS request so that exactly waves_per_cu
        // workgroups fit on a CU (the dispatcher otherwise packs up to 10 on some CUs and leaves others short, and with
        // a static partition the slowest CU sets the kernel time)
        // (measured +2 % at 6 and 8 per CU; at 5 per CU the padded request only fits 4 -- pad the tested counts only)
        size_t lds_launch = lds_scan;
        if (wgs_per_cu == 4 || wgs_per_cu == 6 || wgs_per_cu == 8 || (nw > 1 && wgs_per_cu <= 2) || (use_rl && wgs_per_cu == 3))
            lds_launch = std::max<size_t>(lds_scan, (size_t)(160 * 1024) / wgs_per_cu - 512);
        if (use_rl) lds_launch &= ~(size_t)15;  // (per-wave slices of one allocation: float4 reads need the alignment)
        // dynamic tail: measured wave end times spread over 65-100 % of the kernel with a purely static cut
        static const int dyn_pct = qk_env_int("QK_SCAN_DYN_PCT", QK_DYN_PCT_DEFAULT);
        static const int dyn_chunk = qk_env_int("QK_SCAN_DYN_CHUNK", QK_DYN_CHUNK_DEFAULT);
        sp.dyn_counter = (nw == 1 && dyn_pct > 0 && !use_rl) ? (unsigned long long *)(scal + 16) : nullptr;  // zeroed with the counters
        sp.dyn_chunk = std::max(1, dyn_chunk);
        sp.dyn_pct = std::min(90, std::max(0, dyn_pct));
        if (use_rl) {  // row-per-lane form: dynamic tail in ranges of rl_dyn_chunk units (>= tail / QK_RL_DYN_MAX)
            // (share of the sequence handed out dynamically: 40 % when partitions are mostly probed by one query -- the cost
            //  model has little to say there and the tail evens out XCD / placement differences --, 25 % otherwise)
            // (re-checked on the final kernels, 25 / 40 / 50 % x ranges of 16 / 32 / 64 units -- nprobe 8 within 1 %
            //  everywhere, nprobe 16 on the skewed mixture best at 25 % (0.610 against 0.630 ms at 40 %), the uniformly probed corpus best at
            //  40 % / 16 units (0.778 against 0.793 ms): no setting wins both, the rule stays)
            static const int rl_dyn_env = qk_env_int("QK_SCAN_RL_DYN_PCT", -1);
            // (small launches -- under 1024 pairs, a wave's static share is a chunk or two -- finish sooner without a tail to
            //  claim: 64 queries x nprobe 10: scan 60 -> 55 us, 8 queries: 46 -> 25 us; from 2560 pairs on the tail pays)
            // (mixed form: half of the walk -- the hot-first workgroups join it when the items are gone, see hot_first_pct)
            const int rl_dyn_pct = rl_dyn_env >= 0 ? rl_dyn_env : (npairs < 1024 ? 0 : hot.min > 0 ? 50 : rl_per_list <= 1 ? 40 : 25);
            static const int rl_dyn_chunk = qk_env_int("QK_SCAN_RL_DYN_CHUNK", 64);
            sp.dyn_counter = rl_dyn_pct > 0 ? (unsigned long long *)(scal + 16) : nullptr;
            sp.dyn_chunk = std::max(1, rl_dyn_chunk);
            sp.dyn_pct = std::min(90, std::max(0, rl_dyn_pct));
        }
        // single-wave workgroups are bundled four to a hardware workgroup: one wave per SIMD, guaranteed (see k_scan)
        static const bool no_pack = qk_env_set("QK_SCAN_NO_PACK");
        const bool pack4 = use_rl || (nw == 1 && !qshare && !no_pack && (wgs_per_cu == 4 || wgs_per_cu == 8) && lds_launch % 16 == 0);
        const int pk = use_rl ? rl_waves : 4;
        sp.pack = pack4 ? pk : 1;
        sp.pack_lds = (int)lds_launch;
        const int wpw = pack4 ? pk : nw;  // waves per hardware workgroup
        if (pack4) {
            grid = (grid + pk - 1) / pk;
            lds_launch *= pk;
        }
        // ---- XCD balance (see qk_ctx::xcd_state) ---------------------------------------------------------------------------
        static const int xcd_adapt = qk_env_int("QK_SCAN_XCD_ADAPT", 1);
        const bool xcd_ready = ctx->xcd_pending && hipEventQuery(ctx->xcd_ev) == hipSuccess;
        if (ctx->xcd_pending && !xcd_ready) (void)hipGetLastError();  // "not ready" is not an error: keep it out of the later checks
        if (xcd_ready) {  // a finished sample: speed = share / time
            ctx->xcd_pending = false;
            qk_ctx::xcd_state &xs = ctx->xcd[ctx->xcd_key];
            double sp8[8], mean = 0;
            int nz = 0;
            for (int c = 0; c < 8; c++) {
                const double cnt = (double)ctx->xcd_host[8 + c], ticks = (double)ctx->xcd_host[c];
                sp8[c] = (cnt > 0 && ticks > 0) ? ctx->xcd_wsnap[c] / (ticks / cnt) : 0.0;
                if (sp8[c] > 0) {
                    mean += sp8[c];
                    nz++;
                }
            }
            if (nz == 8) {
                mean /= 8;
                double sum = 0;
                for (int c = 0; c < 8; c++) {
                    const double target = std::min(1.7, std::max(0.3, sp8[c] / mean));
                    xs.w[c] = 0.5 * xs.w[c] + 0.5 * target;
                    sum += xs.w[c];
                }
                for (int c = 0; c < 8; c++) xs.w[c] *= 8.0 / sum;
                xs.samples++;
            }
        }
        sp.xcd_on = 0;
        sp.xcd_stat = nullptr;
        for (int c = 0; c < 8; c++) sp.xcd_w[c] = 1024;
        if (xcd_adapt && (sp.dyn_counter == nullptr || use_rl) && grid >= 64 && tiles_est >= (int64_t)grid * wpw * 32) {
            qk_ctx::xcd_state &xs = ctx->xcd[s->uid];
            sp.xcd_on = 1;
            for (int c = 0; c < 8; c++) sp.xcd_w[c] = std::max(1, (int)(xs.w[c] * 1024.0 + 0.5));
            const long long nl = xs.launches++;
            const bool sample = !ctx->xcd_pending && (xs.samples < 8 || (nl & 63) == 0);  // 8 completed samples, then 1 in 64
            if (sample) {
                if (!ctx->xcd_host) QK_HIP(hipHostMalloc((void **)&ctx->xcd_host, 16 * sizeof(unsigned long long)));
                if (!ctx->xcd_ev) QK_HIP(hipEventCreateWithFlags(&ctx->xcd_ev, hipEventDisableTiming));
                sp.xcd_stat = (unsigned long long *)(scal + 32);  // zeroed with the counters
                ctx->xcd_key = s->uid;
                for (int c = 0; c < 8; c++) ctx->xcd_wsnap[c] = sp.xcd_w[c] / 1024.0;
            }
        }
        static const bool probe_clock = qk_env_set("QK_SCAN_WAVE_CLOCK");
        static long long *d_clock = nullptr;
        sp.wave_clock = nullptr;
        if (probe_clock) {
            if (!d_clock) QK_HIP(hipMalloc((void **)&d_clock, (size_t)1 << 20));
            QK_HIP(hipMemsetAsync(d_clock, 0, (size_t)grid * wpw * 64 * 2, st));
            sp.wave_clock = d_clock;
        }
        ctx->last_scan_kernel = use_rl ? (hot.min > 0 ? "k_scan_rl (mixed)" : "k_scan_rl") : qshare ? "k_scan (query-sharing)" : "k_scan";
        if (use_rl)
            QK_TRY(qk_launch_scan_rl(nblk, dim3((unsigned)grid), lds_launch, st, sp));
        else
            QK_TRY(launch_scan(DB, maxch, dim3((unsigned)grid), dim3(64 * wpw), lds_launch, st, sp));
        if (sp.xcd_stat) {
            QK_HIP(hipMemcpyAsync(ctx->xcd_host, sp.xcd_stat, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            QK_HIP(hipEventRecord(ctx->xcd_ev, st));
            ctx->xcd_pending = true;
        }
#ifdef QK_PROBES
        if (probe_clock) {  // debug probe: distribution of the waves' busy time (tail = what a dynamic split could recover)
            std::vector<long long> h((size_t)grid * wpw * 8);
            QK_HIP(hipMemcpyAsync(h.data(), d_clock, h.size() * 8, hipMemcpyDeviceToHost, st));
            int32_t hscal[8];
            QK_HIP(hipMemcpyAsync(hscal, scal, sizeof(hscal), hipMemcpyDeviceToHost, st));
            QK_HIP(hipStreamSynchronize(st));
            {
                int64_t units, rows;
                std::memcpy(&units, hscal + 4, 8);
                std::memcpy(&rows, hscal + 2, 8);
                fprintf(stderr, "[k_scan launch] pack=%d grid=%lld nw=%d qshare=%d seg_ovh=%d active=%d records=%d unique_rows=%lld sequence_units=%lld\n",
                        sp.pack, (long long)grid, nw, (int)qshare, seg_ovh, hscal[0], hscal[1], (long long)rows, (long long)units);
                fprintf(stderr, "[k_scan xcd weights] on=%d %d %d %d %d %d %d %d %d\n", sp.xcd_on, sp.xcd_w[0], sp.xcd_w[1], sp.xcd_w[2], sp.xcd_w[3],
                        sp.xcd_w[4], sp.xcd_w[5], sp.xcd_w[6], sp.xcd_w[7]);
                fprintf(stderr, "[k_scan params] DB=%d maxch=%d C=%d k=%d lds=%zu max_recs=%lld gtau=%d refresh=%d publish=%d npairs=%lld Q=%lld ws=%p vecs=%p\n",
                        DB, maxch, sp.C, sp.k, lds_launch, (long long)max_recs, sp.gtau != nullptr, sp.tau_refresh, sp.tau_publish,
                        (long long)npairs, (long long)Q, (void *)g_cnt, (void *)sp.vecs);
            }
            long long t0 = LLONG_MAX, t1 = 0;
            const size_t nwv = h.size() / 8;
            for (size_t i = 0; i < nwv; i++) {
                if (h[8 * i + 1] == 0) continue;
                t0 = std::min(t0, h[8 * i]);
                t1 = std::max(t1, h[8 * i + 1]);
            }
            {  // mean end time by XCD (workgroups go round-robin over the 8 XCDs) and the slowest workgroups
                double xs[8] = {0}, xn[8] = {0};
                std::vector<std::pair<long long, int>> byend;
                for (size_t i = 0; i < nwv; i++) {
                    if (h[8 * i + 1] == 0) continue;
                    xs[(i / wpw) & 7] += (double)(h[8 * i + 1] - t0);
                    xn[(i / wpw) & 7] += 1;
                    byend.push_back({h[8 * i + 1] - t0, (int)i});
                }
                fprintf(stderr, "[k_scan xcd] mean end by blockIdx%%8:");
                for (int x = 0; x < 8; x++) fprintf(stderr, " %.0f", xs[x] / std::max(1.0, xn[x]));
                std::sort(byend.begin(), byend.end());
                fprintf(stderr, "\n[k_scan slowest] workgroup:start+duration");
                for (size_t i = byend.size() >= 24 ? byend.size() - 24 : 0; i < byend.size(); i++) {
                    const int w = byend[i].second;
                    fprintf(stderr, " %d:%lld+%lld(c%lld a%lld s%lld e%lld q%lld)", w, h[8 * (size_t)w] - t0, h[8 * (size_t)w + 1] - h[8 * (size_t)w],
                            h[8 * (size_t)w + 2], h[8 * (size_t)w + 3], h[8 * (size_t)w + 4], h[8 * (size_t)w + 5], h[8 * (size_t)w + 6]);
                }
                fprintf(stderr, "\n");
                // placement: waves per (xcc, se, sh, cu, simd); how many of the slowest 10 % share their SIMD with another wave
                auto place = [&](int w, bool with_simd) {
                    const unsigned long long v = (unsigned long long)h[8 * (size_t)w + 7];
                    const unsigned hw = (unsigned)(v & 0xFFFFFFFFu), xcc = (unsigned)(v >> 32) & 15u;
                    const unsigned simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
                    return (((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4) + (with_simd ? simd : 0)) * 2 + (with_simd ? 1 : 0);
                };
                std::map<unsigned, int> per_simd, per_cu;
                for (auto &e : byend) {
                    per_simd[place(e.second, true)]++;
                    per_cu[place(e.second, false)]++;
                }
                int hist_simd[8] = {0}, hist_cu[16] = {0};
                for (auto &kv : per_simd) hist_simd[std::min(kv.second, 7)]++;
                for (auto &kv : per_cu) hist_cu[std::min(kv.second, 15)]++;
                fprintf(stderr, "[k_scan placement] SIMDs holding 1/2/3/4 waves: %d %d %d %d; CUs holding 1..8 waves:", hist_simd[1], hist_simd[2], hist_simd[3], hist_simd[4]);
                for (int c = 1; c <= 8; c++) fprintf(stderr, " %d", hist_cu[c]);
                int slow_shared = 0, slow_n = 0, fast_shared = 0, fast_n = 0;
                for (size_t i = 0; i < byend.size(); i++) {
                    const bool shared = per_simd[place(byend[i].second, true)] > 1;
                    if (i >= byend.size() - byend.size() / 10) { slow_n++; slow_shared += shared; }
                    else { fast_n++; fast_shared += shared; }
                }
                fprintf(stderr, "; waves sharing a SIMD: %d of the slowest %d, %d of the other %d\n", slow_shared, slow_n, fast_shared, fast_n);
            }
            struct W { long long end, comp, app, seg, tend, tstage; };
            std::vector<W> ws;
            double sum = 0;
            for (size_t i = 0; i < nwv; i++) {
                if (h[8 * i + 1] == 0) continue;
                ws.push_back({h[8 * i + 1] - t0, h[8 * i + 2], h[8 * i + 3], h[8 * i + 4], h[8 * i + 5], h[8 * i + 6]});
                sum += (double)(h[8 * i + 1] - t0);
            }
            std::sort(ws.begin(), ws.end(), [](const W &a, const W &b) { return a.end < b.end; });
            if (use_rl && hot.min > 0) {  // mixed sequence: phases of the hot items (second half of the probe buffer)
                std::vector<long long> hh((size_t)grid * wpw * 8);
                QK_HIP(hipMemcpy(hh.data(), d_clock + (size_t)grid * wpw * 8, hh.size() * 8, hipMemcpyDeviceToHost));
                double t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nf = 0;
                for (size_t i = 0; i < (size_t)grid * wpw; i++) {
                    for (int c = 0; c < 8; c++) t[c] += (double)hh[8 * i + c];
                }
                fprintf(stderr, "[k_scan_rl prefilter] %.0f row-tile x query-tile products tested, %.0f recomputed exactly (%.3f)\n", t[6], t[7], t[7] / std::max(1.0, t[6]));
                const double ni = std::max(1.0, t[5]);
                fprintf(stderr, "[k_scan_rl hot] items per wave %.1f (hot-first waves %.0f); ticks per item and wave: staging %.0f, chains+keys+appends %.0f, "
                        "wait for the workgroup %.0f, emission %.0f, whole item %.0f; hot share of the wave time %.2f\n",
                        t[5] / (grid * wpw), nf, t[0] / ni, t[1] / ni, t[2] / ni, t[3] / ni, t[4] / ni, t[4] / std::max(1.0, sum));
            }
            if (use_rl) {  // row-per-lane form: field 5 = shader cycles, field 6 = MFMA loop steps
                double cyc = 0, tk = 0, steps = 0;
                for (size_t i = 0; i < nwv; i++) {
                    if (h[8 * i + 1] == 0) continue;
                    cyc += (double)h[8 * i + 5];
                    tk += (double)(h[8 * i + 1] - h[8 * i]);
                    steps += (double)h[8 * i + 6];
                }
                fprintf(stderr, "[k_scan_rl] effective shader clock %.3f GHz, %.0f MFMA loop steps in all, %.0f cycles of wave time per step\n",
                        tk > 0 ? cyc / tk * 0.1 : 0.0, steps, steps > 0 ? cyc / steps : 0.0);
            }
            if (!ws.empty()) {
                const size_t n = ws.size();
                fprintf(stderr, "[k_scan waves] n=%zu span=%lld ticks  end-time pct: p10=%lld p50=%lld p90=%lld p99=%lld max=%lld mean=%.0f (100 MHz ticks)\n",
                        n, t1 - t0, ws[n / 10].end, ws[n / 2].end, ws[n * 9 / 10].end, ws[n * 99 / 100].end, ws.back().end, sum / n);
                for (int dec = 0; dec < 10; dec++) {  // per decile of end time: mean compactions / appends / segment starts
                    double c = 0, a_ = 0, sg = 0, e = 0, te = 0, tsg = 0;
                    size_t lo = n * dec / 10, hi = n * (dec + 1) / 10;
                    for (size_t i = lo; i < hi; i++) { c += ws[i].comp; a_ += ws[i].app; sg += ws[i].seg; e += ws[i].end; te += ws[i].tend; tsg += ws[i].tstage; }
                    const double m = (double)std::max<size_t>(1, hi - lo);
                    fprintf(stderr, "   decile %d: end=%.0f compactions=%.1f appends=%.0f segments=%.2f seg_end_ticks=%.0f staging_ticks=%.0f\n", dec, e / m, c / m, a_ / m, sg / m, te / m, tsg / m);
                }
            }
        }
#endif
    }
    QK_TRY(pe.mark(2));

    if (emit) {  // the caller selects from the emitted keys
        QK_TRY(pe.mark(3));
        return QK_OK;
    }
    // ---- merge ---------------------------------------------------------------------------------------------------
    MergeParams mp;
    mp.P = a.per_pair ? 1 : P;  // per_pair: every (query, list) pair is merged on its own
    mp.pair_head = pair_head;
    mp.pair_slots = pair_slots;
    mp.rec_hdr = rec_hdr;
    mp.rec_ord = rec_ord;
    mp.rec_id = rec_id;
    mp.max_recs = (int32_t)max_recs;
    mp.k = k;
    mp.Cm = Cm;
    mp.metric = a.metric;
    mp.out_ids = a.out_ids;
    mp.out_dist = a.out_dist;
    mp.sqrt_l2 = a.sqrt_l2 ? 1 : 0;
    const dim3 mgrid((unsigned)(a.per_pair ? Q * P : Q));
    QK_TRY(qk_launch_merge(ctx, mp, mgrid));
    if (fmeasure && hipEventRecord(fmeasure->e1, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        fmeasure->pending = -1;
    }
    QK_TRY(pe.mark(3));
    if (timing) {
        // device scalars come back through pinned memory; the caller synchronises before reading them
        QK_TRY(qk_pinned_reserve(ctx, 64));
        QK_HIP(hipMemcpyAsync(ctx->pinned, scal, 32, hipMemcpyDeviceToHost, st));
    }
    return QK_OK;
}

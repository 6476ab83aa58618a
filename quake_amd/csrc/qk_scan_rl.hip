// qk_scan_rl.hip -- the partition scan in ROW-PER-LANE form: v_mfma_f32_4x4x1_16b_f32, 64 rows x 4 queries per instruction.
//
// Same contract as k_scan (qk_scan.hip): distance of every (query, probed partition) pair's rows on k-ordered fp32 chains +
// fused top-k, records chained to the pair, merged by k_merge.  Replaces scan_list / batched_scan_list + TopkBuffer
// (src/cpp/include/list_scanning.h:41-204,241-366) inside serial_scan / batched_serial_scan
// (src/cpp/src/query_coordinator.cpp:471-611,675-799).
//
// Why a second form.  k_scan feeds 16 rows x 16 queries to v_mfma_f32_16x16x4_f32: a partition probed by q queries of the
// batch costs ceil(q/16) full MFMA tiles per 16 rows whatever q is.  With realistic nprobe (8-32) a probed partition is
// shared by 5-20 queries of a 1024-query batch, 40 % of the query slots are live, and the launch turns MFMA-bound while
// streaming every shared partition once per 16-query tile.  Here the 16 blocks of the 4x4x1 form are
// 16 row groups of 4 rows against the SAME 4 queries: one instruction = 64 rows x 4 queries x 1 column at the same
// 64 FLOP/clk/SIMD, so the matrix work follows the live queries in steps of 4 and a pass over a partition serves up to
// RlCost::qb (32-64) queries from ONE read of its rows.  k = 1 per instruction: the accumulator chain is literally
// acc = fma(row[k], query[k], acc) in column order -- the canonical arithmetic, bit-identical to k_scan.
//
// Operand layout (scripts/micro/mfma_4x4x1.hip prints it from the hardware and checks the chain against fmaf):
//   D[lane 4b + j][register i] = A(lane 4b + i) * B(lane 4b + j); with CBSZ = 4 / ABID = n the A values of block n feed all
//   16 blocks: D[lane l][register i] = A(lane 4n + i) * B(lane l).
//   B: lane l holds the value of ROW l of the 64-row chunk, column k
//   A: the four QUERIES of the group; lane (b, i) keeps q_i[16c + b] in register c, so that column k = 16c + n is the
//      instruction (register c, ABID = n): a group's queries live in d/16 VGPRs and the inner loop reads no LDS at all
//      (round 2, first form: queries as B operands re-read from LDS, one ds_read_b128 per 4 MFMAs -- 64 KB per wave and
//      loop step, 77 % of the CU's LDS bandwidth with four waves: the loop step took 4700 cycles instead of 2700)
//   D: lane l, register i = row l x query i of the group: a lane owns ONE row (its norm, its id) and four queries
// Rows come straight from the tile-major arena: lane l = (tile l/16 of the chunk, row l%16) loads, for every 16-column block c
// and k-slice g, the float4 {columns 16c+g, +4, +8, +12} -- 4 x 256 contiguous bytes per instruction -- and element t of it
// is the B operand of column 16c + 4t + g.  Queries are staged per pass in LDS as [group][c][lane] floats (conflict-free
// ds_read_b32, d/16 per group and chunk).
#include "qk_internal.h"
#include "qk_device.h"
#include "qk_scan_types.h"

#include <algorithm>

__device__ __forceinline__ float4 rl_ld_nt(const float4 *p) {
    const f32x4 t = __builtin_nontemporal_load((const f32x4 *)p);
    return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ float f4c(const float4 &v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }

// NB = 16-column blocks per row (d <= 128).  One hardware workgroup = 4 independent waves (own range, own LDS, no barrier).
// (the metric is a template parameter: as a runtime flag it left uniform branches around every result element of the epilogue)
template <int NB, bool L2>
__global__ __launch_bounds__(256) void k_scan_rl(ScanParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NKK = NB * 4;        // float4 (4 consecutive columns) per padded row
    constexpr int NGMAX = QK_RL_QB_MAX / 4;
    const int QB = P.rl_qb;            // query slots per pass (lane s owns slot s)
    const int NG = QB >> 2;            // groups per pass
    const int APP = P.rl_app;          // lanes per append round: a pool at most k long has room for APP more
    const int lane = threadIdx.x & 63, wvp = threadIdx.x >> 6;
    const int qi = lane & 3, b4 = lane >> 2;  // A: lane (block b4, query qi of the group) keeps q_qi[16c + b4] in register c
    const int tq = lane >> 4, r = lane & 15;  // B / D: lane = row of the chunk = (tile tq, row r)
    const int C = P.C, k = P.k;
    constexpr bool l2 = L2;
    unsigned char *sm = smem + (size_t)wvp * P.pack_lds;
    float *sQ = (float *)sm;                                              // [NG][NB][64]
    int64_t *pool_id = (int64_t *)(sm + (size_t)NG * NB * 64 * 4);        // [QB][C]
    uint32_t *pool_ord = (uint32_t *)((unsigned char *)pool_id + (size_t)QB * C * 8);  // [QB][C]
    int *s_q = (int *)(pool_ord + (size_t)QB * C);                        // per slot: query, pair, bound, pool fill, |x|^2
    int *s_pair = s_q + QB;
    uint32_t *s_tau = (uint32_t *)(s_pair + QB);
    int *s_cnt = (int *)(s_tau + QB);
    float *s_xn = (float *)(s_cnt + QB);

    // ---- this wave's contiguous share of the work sequence (units of RlCost; XCD-weighted like k_scan's) -----------------
    // The first (100 - dyn_pct) % of the sequence is cut statically; the tail is claimed in ranges of dyn_chunk units through
    // one atomic counter by whoever finishes first (the cost model cannot know how much HBM bandwidth a wave will get while
    // others are in MFMA-bound passes)
    const long long T = *P.n_tiles;
    const int PK = P.pack;  // waves per hardware workgroup (4: one per SIMD; 3 when a wider pass needs the LDS)
    const long long W = (long long)gridDim.x * PK;
    const long long vblock = (long long)blockIdx.x * PK + wvp;
    const bool dyn = P.dyn_counter != nullptr;
    const long long Ts = dyn ? T - (T * P.dyn_pct) / 100 : T;
    long long T0 = (Ts * vblock) / W, T1 = (Ts * (vblock + 1)) / W;
    if (P.xcd_on) {
        long long pre[9];
        pre[0] = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) pre[i + 1] = pre[i] + P.xcd_w[i];
        const long long u = blockIdx.x, gu = gridDim.x;
        const long long total = ((gu >> 3) * pre[8] + pre[gu & 7]) * PK;
        const long long a0 = ((u >> 3) * pre[8] + pre[u & 7]) * PK + (long long)P.xcd_w[u & 7] * wvp;
        const long long a1 = a0 + P.xcd_w[u & 7];
        T0 = a0 <= 0 ? 0 : (long long)((double)Ts * (double)a0 / (double)total);
        T1 = a1 >= total ? Ts : (long long)((double)Ts * (double)a1 / (double)total);
    }
    if (!dyn && T1 <= T0) return;
    const long long wc0 = (P.xcd_stat || P.wave_clock) ? wall_clock64() : 0;
    const long long cy0 = P.wave_clock ? clock64() : 0;
    int dbg_comp = 0, dbg_app = 0, dbg_seg = 0;  // probe counters (QK_SCAN_WAVE_CLOCK)
    long long dbg_t_end = 0, dbg_t_stage = 0, dbg_gc = 0;
    const RlCost rc{P.rl_h0, P.rl_h1, P.rl_e, P.seg_ovh, P.rl_m, QB};
    const int n_active = *P.n_active;
    long long xcd_ticks = 0;  // time spent on the static share (what the XCD balance learns from)
    for (;;) {
    if (T1 > T0) {
    // 64-ary search for the partition that holds unit T0: active[lo].toff <= T0 < active[lo+1].toff
    int lo = 0, hi = n_active;
    while (hi - lo > 1) {
        const int span = hi - lo;
        const int step = (span + 63) >> 6;
        const int probe = min(lo + (lane + 1) * step, hi);
        const bool gt = (probe >= hi) || (P.active[probe].toff > T0);
        const uint64_t m = __ballot(gt);
        const int first = __ffsll((unsigned long long)m) - 1;
        const int nlo = min(lo + first * step, hi - 1);
        const int nhi = min(lo + (first + 1) * step, hi);
        lo = nlo;
        hi = nhi;
    }
    int ai = lo;
    long long cur = T0;

    while (cur < T1) {
        // ---- segment = chunks [ch0, ch1) of pass b over partition active[ai] ---------------------------------------------
        const ActiveInfo inf = P.active[ai];
        const int size_p = inf.size, cnt_p = inf.cnt;
        const int nch = (size_p + 63) >> 6, ntl = (size_p + 15) >> 4;
        const int nqb = (cnt_p + QB - 1) / QB;
        const int g_last = (cnt_p - QB * (nqb - 1) + 3) >> 2;
        const long long local = cur - inf.toff;
        int b = 0, w;
        long long boff = 0, blen;
        for (;;) {
            w = rl_w(b == nqb - 1 ? g_last : NG, b == 0, rc);
            blen = rc.ovh + (long long)nch * w;
            if (b >= nqb - 1 || local < boff + blen) break;
            boff += blen;
            b++;
        }
        const long long off = local - boff;
        const long long off_end = min(blen, off + (T1 - cur));
        // a chunk belongs to the range that holds its first unit
        const int ch0 = (int)((max(0ll, off - rc.ovh) + w - 1) / w);
        const int ch1 = (int)((max(0ll, off_end - rc.ovh) + w - 1) / w);
        cur += off_end - off;
        if (b >= nqb - 1 && off_end == blen) ai++;  // partition exhausted
        if (ch1 <= ch0) continue;

        const int nq = min(QB, cnt_p - QB * b);
        const int ng = (nq + 3) >> 2;
        const int gidx = inf.qoff + QB * b + lane;
        // lanes 0..31 own the slot of their number: query and pair of the slot (requested first, they return first)
        const int myq = (lane < nq) ? P.grouped_q[gidx] : -1;
        const int mypair = (lane < nq) ? P.grouped_pair[gidx] : -1;
        const int64_t tile_p0 = inf.row_off >> 4;

        float4 a0[NB * 4], a1[NB * 4];
        float y0, y1;      // |row|^2 and id of this lane's row
        int64_t i0, i1;
        int lch = ch0;  // next chunk to load
#define RL_LOAD(A, Y, I)                                                                              \
    {                                                                                                 \
        const int64_t ta_ = tile_p0 + min(4 * lch + tq, ntl - 1);                                     \
        const float4 *src_ = P.vecs + ta_ * (NB * 64) + r;                                            \
        _Pragma("unroll") for (int c_ = 0; c_ < NB; c_++)                                             \
            _Pragma("unroll") for (int g_ = 0; g_ < 4; g_++) A[c_ * 4 + g_] = rl_ld_nt(src_ + c_ * 64 + g_ * 16); \
        Y = P.norms[(ta_ << 4) + r];                                                                  \
        I = P.ids[(ta_ << 4) + r];                                                                    \
        lch++;                                                                                        \
    }
        RL_LOAD(a0, y0, i0);
        dbg_seg++;
        const long long dbg_s0 = P.wave_clock ? wall_clock64() : 0;

        // ---- stage the pass: per-slot state + the A operands of its queries, while the first chunk is in flight ----------
        {
            const float *xp = (const float *)P.xp4;
#pragma unroll
            for (int g = 0; g < NGMAX; g++) {
                if (g < ng) {
                    const int qsl = __shfl(myq, 4 * g + qi);
                    const float *qsrc = xp + (int64_t)max(qsl, 0) * (NKK * 4) + b4;
                    float qv[NB];
#pragma unroll
                    for (int c = 0; c < NB; c++) qv[c] = qsl >= 0 ? qsrc[16 * c] : 0.0f;
#pragma unroll
                    for (int c = 0; c < NB; c++) sQ[(g * NB + c) * 64 + lane] = qv[c];
                }
            }
            if (lane < QB) {
                const int qs = max(myq, 0);
                uint32_t t0 = 0xFFFFFFFFu;
                if (P.gtau && myq >= 0) t0 = ~__hip_atomic_load(&P.gtau[qs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_q[lane] = myq;
                s_pair[lane] = mypair;
                s_tau[lane] = t0;
                s_cnt[lane] = 0;
                s_xn[lane] = (l2 && myq >= 0) ? P.xn[qs] : 0.0f;
            }
        }

        if (P.wave_clock) dbg_t_stage += wall_clock64() - dbg_s0;

        // ---- fused top-k of one group's 64 rows x 4 queries: lane = row, acc[i] = query i of the group ----------------------
        // (xn4 / tau4: the group's four |x|^2 and bounds, requested from LDS before the group's chains start -- with one wave
        //  per SIMD a read issued here would be waited for in full, twice per loop step)
        auto epilogue = [&](int g, const f32x4 acc, int ch, const float yn, const int64_t idr, const float4 xn4, const uint4 tau4) {
            const bool rowvalid = 64 * ch + lane < size_p;
            const float xnv[4] = {xn4.x, xn4.y, xn4.z, xn4.w};
            const uint32_t tauv[4] = {tau4.x, tau4.y, tau4.z, tau4.w};
            uint32_t ordv[4];
            bool anyp = false;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t o = l2 ? ord_from_l2(l2_expanded(xnv[i], yn, acc[i])) : ord_from_ip(acc[i]);
                // (slots beyond the pass's queries and rows beyond the partition never pass)
                ordv[i] = (rowvalid && 4 * g + i < nq) ? o : 0xFFFFFFFFu;
                anyp |= ordv[i] <= tauv[i] && ordv[i] != 0xFFFFFFFFu;
            }
            // steady state: nothing beats the running k-th best -> one ballot, one branch per group and chunk
            if (__ballot(anyp)) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int s = 4 * g + i;
                    const uint32_t ord = ordv[i];
                    uint32_t tau = tauv[i];
                    const bool pass = ord != 0xFFFFFFFFu && ord <= tau;
                    const uint64_t m = __ballot(pass);
                    if (m) {
                        int cnt = s_cnt[s];  // wave-uniform
                        uint32_t *my_ord = pool_ord + s * C;
                        int64_t *my_id = pool_id + s * C;
                        dbg_app += __popcll(m);
                        // the pool has room for APP more entries whenever it is at most k long: append by half / quarter waves
                        for (int h = 0; h < 64; h += APP) {
                            const uint64_t mh = m & ((APP == 32 ? 0xFFFFFFFFull : 0xFFFFull) << h);
                            if (!mh) continue;
                            if (pass && (lane & -APP) == h) {
                                const int slot = cnt + __popcll(mh & ((1ull << lane) - 1ull));
                                my_ord[slot] = ord;
                                my_id[slot] = idr;
                            }
                            cnt += __popcll(mh);
                            if (cnt > C - APP) {
                                dbg_comp++;
                                uint32_t kth;
                                cnt = select_pool<1>(my_ord, my_id, cnt, k, lane, kth);
                                if (cnt >= k) {
                                    tau = min(tau, kth);
                                    if (P.gtau && P.tau_publish && lane == 0) atomicMax(&P.gtau[s_q[s]], ~tau);
                                }
                            }
                        }
                        if (lane == 0) {
                            s_cnt[s] = cnt;
                            s_tau[s] = tau;
                        }
                    }
                }
            }
        };

// One loop step = two groups' chains interleaved over the chunk: 2 x 16 NB instructions, no LDS access inside; the A
// registers of the NEXT pair are requested before the chains of this one.
#define RL_MF1(A, c_, t_, g_)                                                                                             \
    acc0_ = __builtin_amdgcn_mfma_f32_4x4x1f32(qa_[c_], f4c(A[(c_) * 4 + (g_)], t_), acc0_, 4, 4 * (t_) + (g_), 0);        \
    acc1_ = __builtin_amdgcn_mfma_f32_4x4x1f32(qb_[c_], f4c(A[(c_) * 4 + (g_)], t_), acc1_, 4, 4 * (t_) + (g_), 0);
#define RL_MF4(A, c_, t_) RL_MF1(A, c_, t_, 0) RL_MF1(A, c_, t_, 1) RL_MF1(A, c_, t_, 2) RL_MF1(A, c_, t_, 3)
#define RL_MF16(A, c_) RL_MF4(A, c_, 0) RL_MF4(A, c_, 1) RL_MF4(A, c_, 2) RL_MF4(A, c_, 3)
#define RL_STEP(A, Y, I, CH)                                                                                          \
    {                                                                                                                 \
        float qa_[NB], qb_[NB], na_[NB], nb_[NB];                                                                     \
        {                                                                                                             \
            const float *p0_ = sQ + lane, *p1_ = sQ + (size_t)(ng > 1 ? 1 : 0) * NB * 64 + lane;                      \
            _Pragma("unroll") for (int c_ = 0; c_ < NB; c_++) {                                                       \
                na_[c_] = p0_[c_ * 64];                                                                               \
                nb_[c_] = p1_[c_ * 64];                                                                               \
            }                                                                                                         \
        }                                                                                                             \
        for (int g0_ = 0; g0_ < ng; g0_ += 2) {                                                                       \
            dbg_gc++;                                                                                                 \
            f32x4 acc0_ = {0.f, 0.f, 0.f, 0.f}, acc1_ = {0.f, 0.f, 0.f, 0.f};                                         \
            _Pragma("unroll") for (int c_ = 0; c_ < NB; c_++) {                                                       \
                qa_[c_] = na_[c_];                                                                                    \
                qb_[c_] = nb_[c_];                                                                                    \
            }                                                                                                         \
            {                                                                                                         \
                const int gn0_ = min(g0_ + 2, ng - 1), gn1_ = min(g0_ + 3, ng - 1);                                   \
                const float *p0_ = sQ + (size_t)gn0_ * NB * 64 + lane, *p1_ = sQ + (size_t)gn1_ * NB * 64 + lane;     \
                _Pragma("unroll") for (int c_ = 0; c_ < NB; c_++) {                                                   \
                    na_[c_] = p0_[c_ * 64];                                                                           \
                    nb_[c_] = p1_[c_ * 64];                                                                           \
                }                                                                                                     \
            }                                                                                                         \
            const int g1_ = min(g0_ + 1, ng - 1);                                                                     \
            const float4 xa_ = *(const float4 *)(s_xn + 4 * g0_), xb_ = *(const float4 *)(s_xn + 4 * g1_);            \
            const uint4 ta_ = *(const uint4 *)(s_tau + 4 * g0_), tb_ = *(const uint4 *)(s_tau + 4 * g1_);             \
            __builtin_amdgcn_sched_barrier(0); /* (keep the reads up here: the scheduler sinks them to their use otherwise) */ \
            if (!(P.rl_probe & 2)) {                                                                                  \
                _Pragma("unroll") for (int c_ = 0; c_ < NB; c_++) { RL_MF16(A, c_) }                                  \
            } else {                                                                                                  \
                acc0_[0] += A[0].x + A[NB * 4 - 1].w + qa_[0];                                                        \
                acc1_[1] += A[NB * 2].y + qb_[NB - 1];                                                                \
            }                                                                                                         \
            if (!(P.rl_probe & 1)) {                                                                                  \
                epilogue(g0_, acc0_, CH, Y, I, xa_, ta_);                                                             \
                if (g0_ + 1 < ng) epilogue(g0_ + 1, acc1_, CH, Y, I, xb_, tb_);                                       \
            } else if (acc0_[0] + acc1_[1] + Y + (float)I == 12345.678f) {                                            \
                s_cnt[0] = 1;                                                                                         \
            }                                                                                                         \
        }                                                                                                             \
    }

        // Every load below is unconditional inside its block (counted s_waitcnt vmcnt(N): the next chunk stays in flight under
        // the MFMA chains of the current one) and no chunk is requested twice: the steady loop needs two more chunks after the
        // one it computes; the last one or two chunks are peeled.
        // (every load of the steady loop is UNCONDITIONAL: a load inside a branch makes hipcc fall back to s_waitcnt vmcnt(0),
        //  which drains the prefetched chunk.  Bounds published by other waves are picked up at every pass start -- the staging
        //  above -- not inside a pass: an unconditional agent-scope re-read per step from every wave was measured 2x slower.)
        int ch = ch0;
        while (ch + 2 < ch1) {
            RL_LOAD(a1, y1, i1);
            RL_STEP(a0, y0, i0, ch);
            RL_LOAD(a0, y0, i0);
            RL_STEP(a1, y1, i1, ch + 1);
            ch += 2;
        }
        if (ch + 1 < ch1) {
            RL_LOAD(a1, y1, i1);
            RL_STEP(a0, y0, i0, ch);
            RL_STEP(a1, y1, i1, ch + 1);
        } else {
            RL_STEP(a0, y0, i0, ch);
        }
#undef RL_LOAD
#undef RL_STEP
#undef RL_MF16
#undef RL_MF4
#undef RL_MF1

        // ---- segment end: sort every non-empty pool, publish its bound, emit it as a record of its pair -------------------
        const long long dbg_e0 = P.wave_clock ? wall_clock64() : 0;
        {
            int cntl = lane < QB ? s_cnt[lane] : 0;
            uint64_t need = __ballot(cntl > 0);
            while (need) {
                const int sq = __ffsll((unsigned long long)need) - 1;
                need &= need - 1;
                const int n = __builtin_amdgcn_readlane(cntl, sq);
                const int nn = compact_pool<1>(pool_ord + sq * C, pool_id + sq * C, n, k, lane);
                if (lane == sq) cntl = nn;
            }
            const uint64_t have = __ballot(cntl > 0);
            if (have) {
                // slot in the pair's line and record numbers of the segment: independent atomics, one round trip
                int slot = -1, base_rec = 0;
                if (cntl > 0) slot = atomicAdd(&P.pair_slots[(int64_t)mypair * QK_SLOTS], 1);
                if (lane == 0) base_rec = atomicAdd(P.rec_counter, nq);
                const int rec0 = __builtin_amdgcn_readfirstlane(base_rec);
                int myrec = -1;
                if (cntl > 0) {
                    myrec = rec0 + lane;
                    if (slot < QK_SLOTS - 1) P.pair_slots[(int64_t)mypair * QK_SLOTS + 1 + slot] = myrec < P.max_recs ? myrec : -1;
                    if (myrec >= P.max_recs) *P.overflow = 1;
                    if (myrec < P.max_recs) {
                        int old = -1;
                        if (slot >= QK_SLOTS - 1) old = atomicExch(&P.pair_head[mypair], myrec);  // beyond the line: chained
                        P.rec_hdr[myrec] = make_int2(old, cntl);
                        if (P.gtau && P.tau_publish && cntl >= k) atomicMax(&P.gtau[myq], ~pool_ord[lane * C + k - 1]);
                    }
                }
                uint64_t todo = have;
                while (todo) {
                    const int sq = __ffsll((unsigned long long)todo) - 1;
                    todo &= todo - 1;
                    const int n = __builtin_amdgcn_readlane(cntl, sq);
                    const int rec = __builtin_amdgcn_readlane(myrec, sq);
                    if (rec < P.max_recs)
                        for (int e = lane; e < n; e += 64) {
                            P.rec_ord[(int64_t)rec * k + e] = pool_ord[sq * C + e];
                            P.rec_id[(int64_t)rec * k + e] = pool_id[sq * C + e];
                        }
                }
            }
        }
        if (P.wave_clock) dbg_t_end += wall_clock64() - dbg_e0;
    }
    }  // range
        if (!dyn) break;
        if (xcd_ticks == 0 && P.xcd_stat) xcd_ticks = wall_clock64() - wc0;
        const long long chunk = max((long long)P.dyn_chunk, (T - Ts + QK_RL_DYN_MAX - 1) / QK_RL_DYN_MAX);
        unsigned long long c = 0;
        if (lane == 0) c = atomicAdd(P.dyn_counter, (unsigned long long)chunk);
        c = __shfl(c, 0);
        T0 = Ts + (long long)c;
        if (T0 >= T) break;
        T1 = min(T, T0 + chunk);
    }
    if (P.xcd_stat && lane == 0) {
        atomicAdd(&P.xcd_stat[blockIdx.x & 7], (unsigned long long)(xcd_ticks ? xcd_ticks : wall_clock64() - wc0));
        atomicAdd(&P.xcd_stat[8 + (blockIdx.x & 7)], 1ull);
    }
    if (P.wave_clock && lane == 0) {
        long long *wcp = P.wave_clock + 8 * vblock;
        wcp[0] = wc0;
        wcp[1] = wall_clock64();
        wcp[2] = dbg_comp;
        wcp[3] = dbg_app;
        wcp[4] = dbg_seg;
        wcp[5] = clock64() - cy0;  // shader cycles of the wave (with [1] - [0] in 100 MHz ticks: the effective clock)
        wcp[6] = dbg_gc;  // (group-pair x chunk) steps instead of the staging ticks k_scan reports here
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
        wcp[7] = ((long long)xcc << 32) | hw;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
size_t qk_scan_rl_lds_per_wave(int nblk, int C, int qb) {
    return (size_t)(qb / 4) * nblk * 64 * 4 + (size_t)qb * C * 12 + (size_t)qb * 4 * 5;
}

template <int NB, bool L2>
static int launch_rl_m(dim3 grid, size_t lds, hipStream_t st, const ScanParams &sp) {
    QK_HIP(hipFuncSetAttribute((const void *)k_scan_rl<NB, L2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_scan_rl<NB, L2>), grid, dim3(64 * sp.pack), lds, st, sp);
    return QK_OK;
}
template <int NB>
static int launch_rl_t(dim3 grid, size_t lds, hipStream_t st, const ScanParams &sp) {
    return sp.metric == QK_METRIC_L2 ? launch_rl_m<NB, true>(grid, lds, st, sp) : launch_rl_m<NB, false>(grid, lds, st, sp);
}

// grid = hardware workgroups of sp.pack independent waves; lds = sp.pack x sp.pack_lds
int qk_launch_scan_rl(int nblk, dim3 grid, size_t lds, hipStream_t st, const ScanParams &sp) {
    switch (nblk) {
        case 1: return launch_rl_t<1>(grid, lds, st, sp);
        case 2: return launch_rl_t<2>(grid, lds, st, sp);
        case 3: return launch_rl_t<3>(grid, lds, st, sp);
        case 4: return launch_rl_t<4>(grid, lds, st, sp);
        case 5: return launch_rl_t<5>(grid, lds, st, sp);
        case 6: return launch_rl_t<6>(grid, lds, st, sp);
        case 7: return launch_rl_t<7>(grid, lds, st, sp);
        case 8: return launch_rl_t<8>(grid, lds, st, sp);
    }
    QK_FAIL(QK_ERR_UNSUPPORTED, "row-per-lane scan: d > 128 (nblk=%d)", nblk);
}

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
//
// Round 3: the same kernel carries the MIXED work sequence (template parameter HOT): lists shared by many queries of the batch
// leave the per-wave sequence and become dense workgroup items behind a bf16 prefilter -- see "hot lists" below.
#include "qk_internal.h"
#include "qk_device.h"
#include "qk_scan_types.h"

#include <algorithm>

__device__ __forceinline__ float4 rl_ld_nt(const float4 *p) {
    const f32x4 t = __builtin_nontemporal_load((const f32x4 *)p);
    return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ float f4c(const float4 &v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }


// ---- hot lists: dense items on v_mfma_f32_16x16x4_f32, one workgroup per item (HOT form) --------------------------------------
// With realistic nprobe a few lists are probed by 33 ... 600 queries of a 1024-query batch: on the bench mixture they hold 9 / 21 /
// 32 % of the probed bytes and 57 / 78 / 89 % of the multiply-adds at nprobe 8 / 16 / 32.  The per-wave walk above serves them in
// passes of 32 queries (one more read of the list per pass) on the 4x4x1 instruction, whose two-pass issue leaves no room to hide
// the top-k epilogue.  Here such a list is a GEMM: an ITEM = (list, block of <= hq queries, row range).  The block's queries sit
// in LDS ONCE for the workgroup, in B-operand order (hq / 16 query tiles, 1 KB per tile and 16 columns); the four waves take
// different row-tile pairs of the range (A operands straight from the tile-major arena: 2 x NB float4 per lane, register
// double-buffered) and multiply each pair with every query tile: two interleaved accumulator chains of 4 NB instructions -- the
// same k-ordered chain as k_scan, so the same bits.  16x16x4 is an 8-pass instruction: the epilogue of the previous query tile
// (8 keys per lane against the tile's bounds, one ballot) issues in the shadow of the next tile's chains.
// Top-k: ONE pool per query of the block, shared by the four waves (an append takes the query's LDS spin lock; appends are rare
// once the bound has settled, and a wave never holds two locks or waits at a barrier inside one).  At the end of the item every
// non-empty pool leaves as a record of its (query, list) pair, exactly like a segment of the per-wave walk.
//
// PREFILTER.  Every (row tile pair, query tile) product is first computed on v_mfma_f32_16x16x32_bf16 -- the same fragments
// rounded to bf16, 16x the fp32 rate -- and only when some lane's APPROXIMATE key could beat its query's bound does the pair
// run the exact fp32 chains.  The test is one-sided and rigorous: with x~ = bf16(x), y~ = bf16(y) (round to nearest even -- checked
// on the hardware, scripts/micro/cvt_bf16.hip: 2^24 values, all equal to the reference -- 8 significant bits: relative error
// <= 2^-8 each) |x~.y~ - x.y| <= (2^-7 + 2^-16) sum|x_i y_i| <= (2^-7 + 2^-16) (|x|^2 + |y|^2) / 2, the fp32
// accumulation of either path adds less than 2^-16 of that, so with c = 2^-7 * 129/128 + 2^-21
//     L2:  d2_exact >= (|x|^2 + |y|^2)(1 - c) - 2 x~.y~        IP:  x.y <= x~.y~ + c (|x|^2 + |y|^2) / 2
// and a candidate whose bound already loses against the running k-th key cannot enter the top-k.  Everything that IS appended
// went through the exact chain, so ids and distance bits are those of the unfiltered scan; the k-order of the bf16
// instruction is free, so its operands are the fp32 fragments converted in place (no second layout).
// (fp16 operands -- a band of 2^-10 instead of 2^-8, with range guards for |x|^2 > 2^30 and an absolute term for its subnormals --
//  were built and measured (against a bf16 band that was still taken as 2^-8): parity green, fewer row tiles recomputed, and 2-9 % SLOWER at nprobe 8-64: the conversions cost two
//  to three VALU instructions per value where v_cvt_pk_bf16_f32 converts a pair in one, and the test is VALU-bound.)
#ifdef QK_PF_C_PROBE                     // probe builds only: what a tighter (unproven) constant would be worth
constexpr float QK_PF_C = QK_PF_C_PROBE;
constexpr float QK_PF_K1 = 1.0f - QK_PF_C_PROBE;
#else
// (round 4: + 2^-15 + 2^-20.  The accumulator of the bf16 product now STARTS at the row's part of the bound -- -|y|^2 (1 - c) / 2 --
//  so that the test is one compare of the instruction's result with a per-query threshold: its roundings are then relative to
//  |y|^2 / 2 + sum|x_i y_i| <= |x|^2 + |y|^2, 2^-16 of that on a quantity that enters the key's scale twice; the threshold's own
//  rounding is 2^-24 of |x|^2 + tau.)
constexpr float QK_PF_C = 0.0079071f;   // >= 2^-7 * 129/128 + 2^-21 + 2^-15 + 2^-20
constexpr float QK_PF_K1 = 0.992092f;   // <= 1 - QK_PF_C
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct HotLds {
    float4 *sB;          // [hq/16][NB][64] query tiles, B-operand lane order
    uint4 *sBh;          // [hq/16][(NB+1)/2][64] the same tiles in bf16: 8 values per lane and 32-column step
    int64_t *pool_id;    // [hq][C]
    uint32_t *pool_ord;  // [hq][C]
    int *q, *pair;       // [hq] query / pair of every slot (-1: dead slot)
    uint32_t *tau;       // [hq] running bound (key)
    int *cnt, *lock;     // [hq] pool fill, spin lock
    float *xn;           // [hq] |x|^2
    float *theta;        // [hq] the prefilter's threshold of the slot (hot_theta): follows tau
    int *item;           // [4] claimed item (broadcast)
};
__device__ __forceinline__ HotLds hot_lds(unsigned char *smem, int hq, int nb, int C) {
    HotLds h;
    h.sB = (float4 *)smem;
    h.sBh = (uint4 *)(smem + (size_t)hq * nb * 64);
    h.pool_id = (int64_t *)(smem + (size_t)hq * nb * 64 + (size_t)hq * ((nb + 1) / 2) * 64);
    h.pool_ord = (uint32_t *)((unsigned char *)h.pool_id + (size_t)hq * C * 8);
    h.q = (int *)(h.pool_ord + (size_t)hq * C);
    h.pair = h.q + hq;
    h.tau = (uint32_t *)(h.pair + hq);
    h.cnt = (int *)(h.tau + hq);
    h.lock = h.cnt + hq;
    h.xn = (float *)(h.lock + hq);
    h.theta = h.xn + hq;
    h.item = (int *)(h.theta + hq);
    return h;
}
size_t qk_scan_hot_lds(int nblk, int C, int hq) {
    return (size_t)hq * ((size_t)nblk * 64 + (size_t)((nblk + 1) / 2) * 64 + (size_t)C * 12 + 28) + 64;
}

// The prefilter's test of an approximate product against a query's bound, as ONE compare.  With a~ the bf16 product:
//   L2  a row cannot enter the top-k if (|x|^2 + |y|^2)(1 - c) - 2 a~ > tau  <=>  a~ - |y|^2 (1 - c) / 2 < (|x|^2 (1 - c) - tau) / 2
//   IP  ... if a~ + c (|x|^2 + |y|^2) / 2 < tau                               <=>  a~ + c |y|^2 / 2       < tau - c |x|^2 / 2
// The left side comes out of the MFMA itself (its accumulator starts at the row's term), the right side is hot_theta: per query
// slot, recomputed only when the slot's bound moves.  "No bound yet" (all ones) gives -inf: everything passes.
template <bool L2>
__device__ __forceinline__ float hot_theta(uint32_t tau, float xn) {
    if (L2) {
        const float tf = tau == 0xFFFFFFFFu ? __builtin_inff() : __uint_as_float(tau);
        return 0.5f * (xn * QK_PF_K1 - tf);
    }
    const float tf = tau == 0xFFFFFFFFu ? -__builtin_inff() : ip_from_ord(tau);
    return tf - xn * (0.5f * QK_PF_C);
}

// one item; every thread of the workgroup calls it with the same arguments.  Barriers: after staging, before and after the
// record emission.
template <int NB, bool L2>
__device__ __forceinline__ void rl_hot_item(const ScanParams &P, const HotLds &H, const ActiveInfo &inf, const int q0, const int nq,
                                            const int t_lo, const int t_hi, int &dbg_app, int &dbg_comp, long long *ht) {
    long long dbg_prod = 0, dbg_exact = 0;  // probe: products tested, row tiles recomputed exactly
    const long long ht0 = P.wave_clock ? wall_clock64() : 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int C = P.hot.C, k = P.k;
    constexpr int NM = (NB + 1) / 2;  // bf16 instructions per product (32 columns each)
    const int QT = (nq + 15) >> 4;
    const int size_p = inf.size;
    const int64_t tile_p0 = inf.row_off >> 4;
    const int npairs_t = (t_hi - t_lo + 1) >> 1;  // row-tile pairs of the range; wave w takes pairs w, w + 4, ...

    float4 a0[2 * NB], a1[2 * NB];  // [tile of the pair][block]
    float4 y0[2], y1[2];
    longlong2 i0[4], i1[4];
    int lp = wv;  // next pair to load
    // EVERY load of the pair loop is unconditional (a wave that has run out of pairs requests its last one again): a load inside
    // a runtime branch makes hipcc wait for vmcnt(0) -- or nearly: the ISA of round 3 held `s_waitcnt vmcnt(2)` right behind the 22
    // loads of the NEXT pair -- before the fragments of the CURRENT pair are touched, so every pair paid a whole HBM round trip
    // with nothing under it (one wave per SIMD); with counted waits the next pair stays in flight under this pair's products
#define HOT_LOAD(A, Y, I)                                                                                   \
    {                                                                                                       \
        const int pc_ = min(lp, max(npairs_t - 1 - ((npairs_t - 1 - wv) & 3), 0));                          \
        _Pragma("unroll") for (int u_ = 0; u_ < 2; u_++) {                                                  \
            const int64_t ta_ = tile_p0 + min(t_lo + 2 * pc_ + u_, t_hi - 1);                               \
            const float4 *src_ = P.vecs + ta_ * (NB * 64) + lane;                                           \
            _Pragma("unroll") for (int c_ = 0; c_ < NB; c_++) A[u_ * NB + c_] = rl_ld_nt(src_ + c_ * 64);   \
            Y[u_] = ((const float4 *)(P.norms + (ta_ << 4)))[g];                                            \
            I[2 * u_] = ((const longlong2 *)(P.ids + (ta_ << 4)))[2 * g];                                   \
            I[2 * u_ + 1] = ((const longlong2 *)(P.ids + (ta_ << 4)))[2 * g + 1];                           \
        }                                                                                                   \
        lp += 4;                                                                                            \
    }
    HOT_LOAD(a0, y0, i0);

    // ---- stage the block: query tiles in B-operand order + per-slot state, while the first pair is in flight ---------------
    // Two memory round trips for the whole block: every thread first requests the query numbers it needs (one per query tile:
    // thread (wave w, g, j) copies blocks c = w and w + 4 of slot 16 qt + j for every tile qt), then all its query pieces and
    // the state of the slot it initialises (thread t < 16 QT: slot t) -- a loop of dependent loads was 13 us of a 60 us item
    {
        const int qbase = inf.qoff + q0;
        int qq[8];
#pragma unroll
        for (int qt = 0; qt < 8; qt++) {
            const int sl = 16 * qt + j;
            qq[qt] = (qt < QT && sl < nq) ? P.grouped_q[qbase + sl] : -1;
        }
        const int sl_own = threadIdx.x;
        const bool own = sl_own < QT * 16;
        const bool live_own = sl_own < nq;
        const int q_own = live_own ? P.grouped_q[qbase + sl_own] : -1;
        const int pair_own = live_own ? P.grouped_pair[qbase + sl_own] : -1;
        float4 v[8][2];
#pragma unroll
        for (int qt = 0; qt < 8; qt++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int c = wv + 4 * u;
                v[qt][u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < NB && qq[qt] >= 0) v[qt][u] = P.xq4[((int64_t)qq[qt] * NB + c) * 4 + g];
            }
        uint32_t t_own = 0xFFFFFFFFu;
        float xn_own = 0.0f;
        if (live_own) {
            if (P.gtau) t_own = ~__hip_atomic_load(&P.gtau[q_own], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xn_own = P.xn[q_own];  // (IP too: the prefilter's error bound scales with |x|^2 + |y|^2 under either metric)
        }
#pragma unroll
        for (int qt = 0; qt < 8; qt++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int c = wv + 4 * u;
                if (c < NB && qt < QT) H.sB[((size_t)qt * NB + c) * 64 + lane] = v[qt][u];
                if (c < NB + (NB & 1) && qt < QT) {  // bf16 copy: block c is half (c & 1) of the lane's 16 bytes of step c / 2
                    const float4 f = v[qt][u];   // (zero for the padding block of an odd NB)
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 hv = {(__bf16)f.x, (__bf16)f.y, (__bf16)f.z, (__bf16)f.w};
                    uint2 *dst = (uint2 *)(H.sBh + ((size_t)qt * NM + (c >> 1)) * 64 + lane) + (c & 1);
                    *dst = __builtin_bit_cast(uint2, hv);
                }
            }
        if (own) {
            H.q[sl_own] = q_own;
            H.pair[sl_own] = pair_own;
            H.tau[sl_own] = t_own;
            H.cnt[sl_own] = 0;
            H.lock[sl_own] = 0;
            H.xn[sl_own] = xn_own;
            H.theta[sl_own] = live_own ? hot_theta<L2>(t_own, xn_own) : __builtin_inff();  // (a dead slot passes nothing)
        }
    }
    __syncthreads();
    const long long ht1 = P.wave_clock ? wall_clock64() : 0;

    // ---- epilogue of one query tile: keys of this lane's 2 x 4 rows for query j of the tile ----------------------------------
    auto keys = [&](const int qt, const f32x4 ac0, const f32x4 ac1, const int pi, const float4 *Y, uint32_t *ordv, const float xnj,
                    const uint32_t tauj, const int umask) -> bool {
        const bool livej = 16 * qt + j < nq;
        bool anyp = false;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int tl = t_lo + 2 * pi + u;
            const float yv[4] = {Y[u].x, Y[u].y, Y[u].z, Y[u].w};
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const float v = u == 0 ? ac0[reg] : ac1[reg];
                const uint32_t o = L2 ? ord_from_l2(l2_expanded(xnj, yv[reg], v)) : ord_from_ip(v);
                const bool valid = livej & (tl < t_hi) & (16 * tl + 4 * g + reg < size_p) & ((umask >> u) & 1);
                ordv[4 * u + reg] = valid ? o : 0xFFFFFFFFu;
                anyp |= valid & (o <= tauj);
            }
        }
        return anyp;
    };
    // slow path: the lanes of column jq append under the query's lock
    auto append = [&](const int qt, const uint32_t *ordv, const longlong2 *I, const uint64_t pm) {
        uint32_t cols = (uint32_t)((pm | (pm >> 16) | (pm >> 32) | (pm >> 48)) & 0xFFFFull);
        while (cols) {
            const int jq = __ffs(cols) - 1;
            cols &= cols - 1;
            const int sl = 16 * qt + jq;
            if (lane == 0) {
                while (__hip_atomic_exchange(&H.lock[sl], 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t tau = __hip_atomic_load(&H.tau[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int cnt = __hip_atomic_load(&H.cnt[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            tau = __builtin_amdgcn_readfirstlane(tau);
            cnt = __builtin_amdgcn_readfirstlane(cnt);
            uint32_t *my_ord = H.pool_ord + (size_t)sl * C;
            int64_t *my_id = H.pool_id + (size_t)sl * C;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const bool pass = (j == jq) && ordv[e] != 0xFFFFFFFFu && ordv[e] <= tau;
                const uint64_t m = __ballot(pass);
                if (m) {
                    if (pass) {
                        const int slot = cnt + __popcll(m & ((1ull << lane) - 1ull));
                        const longlong2 iv = I[e >> 1];
                        my_ord[slot] = ordv[e];
                        my_id[slot] = (e & 1) ? iv.y : iv.x;
                    }
                    cnt += __popcll(m);
                    dbg_app += __popcll(m);
                    if (cnt > C - 4) {
                        dbg_comp++;
                        uint32_t kth;
                        cnt = select_pool<1>(my_ord, my_id, cnt, k, lane, kth);
                        // (the tightened bound is published at the item's end, with its record: a global atomic HERE puts a memory
                        //  operation inside the pair loop's body, and hipcc then drains the prefetched pair -- s_waitcnt vmcnt(0) --
                        //  before every query-tile loop)
                        if (cnt >= k) tau = min(tau, kth);
                    }
                }
            }
            if (lane == 0) {
                H.cnt[sl] = cnt;
                H.tau[sl] = tau;
                H.theta[sl] = hot_theta<L2>(tau, H.xn[sl]);
                __hip_atomic_store(&H.lock[sl], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

#ifdef QK_PROBES  // (probe build only: rl_probe bit 8 drops the chains of the hot items -- the branch splits the block the keys hide in)
#define HOT_PF_PROBE(x) ((P.rl_probe & 32) ? 3 : (P.rl_probe & 64) ? 0 : (x))  /* 32: every product exact; 64: none */
#define HOT_CHAIN1_PROBE(A, OFF_, B, AC) \
    if (P.rl_probe & 8) {                \
        AC[0] += A[OFF_].x + B[0].x;     \
    } else
#define HOT_CHAIN_PROBE(A, AC0, AC1, B) \
    if (P.rl_probe & 8) {               \
        AC0[0] += A[0].x + B[0].x;      \
        AC1[1] += A[NB].y + B[NB - 1].y; \
    } else
#else
#define HOT_CHAIN_PROBE(A, AC0, AC1, B)
#define HOT_CHAIN1_PROBE(A, OFF_, B, AC)
#define HOT_PF_PROBE(x) (x)
#endif
// the B operands of a query tile: NB float4 per lane, requested a whole tile AHEAD of their chains (with one buffer hipcc put a
// ds_read_b128 and a full s_waitcnt lgkmcnt(0) in front of every 8 instructions of the chain)
#define HOT_BLOAD(B, QT_)                                                                                    \
    {                                                                                                        \
        const float4 *bq_ = H.sB + (size_t)(QT_) * NB * 64 + lane;                                           \
        _Pragma("unroll") for (int c_ = 0; c_ < NB; c_++) B[c_] = bq_[c_ * 64];                              \
    }
#define HOT_CHAIN(A, B, AC0, AC1)                                                                            \
    {                                                                                                        \
        AC0 = (f32x4){0.f, 0.f, 0.f, 0.f};                                                                   \
        AC1 = (f32x4){0.f, 0.f, 0.f, 0.f};                                                                   \
        HOT_CHAIN_PROBE(A, AC0, AC1, B)                                                                      \
        _Pragma("unroll") for (int c_ = 0; c_ < NB; c_++) {                                                  \
            const float4 b_ = B[c_];                                                                         \
            AC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c_].x, b_.x, AC0, 0, 0, 0);                         \
            AC1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[NB + c_].x, b_.x, AC1, 0, 0, 0);                    \
            AC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c_].y, b_.y, AC0, 0, 0, 0);                         \
            AC1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[NB + c_].y, b_.y, AC1, 0, 0, 0);                    \
            AC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c_].z, b_.z, AC0, 0, 0, 0);                         \
            AC1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[NB + c_].z, b_.z, AC1, 0, 0, 0);                    \
            AC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c_].w, b_.w, AC0, 0, 0, 0);                         \
            AC1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[NB + c_].w, b_.w, AC1, 0, 0, 0);                    \
        }                                                                                                    \
    }
// exact product of the pair in A with query tile QT_: fp32 operands from LDS, the canonical chains, keys, appends
#define HOT_EXACT(A, Y, I, PI, QT_, UM_)                                                                     \
    {                                                                                                        \
        float4 bx_[NB];                                                                                      \
        HOT_BLOAD(bx_, QT_);                                                                                 \
        const float xne_ = H.xn[16 * (QT_) + j];                                                             \
        const uint32_t taue_ = H.tau[16 * (QT_) + j];                                                        \
        f32x4 ea0_ = {0.f, 0.f, 0.f, 0.f}, ea1_ = {0.f, 0.f, 0.f, 0.f};                                      \
        if ((UM_) == 3) {                                                                                    \
            HOT_CHAIN(A, bx_, ea0_, ea1_);                                                                   \
        } else if ((UM_) == 1) {                                                                             \
            HOT_CHAIN1(A, 0, bx_, ea0_);                                                                     \
        } else {                                                                                             \
            HOT_CHAIN1(A, NB, bx_, ea1_);                                                                    \
        }                                                                                                    \
        uint32_t ordv_[8];                                                                                   \
        const bool anyp_ = keys(QT_, ea0_, ea1_, PI, Y, ordv_, xne_, taue_, UM_);                            \
        const uint64_t pm_ = __ballot(anyp_);                                                                \
        if (pm_ && !(P.rl_probe & 128)) append(QT_, ordv_, I, pm_);  /* (probe 128: exact chains, nothing appended) */ \
    }
// (one row tile only: a single dependent chain)
#define HOT_CHAIN1(A, OFF_, B, AC)                                                                           \
    {                                                                                                        \
        HOT_CHAIN1_PROBE(A, OFF_, B, AC)                                                                     \
        _Pragma("unroll") for (int c_ = 0; c_ < NB; c_++) {                                                  \
            const float4 b_ = B[c_];                                                                         \
            AC = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(OFF_) + c_].x, b_.x, AC, 0, 0, 0);                  \
            AC = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(OFF_) + c_].y, b_.y, AC, 0, 0, 0);                  \
            AC = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(OFF_) + c_].z, b_.z, AC, 0, 0, 0);                  \
            AC = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(OFF_) + c_].w, b_.w, AC, 0, 0, 0);                  \
        }                                                                                                    \
    }
#define HOT_HLOAD(B, QT_)                                                                                    \
    {                                                                                                        \
        const uint4 *bq_ = H.sBh + (size_t)(QT_) * NM * 64 + lane;                                           \
        _Pragma("unroll") for (int m_ = 0; m_ < NM; m_++) B[m_] = bq_[m_ * 64];                              \
    }
// one query tile: the bf16 product of the pair on the operands in BC (those of the next tile requested into BN first), the
// one-sided test of its 8 approximate keys, and the exact product if any lane of the wave could still be a candidate
#define HOT_TILE(A, Y, I, PI, BC, BN, QT_)                                                                   \
    {                                                                                                        \
        HOT_HLOAD(BN, min((QT_) + 1, QT - 1));                                                               \
        const float thj_ = H.theta[16 * (QT_) + j];                                                          \
        f32x4 d0_ = {ykh_[0], ykh_[1], ykh_[2], ykh_[3]}, d1_ = {ykh_[4], ykh_[5], ykh_[6], ykh_[7]};        \
        _Pragma("unroll") for (int m_ = 0; m_ < NM; m_++) {                                                  \
            const bf16x8 bh_ = __builtin_bit_cast(bf16x8, BC[m_]);                                           \
            d0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_[m_], bh_, d0_, 0, 0, 0);                       \
            d1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_[NM + m_], bh_, d1_, 0, 0, 0);                  \
        }                                                                                                    \
        /* one compare per row tile (hot_theta); negated: a NaN on either side stays a candidate, as in the per-wave walk and */ \
        /* k_pf_gemm; a row beyond the list started at -inf and passes only a query without a bound (the exact path masks it) */ \
        const bool flag_ = !(fmaxf(fmaxf(d0_[0], d0_[1]), fmaxf(d0_[2], d0_[3])) < thj_);                    \
        const bool flag1_ = !(fmaxf(fmaxf(d1_[0], d1_[1]), fmaxf(d1_[2], d1_[3])) < thj_);                   \
        const int um_ = HOT_PF_PROBE((__ballot(flag_) ? 1 : 0) | (__ballot(flag1_) ? 2 : 0));                \
        dbg_prod += 2;                                                                                       \
        if (um_) {                                                                                           \
            dbg_exact += (um_ & 1) + (um_ >> 1);                                                             \
            HOT_EXACT(A, Y, I, PI, QT_, um_);                                                                \
        }                                                                                                    \
    }
#define HOT_PAIR(A, Y, I, PI)                                                                                \
    {                                                                                                        \
        /* bf16 copies of the pair's fragments (the fp32 lane layout, converted in place), the row part of the bound */ \
        bf16x8 ah_[2 * NM];                                                                                  \
        float ykh_[8];  /* where the accumulator of a row starts (hot_theta) */                              \
        _Pragma("unroll") for (int u_ = 0; u_ < 2; u_++) {                                                   \
            _Pragma("unroll") for (int m_ = 0; m_ < NM; m_++) {                                              \
                const float4 f0_ = A[u_ * NB + 2 * m_];                                                      \
                const float4 f1_ = 2 * m_ + 1 < NB ? A[u_ * NB + (2 * m_ + 1 < NB ? 2 * m_ + 1 : 0)] : make_float4(0.f, 0.f, 0.f, 0.f); \
                ah_[u_ * NM + m_] = (bf16x8){(__bf16)f0_.x, (__bf16)f0_.y, (__bf16)f0_.z, (__bf16)f0_.w,     \
                                             (__bf16)f1_.x, (__bf16)f1_.y, (__bf16)f1_.z, (__bf16)f1_.w};    \
            }                                                                                                \
            const int tl_ = t_lo + 2 * (PI) + u_;                                                            \
            const float yv_[4] = {Y[u_].x, Y[u_].y, Y[u_].z, Y[u_].w};                                       \
            _Pragma("unroll") for (int reg_ = 0; reg_ < 4; reg_++) {                                         \
                const bool rv_ = (tl_ < t_hi) & (16 * tl_ + 4 * g + reg_ < size_p);                          \
                ykh_[4 * u_ + reg_] = !rv_ ? -__builtin_inff() : L2 ? -0.5f * (yv_[reg_] * QK_PF_K1) : yv_[reg_] * (0.5f * QK_PF_C); \
            }                                                                                                \
        }                                                                                                    \
        uint4 bc_[NM], bn_[NM];                                                                              \
        HOT_HLOAD(bc_, 0);                                                                                   \
        for (int qt_ = 0; qt_ < QT; qt_ += 2) {                                                              \
            HOT_TILE(A, Y, I, PI, bc_, bn_, qt_);                                                            \
            if (qt_ + 1 < QT) HOT_TILE(A, Y, I, PI, bn_, bc_, qt_ + 1);                                      \
        }                                                                                                    \
    }

    {
        int pi = wv;  // (the pair a0 holds, if it exists)
        while (pi < npairs_t) {
            HOT_LOAD(a1, y1, i1);
            HOT_PAIR(a0, y0, i0, pi);
            pi += 4;
            if (pi >= npairs_t) break;
            HOT_LOAD(a0, y0, i0);
            HOT_PAIR(a1, y1, i1, pi);
            pi += 4;
        }
    }
#undef HOT_LOAD
#undef HOT_CHAIN
#undef HOT_CHAIN_PROBE
#undef HOT_CHAIN1_PROBE
#undef HOT_PF_PROBE
#undef HOT_EXACT
#undef HOT_CHAIN1
#undef HOT_HLOAD
#undef HOT_BLOAD
#undef HOT_TILE
#undef HOT_PAIR

    // ---- item end: every non-empty pool becomes a record of its pair (wave w takes slots [w * hq / 4, (w + 1) * hq / 4)) -----
    const long long ht2 = P.wave_clock ? wall_clock64() : 0;
    __syncthreads();
    const long long ht3 = P.wave_clock ? wall_clock64() : 0;
    {
        const int spw = P.hot.hq >> 2;  // slots per wave (multiple of 4, <= 32)
        const int sl = wv * spw + lane;
        const bool own = lane < spw && sl < QT * 16;
        int cntl = own ? H.cnt[sl] : 0;
        uint64_t need = __ballot(cntl > 0);
        while (need) {
            const int sq = __ffsll((unsigned long long)need) - 1;
            need &= need - 1;
            const int n = __builtin_amdgcn_readlane(cntl, sq);
            const int nn = compact_pool<1>(H.pool_ord + (size_t)(wv * spw + sq) * C, H.pool_id + (size_t)(wv * spw + sq) * C, n, k, lane);
            if (lane == sq) cntl = nn;
        }
        const uint64_t have = __ballot(cntl > 0);
        if (have) {
            const int mypair = own ? H.pair[sl] : -1;
            const int myq = own ? H.q[sl] : -1;
            int slot = -1, base_rec = 0;
            if (cntl > 0) slot = atomicAdd(&P.pair_slots[(int64_t)mypair * QK_SLOTS], 1);
            if (lane == 0) base_rec = atomicAdd(P.rec_counter, __popcll(have));
            const int rec0 = __builtin_amdgcn_readfirstlane(base_rec);
            int myrec = -1;
            if (cntl > 0) {
                myrec = rec0 + __popcll(have & ((1ull << lane) - 1ull));
                if (slot < QK_SLOTS - 1) P.pair_slots[(int64_t)mypair * QK_SLOTS + 1 + slot] = myrec < P.max_recs ? myrec : -1;
                if (myrec >= P.max_recs) *P.overflow = 1;
                if (myrec < P.max_recs) {
                    int old = -1;
                    if (slot >= QK_SLOTS - 1) old = atomicExch(&P.pair_head[mypair], myrec);
                    P.rec_hdr[myrec] = make_int2(old, cntl);
                    if (P.gtau && P.tau_publish && cntl >= k) atomicMax(&P.gtau[myq], ~H.pool_ord[(size_t)sl * C + k - 1]);
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
                        P.rec_ord[(int64_t)rec * k + e] = H.pool_ord[(size_t)(wv * spw + sq) * C + e];
                        P.rec_id[(int64_t)rec * k + e] = H.pool_id[(size_t)(wv * spw + sq) * C + e];
                    }
            }
        }
    }
    __syncthreads();
    if (P.wave_clock) {
        const long long ht4 = wall_clock64();
        ht[0] += ht1 - ht0;  // staging (+ first loads)
        ht[1] += ht2 - ht1;  // chains + keys + appends
        ht[2] += ht3 - ht2;  // waiting for the other waves of the item
        ht[3] += ht4 - ht3;  // record emission + closing barrier
        ht[6] += dbg_prod;
        ht[7] += dbg_exact;
    }
}

// NB = 16-column blocks per row (d <= 128).  One hardware workgroup = 4 independent waves (own range, own LDS, no barrier).
// (the metric is a template parameter: as a runtime flag it left uniform branches around every result element of the epilogue)
// HOT: the mixed work sequence -- lists probed by >= P.hot.min queries are items of a second queue (rl_hot_item); workgroups
// below n_hot_first start there and join the per-wave sequence's dynamic tail afterwards, the others walk their static share
// and the tail first and take hot items when the sequence is exhausted: MFMA-bound and HBM-bound work overlap on the chip.
template <int NB, bool L2, bool HOT>
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
    const bool dyn = P.dyn_counter != nullptr;
    const long long Ts = dyn ? T - (T * P.dyn_pct) / 100 : T;
    // hot-first workgroups: their share of the grid = the hot items' share of the launch's cost (multiple of 8: whole rounds
    // over the XCDs); they take no static share of the per-wave sequence
    int n_hot = 0, n_hot_first = 0;
    if (HOT) {
        n_hot = *P.n_hot;
        if (n_hot > 0 && dyn && gridDim.x >= 16) {
            const double hu = (double)*P.hot_units * (double)P.hot_first_pct * 0.01;
            n_hot_first = ((int)((double)gridDim.x * hu / (hu + (double)T + 1.0) + 4.0)) & ~7;
            n_hot_first = max(0, min(n_hot_first, ((int)gridDim.x - 8) & ~7));
        }
    }
    if (HOT && (P.rl_probe & 4)) n_hot_first = 0;  // probe: hot items are dropped, everybody cuts the per-wave sequence
    const bool hot_first = HOT && (int)blockIdx.x < n_hot_first;
    const long long gu = (long long)gridDim.x - n_hot_first;   // workgroups that cut the static share
    const long long u = (long long)blockIdx.x - n_hot_first;   // (n_hot_first is a multiple of 8: u % 8 is still the XCD class)
    const long long W = gu * PK;
    const long long vblock = (long long)blockIdx.x * PK + wvp;
    const long long vb_c = u * PK + wvp;
    long long T0 = 0, T1 = 0;
    if (!hot_first) {
        T0 = (Ts * vb_c) / W;
        T1 = (Ts * (vb_c + 1)) / W;
        if (P.xcd_on) {
            long long pre[9];
            pre[0] = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) pre[i + 1] = pre[i] + P.xcd_w[i];
            const long long total = ((gu >> 3) * pre[8] + pre[gu & 7]) * PK;
            const long long a0 = ((u >> 3) * pre[8] + pre[u & 7]) * PK + (long long)P.xcd_w[u & 7] * wvp;
            const long long a1 = a0 + P.xcd_w[u & 7];
            T0 = a0 <= 0 ? 0 : (long long)((double)Ts * (double)a0 / (double)total);
            T1 = a1 >= total ? Ts : (long long)((double)Ts * (double)a1 / (double)total);
        }
    }
    if (!HOT && !dyn && T1 <= T0) return;
    const long long wc0 = (P.xcd_stat || P.wave_clock) ? wall_clock64() : 0;
    const long long cy0 = P.wave_clock ? clock64() : 0;
    int dbg_comp = 0, dbg_app = 0, dbg_seg = 0;  // probe counters (QK_SCAN_WAVE_CLOCK)
    long long dbg_t_end = 0, dbg_t_stage = 0, dbg_gc = 0;
    const RlCost rc{P.rl_h0, P.rl_h1, P.rl_e, P.seg_ovh, P.rl_m, QB};
    const int n_active = *P.n_active;
    long long xcd_ticks = 0;  // time spent on the static share (what the XCD balance learns from)
    long long hot_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // probe: ticks in the phases of the hot items, whole items, items
    for (int ph = 0; ph < (HOT ? 2 : 1); ph++) {
    if (HOT && ((ph == 0) == hot_first)) {
        // ---- hot items: the whole workgroup, one item at a time from the queue --------------------------------------------
        if (n_hot <= 0) continue;
        if (!hot_first) __syncthreads();  // every wave has left the per-wave sequence (and its slice of the LDS)
        const HotLds H = hot_lds(smem, P.hot.hq, NB, P.hot.C);
        // (the queue is claimed one item AHEAD: the atomic for the next item goes out when this one starts and has long returned
        //  when it ends -- a device-scope round trip of 1-2 us per item otherwise)
        int next_item = 0;
        if (threadIdx.x == 0) next_item = atomicAdd(P.hot_counter, 1);
        for (;;) {
            const long long hc0 = P.wave_clock ? wall_clock64() : 0;
            if (threadIdx.x == 0) H.item[0] = next_item;
            __syncthreads();
            const int item = H.item[0];
            if (item >= n_hot) {
                __syncthreads();  // (the word sits in a wave's slice of the per-wave form)
                break;
            }
            if (threadIdx.x == 0) next_item = atomicAdd(P.hot_counter, 1);
            // 64-ary search: act_hoff[lo] <= item < act_hoff[lo + 1]
            int lo = 0, hi = n_active;
            while (hi - lo > 1) {
                const int span = hi - lo;
                const int step = (span + 63) >> 6;
                const int probe = min(lo + (lane + 1) * step, hi);
                const bool gt = (probe >= hi) || (P.act_hoff[probe] > item);
                const uint64_t m = __ballot(gt);
                const int first = __ffsll((unsigned long long)m) - 1;
                const int nlo = min(lo + first * step, hi - 1);
                const int nhi = min(lo + (first + 1) * step, hi);
                lo = nlo;
                hi = nhi;
            }
            const ActiveInfo inf = P.active[lo];
            const HotShape hs = hot_shape(inf.cnt, inf.size, P.hot);
            const int local = item - P.act_hoff[lo];
            const int qb = local / hs.nrr, rr = local - qb * hs.nrr;
            const int q0 = qb * hs.qpb, nq = min(hs.qpb, inf.cnt - q0);
            int t_lo, t_hi;
            hot_range(inf.size, rr, hs.nrr, &t_lo, &t_hi);
            if (nq <= 0 || t_hi <= t_lo) {
                __syncthreads();
                continue;
            }
            dbg_seg++;
            hot_t[5]++;
            if (P.rl_probe & 4) {  // probe: hot items claimed and dropped (what the per-wave sequence alone costs)
                __syncthreads();
                continue;
            }
            rl_hot_item<NB, L2>(P, H, inf, q0, nq, t_lo, t_hi, dbg_app, dbg_comp, hot_t);
            if (P.wave_clock) hot_t[4] += wall_clock64() - hc0;
        }
        continue;
    }
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
        if (HOT && hot_list(cnt_p, size_p, P.hot)) {  // a hot list: no units in this sequence (rl_hot_item)
            ai++;
            continue;
        }
        if (HOT && (P.rl_probe & 16)) {  // probe: the per-wave walk does nothing (what the hot items alone cost)
            cur = T1;
            continue;
        }
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
    }  // phase
    if (P.xcd_stat && lane == 0 && !hot_first) {
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
        if (HOT) {  // second half of the probe buffer: the hot items of this wave
            long long *hcp = P.wave_clock + 8 * ((long long)gridDim.x * PK + vblock);
#pragma unroll
            for (int i = 0; i < 6; i++) hcp[i] = hot_t[i];
            hcp[6] = hot_t[6];
            hcp[7] = hot_t[7];
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
size_t qk_scan_rl_lds_per_wave(int nblk, int C, int qb) {
    return (size_t)(qb / 4) * nblk * 64 * 4 + (size_t)qb * C * 12 + (size_t)qb * 4 * 5;
}

template <int NB, bool L2, bool HOT>
static int launch_rl_h(dim3 grid, size_t lds, hipStream_t st, const ScanParams &sp) {
    QK_HIP(hipFuncSetAttribute((const void *)k_scan_rl<NB, L2, HOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_scan_rl<NB, L2, HOT>), grid, dim3(64 * sp.pack), lds, st, sp);
    return QK_OK;
}
template <int NB, bool L2>
static int launch_rl_m(dim3 grid, size_t lds, hipStream_t st, const ScanParams &sp) {
    // (the mixed form needs whole workgroups of four waves and the LDS of all four slices)
    return sp.hot.min > 0 ? launch_rl_h<NB, L2, true>(grid, lds, st, sp) : launch_rl_h<NB, L2, false>(grid, lds, st, sp);
}
template <int NB>
static int launch_rl_t(dim3 grid, size_t lds, hipStream_t st, const ScanParams &sp) {
    return sp.metric == QK_METRIC_L2 ? launch_rl_m<NB, true>(grid, lds, st, sp) : launch_rl_m<NB, false>(grid, lds, st, sp);
}

// grid = hardware workgroups of sp.pack independent waves; lds = sp.pack x sp.pack_lds
int qk_launch_scan_rl(int nblk, dim3 grid, size_t lds, hipStream_t st, const ScanParams &sp) {
    switch (nblk) {
#ifndef QK_RL_DEV  // (development builds instantiate d = 128 only: this file is five minutes of hipcc otherwise)
        case 1: return launch_rl_t<1>(grid, lds, st, sp);
        case 2: return launch_rl_t<2>(grid, lds, st, sp);
        case 3: return launch_rl_t<3>(grid, lds, st, sp);
        case 4: return launch_rl_t<4>(grid, lds, st, sp);
        case 5: return launch_rl_t<5>(grid, lds, st, sp);
        case 6: return launch_rl_t<6>(grid, lds, st, sp);
        case 7: return launch_rl_t<7>(grid, lds, st, sp);
#endif
        case 8: return launch_rl_t<8>(grid, lds, st, sp);
    }
    QK_FAIL(QK_ERR_UNSUPPORTED, "row-per-lane scan: d > 128 (nblk=%d)", nblk);
}

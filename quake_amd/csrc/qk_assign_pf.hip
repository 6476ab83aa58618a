// qk_assign_pf.hip -- nearest centroid of MANY rows (k-means assign, the final assignment of an index build) behind a bf16 prefilter.
//
// Replaces the assign step of kmeans() / kmeans_refine_partitions() (src/cpp/src/clustering.cpp:51-66 -> faiss::Clustering::train /
// IndexFlat::search(k = 1), :149-159 batched_scan_list(k = 1)) for n >= 40960 rows, d <= 128 with d % 8 == 0.
//
// k_assign (qk_kmeans.hip) computes every (row, centroid) key on v_mfma_f32_16x16x4_f32: 0.78 of the fp32 MFMA peak, 9.0 ms for
// 2^20 rows x 4096 centroids -- and only ONE key per row matters.  Which one can be settled at bf16 precision with the one-sided
// bound of the scan's hot items (qk_scan_rl.hip: |x~.y~ - x.y| <= c (|x|^2 + |y|^2) / 2), on v_mfma_f32_16x16x32_bf16 at 16x the rate:
//
//   workgroup = 8 waves; a wave keeps XT x 16 rows of x in REGISTERS as bf16 B-operands (XT = 8: 128 VGPRs) for the whole kernel;
//   the centroids -- converted once per call into A-operand order by k_apf_prep, 1 MB for 4096 x 128 -- stream through LDS in chunks
//   of 16 tiles (64 KB, double-buffered, global_load_lds: no staging registers), shared by the 8 waves.
//   pass 1   h = |y|^2 - 2 x~.y~ (IP: -x~.y~) for every centroid; per row the minimum hmin.  With E = c (|x|^2 + max|y|^2) a centroid
//            whose h exceeds hmin + 2 E cannot hold the smallest exact key (nor tie with it).
//   pass 2   the same products again against W = hmin + 2 E (L2: at least E - |x|^2, the place where exact keys clamp at 0): a
//            (row, centroid) under W is a CANDIDATE -- parked in the wave's LDS list; when the list fills, and at the end, every
//            candidate's EXACT key is computed, one k-ordered fmaf chain per lane (the arithmetic of the MFMA path, DESIGN.md
//            section 3), and folded into best[row] = min (key << 32 | centroid) -- the (key, index) order of k_assign.
// Every key that decides is exact and every centroid that could decide is a candidate, so assignments (and the optional
// distances) are the bits of k_assign.  Non-finite data makes a norm non-finite, W with it, and every centroid a candidate (slow,
// exact).  On the bench mixture a row has 1-3 candidates out of 4096.
#include "qk_internal.h"
#include "qk_device.h"

#include <algorithm>

typedef __bf16 apf_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t apf_u32x4 __attribute__((ext_vector_type(4)));
constexpr int APF_CH = 16;     // centroid tiles per LDS chunk
constexpr int APF_CAP = 512;   // candidates a wave parks before it works them off (one round of the slow path adds up to 64): a flush
                               // inside the passes holds the whole workgroup up at the next barrier, so the list should last to the end
// the bound's constant (qk_scan_rl.hip QK_PF_C = 2^-7 * 129/128 + 2^-21, which allows 2^-16 of sum|x_i y_i| for the instruction's own
// fp32 accumulation) plus 2^-15: here the accumulator STARTS at -|y|^2 / 2, so its roundings are relative to |y|^2 / 2 + sum|x_i y_i|
// <= |x|^2 + |y|^2 (2^-16 of that, on a in the key's scale twice), and the roundings of the bound itself (<= 2^-22 of it)
constexpr float APF_C = 0.0079061f;

// (timing / counting switches of side builds -- scripts/build_variant.sh <name> qk_assign_pf.hip -DAPF_...: APF_PROBE_STATS counts
//  tests, slow paths and candidates; APF_V_NOPASS2 / APF_V_NOFLUSH drop pass 2 / the exact keys (wrong answers, timing only);
//  APF_V_W16 runs 16 waves x 4 row tiles per workgroup.  The product build has none of them.)
#ifdef APF_PROBE_STATS
__device__ unsigned long long apf_stats[4];  // steps (groups of 4 row tiles x centroid tile), slow groups, candidates, flushes
#define APF_STAT(i_, v_) do { if ((threadIdx.x & 63) == 0) atomicAdd(&apf_stats[i_], (unsigned long long)(v_)); } while (0)
#else
#define APF_STAT(i_, v_) do { } while (0)
#endif

struct AssignPfParams {
    const float *x;       // [n][d] row-major
    int64_t n;
    int d;
    const uint4 *cbf;     // [nch] chunks of { [APF_CH][NM][64] centroid tiles, bf16, A-operand lane order (zero beyond m);
                          //                  [APF_CH * 16] floats  where the accumulator of a centroid starts: L2 -|y|^2 / 2, IP 0 --
                          //                  and -inf beyond m (never the maximum) }
    const float *c;       // [m][d] row-major fp32 (the exact chains)
    const float *cnorm;   // [>= m] canonical |y|^2
    const float *ynmax;   // [1] max |y|^2 (bits: NaN > inf > finite)
    int m, nch;
    int64_t *assign;      // [n] centroid ROW of every x row (k-means), or nullptr with packed_out
    float *val;           // [n] exact key of the assignment as a distance / dot product, or nullptr
    // the nearest-list form (qk_dense_device, k = 1: PartitionManager::add's parent search over many rows): ties go to the smaller ID,
    // and the answer leaves as k_dense_argmin's packed word (key << 32 | id; key 0 where none was asked for and none was needed)
    const int64_t *ids;           // [m] ids of the centroid rows (each in [0, 2^32)), or nullptr: the row number orders and is returned
    unsigned long long *packed_out;  // [n] or nullptr
    int want_keys;        // exact keys for every row (val / the packed word's key are read)
};

// centroid tile t in bf16, step s of lane (i, g) = columns 32 s + 8 g .. + 7 of centroid 16 t + i (the k-order of the bf16
// instruction is free: rows of x are cut the same way); the norms as the passes want them; the largest norm
__global__ __launch_bounds__(64) void k_apf_prep(const float *__restrict__ c, const float *__restrict__ cnorm, int m, int d, int NM, int l2,
                                                 uint4 *__restrict__ cbf, unsigned int *__restrict__ ynmax) {
    const int t = blockIdx.x, lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int row = 16 * t + i;
    const int64_t CHB_U4 = (int64_t)APF_CH * NM * 64 + 64;
    uint4 *chunk = cbf + (t / APF_CH) * CHB_U4;
    const int tt = t % APF_CH;
    for (int s = 0; s < NM; s++) {
        const int col = 32 * s + 8 * g;
        float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
        if (row < m && col < d) {  // (d % 8 == 0: a group of 8 columns is inside the row or beyond it)
            const float4 *p = (const float4 *)(c + (int64_t)row * d + col);
            f0 = p[0];
            f1 = p[1];
        }
        const apf_bf16x8 h = {(__bf16)f0.x, (__bf16)f0.y, (__bf16)f0.z, (__bf16)f0.w, (__bf16)f1.x, (__bf16)f1.y, (__bf16)f1.z, (__bf16)f1.w};
        chunk[(tt * NM + s) * 64 + lane] = __builtin_bit_cast(uint4, h);
    }
    if (lane < 16) {
        const int r = 16 * t + lane;
        ((float *)(chunk + APF_CH * NM * 64))[tt * 16 + lane] = r < m ? (l2 ? -0.5f * cnorm[r] : 0.0f) : -__builtin_inff();
        if (r < m) atomicMax(ynmax, __float_as_uint(cnorm[r]));
    }
}

// the exact keys of the `cnt` candidates in cbuf, folded into best[] (one wave; see the header).  FAST = the form at the end of the
// kernel, inlined where the row operands are dead and registers are free: a lane walks two rows of its own (64 lanes = 128 different
// cache lines per load round), so every 128-byte line of both rows is requested FIRST -- one miss latency for the whole row instead
// of one per line; the rest of a line then comes from the L1 / L2.  The other form is the call from inside pass 2 when a wave's list
// fills up (it never does on data with any structure): few registers, because a call saves what it clobbers.
template <bool L2, bool FAST>
__device__ __forceinline__ void apf_exact(const float *__restrict__ x, const float *__restrict__ c, const float *__restrict__ cnorm, int64_t n,
                                          int m, int d, int64_t wrow0, const uint32_t *cbuf, int cnt, unsigned long long *best,
                                          const float *xn_w, const int64_t *__restrict__ ids) {
#ifdef APF_V_NOFLUSH
    return;
#endif
    const int lane = threadIdx.x & 63;
    const int nv = d >> 2, npad = (16 - (d & 15)) & 15;
    for (int base = 0; base < cnt; base += 64) {
        const int e = base + lane;
        bool has = e < cnt;
        const uint32_t pk = has ? cbuf[e] : 0u;
        const int xl = (int)(pk >> 24), ci = (int)(pk & 0xFFFFFFu);
        has = has && ci < m;
        const int64_t row = min(wrow0 + xl, n - 1);
        const float4 *xp = (const float4 *)(x + row * d);
        const float4 *cp = (const float4 *)(c + (int64_t)min(ci, m - 1) * d);
        float acc = 0.0f;
        if (FAST) {
            const int nseg = (nv + 7) >> 3;  // 128-byte segments of a row (<= 4)
            float4 xa[4], ca[4];
#pragma unroll
            for (int sg = 0; sg < 4; sg++) {
                const int v0 = min(sg * 8, nv - 1);
                xa[sg] = xp[v0];
                ca[sg] = cp[v0];
            }
#pragma unroll
            for (int sg = 0; sg < 4; sg++) {
                if (sg < nseg) {
                    float4 xb[7], cb2[7];
#pragma unroll
                    for (int u = 0; u < 7; u++) {
                        const int v = min(sg * 8 + 1 + u, nv - 1);
                        xb[u] = xp[v];
                        cb2[u] = cp[v];
                    }
                    acc = __fmaf_rn(ca[sg].x, xa[sg].x, acc);
                    acc = __fmaf_rn(ca[sg].y, xa[sg].y, acc);
                    acc = __fmaf_rn(ca[sg].z, xa[sg].z, acc);
                    acc = __fmaf_rn(ca[sg].w, xa[sg].w, acc);
#pragma unroll
                    for (int u = 0; u < 7; u++) {
                        if (sg * 8 + 1 + u < nv) {
                            acc = __fmaf_rn(cb2[u].x, xb[u].x, acc);
                            acc = __fmaf_rn(cb2[u].y, xb[u].y, acc);
                            acc = __fmaf_rn(cb2[u].z, xb[u].z, acc);
                            acc = __fmaf_rn(cb2[u].w, xb[u].w, acc);
                        }
                    }
                }
            }
        } else {
#pragma unroll 4
            for (int v = 0; v < nv; v++) {
                const float4 a = cp[v], b = xp[v];
                acc = __fmaf_rn(a.x, b.x, acc);
                acc = __fmaf_rn(a.y, b.y, acc);
                acc = __fmaf_rn(a.z, b.z, acc);
                acc = __fmaf_rn(a.w, b.w, acc);
            }
        }
        for (int p = 0; p < npad; p++) acc = __fmaf_rn(0.0f, 0.0f, acc);  // (the MFMA path runs over the zero columns of the last block)
        const uint32_t o = L2 ? ord_from_l2(l2_expanded(xn_w[xl], cnorm[min(ci, m - 1)], acc)) : ord_from_ip(acc);
        const uint32_t low = ids ? (uint32_t)ids[min(ci, m - 1)] : (uint32_t)ci;  // what breaks a tie of keys (and what is returned)
        if (has) atomicMin(&best[xl], ((unsigned long long)o << 32) | (unsigned long long)low);
    }
}
template <bool L2>
__device__ __noinline__ void apf_flush(const float *__restrict__ x, const float *__restrict__ c, const float *__restrict__ cnorm, int64_t n,
                                       int m, int d, int64_t wrow0, const uint32_t *cbuf, int cnt, unsigned long long *best,
                                       const float *xn_w, const int64_t *__restrict__ ids) {
    apf_exact<L2, false>(x, c, cnorm, n, m, d, wrow0, cbuf, cnt, best, xn_w, ids);
}

// chunk `chunk` of the centroid stream (APF_CH tiles, then their 256 norms) into LDS: one wave-instruction = 64 consecutive uint4
// (1 KB), the destination is the wave's base + 16 bytes per lane.  Everything but the lane's source offset is scalar (addresses
// kept in VGPRs were spilled, and every reload waited for the loads before it).
template <int NM, int APF_WAVES>
__device__ __forceinline__ void apf_stage(const uint4 *__restrict__ cbf, int chunk, uint4 *dst) {
    constexpr int CHB_U4 = APF_CH * NM * 64 + 64, APF_THREADS = 64 * APF_WAVES;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // scalar base + 32-bit lane offset (global_load_lds v_off, s[base]): a 64-bit lane address lives in two VGPRs for the whole
    // kernel, and at 256 registers it was the value the allocator spilled -- reloaded at the head of every chunk, behind a
    // s_waitcnt vmcnt(0) that also waited for the prefetch just issued
    const char *base = (const char *)(cbf + (int64_t)chunk * CHB_U4);
    const uint32_t voff = (uint32_t)tid * 16u;
#pragma unroll
    for (int r = 0; r * APF_THREADS < CHB_U4; r++) {
        if (r * APF_THREADS + wave * 64 < CHB_U4) {
            // (written out: the builtin takes ONE pointer and the compiler folds base + offset into a 64-bit lane address again)
            const char *sb = base + (size_t)r * APF_THREADS * 16;
            const uint32_t lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(dst + r * APF_THREADS + wave * 64);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
                         :
                         : "s"(sb), "v"(voff), "s"(__builtin_amdgcn_readfirstlane(lds))
                         : "memory", "m0");
        }
    }
}

template <int NM, int XT, int APF_WAVES, bool L2>
__global__ __launch_bounds__(64 * APF_WAVES) void k_assign_pf(AssignPfParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int TILE_U4 = NM * 64, CH_U4 = APF_CH * TILE_U4, CHB_U4 = CH_U4 + 64, WR = XT * 16;
    uint4 *buf = (uint4 *)smem;                                                        // [2][CHB_U4]: tiles, then norms
    unsigned long long *best = (unsigned long long *)(smem + (size_t)2 * CHB_U4 * 16);  // [8][WR]
    float *xn_s = (float *)(best + APF_WAVES * WR);                                    // [8][WR]
    uint32_t *cand = (uint32_t *)(xn_s + APF_WAVES * WR);                              // [8][APF_CAP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int64_t wrow0 = (int64_t)blockIdx.x * (APF_WAVES * WR) + (int64_t)wave * WR;
    unsigned long long *best_w = best + wave * WR;
    float *xn_w = xn_s + wave * WR;
    uint32_t *cbuf = cand + wave * APF_CAP;
    const int d = P.d;

    apf_stage<NM, APF_WAVES>(P.cbf, 0, buf);

    // the wave's rows as bf16 B-operands: lane (j, g), step s = columns 32 s + 8 g .. + 7 of row 16 xt + j -- and, from the same
    // registers, the canonical |x|^2 (k_assign's chain over the columns in natural order; the IP bound needs it too): the four lane
    // groups of a row take turns, eight columns each, and hand the running sum on (zero columns beyond d add nothing to a sum of
    // squares).  A chain per lane over its own row instead -- 64 rows x 512 bytes per load round against a 32 KB L1 -- read x from
    // L2 about eight times over.
    apf_bf16x8 B[XT][NM];
#pragma unroll
    for (int xt = 0; xt < XT; xt++) {
        const int64_t row = wrow0 + xt * 16 + j;
        float4 f0[NM], f1[NM];
#pragma unroll
        for (int s = 0; s < NM; s++) {
            const int col = 32 * s + 8 * g;
            f0[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            f1[s] = f0[s];
            if (row < P.n && col < d) {
                const float4 *p = (const float4 *)(P.x + row * d + col);
                f0[s] = p[0];
                f1[s] = p[1];
            }
        }
        float xn = 0.0f;
#pragma unroll
        for (int s = 0; s < NM; s++) {
            B[xt][s] = (apf_bf16x8){(__bf16)f0[s].x, (__bf16)f0[s].y, (__bf16)f0[s].z, (__bf16)f0[s].w,
                                    (__bf16)f1[s].x, (__bf16)f1[s].y, (__bf16)f1[s].z, (__bf16)f1[s].w};
#pragma unroll
            for (int gg = 0; gg < 4; gg++) {
                float t = xn;
                t = __fmaf_rn(f0[s].x, f0[s].x, t);
                t = __fmaf_rn(f0[s].y, f0[s].y, t);
                t = __fmaf_rn(f0[s].z, f0[s].z, t);
                t = __fmaf_rn(f0[s].w, f0[s].w, t);
                t = __fmaf_rn(f1[s].x, f1[s].x, t);
                t = __fmaf_rn(f1[s].y, f1[s].y, t);
                t = __fmaf_rn(f1[s].z, f1[s].z, t);
                t = __fmaf_rn(f1[s].w, f1[s].w, t);
                xn = __shfl(t, j + 16 * gg);  // the sum of the group whose turn it was
            }
        }
        if (g == 0) {
            xn_w[xt * 16 + j] = xn;
            best_w[xt * 16 + j] = ~0ull;  // "nothing yet"
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // (chunk 0 has landed: every wave waited for its own loads first)

    const int nch = P.nch;
    // a = x~.y~ - |y|^2 / 2 (IP: x~.y~) comes out of the instruction itself -- the accumulator starts at the norm term -- so the
    // key's approximation h = -2 a (IP: -a) is never formed: pass 1 keeps the MAXIMUM of a per row, pass 2 compares a with a bound
    // (two v_max3 / one compare per row tile and centroid tile instead of four fma and three min: the passes were VALU-bound)
    float hw[XT];  // pass 1: the row's running maximum of a; pass 2: its bound V (candidates: a >= V)
#pragma unroll
    for (int xt = 0; xt < XT; xt++) hw[xt] = -__builtin_inff();
    int cnt = 0;  // candidates parked by this wave (uniform)
    // set bits of a ballot below this lane: v_mbcnt (no per-lane mask to keep in two registers)
#define APF_BELOW(mk_) ((int)__builtin_amdgcn_mbcnt_hi((uint32_t)((mk_) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(mk_), 0u)))

    // LDS reads of the streamed operands as inline assembly: the compiler cannot tell a read of THIS chunk from the global_load_lds
    // writes into the OTHER buffer and put s_waitcnt vmcnt(0) -- the whole next chunk -- in front of the first read after every stage
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
#define APF_TILE_OPERANDS()                                                                          \
    apf_u32x4 a_[4];                                                                                 \
    f32x4 y4;                                                                                        \
    {                                                                                                \
        /* the two LDS addresses are rebuilt from the thread number for every tile (two VALU each against 32 MFMAs): kept in  */ \
        /* registers across the passes they were what the allocator spilled at 256 VGPRs, and the reload at the head of every */ \
        /* chunk of pass 2 sat behind a s_waitcnt vmcnt(0) that also waited for the prefetch just issued                      */ \
        uint32_t ta_, ya_;                                                                           \
        asm volatile("v_and_b32 %0, 63, %2\n\tv_lshl_add_u32 %0, %0, 4, %3\n\tv_and_b32 %1, 48, %2\n\tv_add_u32 %1, %1, %4" \
                     : "=&v"(ta_), "=&v"(ya_)                                                         \
                     : "v"(tid), "s"(cb_s + (uint32_t)t * (NM * 1024)), "s"(yb_s + (uint32_t)t * 64)); \
        asm volatile("ds_read_b128 %0, %1" : "=v"(a_[0]) : "v"(ta_));                                \
        if (NM > 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(a_[1]) : "v"(ta_));        \
        if (NM > 2) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a_[2]) : "v"(ta_));        \
        if (NM > 3) asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(a_[3]) : "v"(ta_));        \
        asm volatile("ds_read_b128 %0, %1" : "=v"(y4) : "v"(ya_));                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_[0]), "+v"(y4));                                \
        if (NM > 1) asm volatile("" : "+v"(a_[1]));                                                  \
        if (NM > 2) asm volatile("" : "+v"(a_[2]));                                                  \
        if (NM > 3) asm volatile("" : "+v"(a_[3]));                                                  \
    }                                                                                                \
    apf_bf16x8 A[NM];                                                                                \
    _Pragma("unroll") for (int s = 0; s < NM; s++) A[s] = __builtin_bit_cast(apf_bf16x8, a_[s]);
#define APF_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0)
#define APF_PRODUCTS(X0)                                                                             \
    f32x4 acc[4];                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 4; u++) acc[u] = y4;                                       \
    _Pragma("unroll") for (int s = 0; s < NM; s++)                                                   \
        _Pragma("unroll") for (int u = 0; u < 4; u++)                                                \
            acc[u] = APF_MFMA(A[s], B[(X0) + u][s], acc[u]);

    // ---- pass 1: the minimum of h per row ----
    for (int it = 0; it < nch; it++) {
        const int cur = it & 1;
        apf_stage<NM, APF_WAVES>(P.cbf, it + 1 < nch ? it + 1 : 0, buf + (size_t)(cur ^ 1) * CHB_U4);  // (the last one: chunk 0 of pass 2)
        const uint32_t cb_s = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)cur * (CHB_U4 * 16));
        const uint32_t yb_s = cb_s + CH_U4 * 16;
        for (int t = 0; t < APF_CH; t++) {
            APF_TILE_OPERANDS();
#pragma unroll
            for (int x0 = 0; x0 < XT; x0 += 4) {
                APF_PRODUCTS(x0);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    hw[x0 + u] = fmaxf(fmaxf(hw[x0 + u], fmaxf(acc[u][0], acc[u][1])), fmaxf(acc[u][2], acc[u][3]));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's share of the next chunk (global_load_lds) has landed ...
        __syncthreads();  // ... everybody's has, and everybody is done with this one
    }
    // the maximum of a row over its four lane groups, then the bound of pass 2.  In the key's scale h = -2 a (IP: -a): a centroid
    // with h > hmin + M, M = 2 E (IP: E as the bound is on the dot product itself... M = c (|x|^2 + max|y|^2)), cannot decide.
    {
        const float ynmax = *P.ynmax;
#pragma unroll
        for (int xt = 0; xt < XT; xt++) {
            float a = hw[xt];
            a = fmaxf(a, __shfl_xor(a, 16));
            a = fmaxf(a, __shfl_xor(a, 32));
            const float xnj = xn_w[xt * 16 + j];
            const float M = (L2 ? 2.0f * APF_C : APF_C) * (xnj + ynmax);
            // L2: h <= hmin + M  <=>  a >= amax - M / 2; and exact keys clamp at 0, where everything that may clamp ties:
            //     h <= M / 2 - |x|^2  <=>  a >= |x|^2 / 2 - M / 4
            // IP: h <= hmin + M  <=>  a >= amax - M
            float v = L2 ? a - 0.5f * M : a - M;
            if (L2) v = fminf(v, 0.5f * (xnj * 0.999999f) - 0.25f * M);
            // (a row beyond n is all zeros: a = -|y|^2 / 2 never reaches +inf; a NaN does, and its candidate lands in a slot nobody reads)
            if (!(wrow0 + xt * 16 + j < P.n)) v = __builtin_inff();
            hw[xt] = v;
        }
    }
    // ---- pass 2: the same products against the bound ----
#ifdef APF_V_NOPASS2
    if (P.assign && hw[0] + hw[XT - 1] == 12345.0f) P.assign[0] = (int64_t)hw[1];
    for (int it = 0; it < 0; it++) {
#else
    for (int it = 0; it < nch; it++) {
#endif
        const int cur = (nch + it) & 1;
        if (it + 1 < nch) apf_stage<NM, APF_WAVES>(P.cbf, it + 1, buf + (size_t)(cur ^ 1) * CHB_U4);
        const uint32_t cb_s = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)cur * (CHB_U4 * 16));
        const uint32_t yb_s = cb_s + CH_U4 * 16;
        for (int t = 0; t < APF_CH; t++) {
            APF_TILE_OPERANDS();
            const int cidx0 = ((it * APF_CH + t) << 4) + 4 * g;
#pragma unroll
            for (int x0 = 0; x0 < XT; x0 += 4) {
                APF_PRODUCTS(x0);  // (sixteen products in flight; the tests follow tile by tile: a test per tile passes 4x less often)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int xt = x0 + u;
                    APF_STAT(0, 1);
                    if (__ballot(!(fmaxf(fmaxf(acc[u][0], acc[u][1]), fmaxf(acc[u][2], acc[u][3])) < hw[xt]))) {
                        APF_STAT(1, 1);
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const bool p = !(acc[u][r] < hw[xt]);
                            const uint64_t mk = __ballot(p);
                            if (cnt > APF_CAP - 64) {
                                apf_flush<L2>(P.x, P.c, P.cnorm, P.n, P.m, d, wrow0, cbuf, cnt, best_w, xn_w, P.ids);
                                cnt = 0;
                            }
                            if (p) cbuf[cnt + APF_BELOW(mk)] = ((uint32_t)(xt * 16 + j) << 24) | (uint32_t)(cidx0 + r);
                            cnt += __popcll(mk);
                            APF_STAT(2, __popcll(mk));
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (as in pass 1: the staged chunk is complete before the barrier publishes it)
        __syncthreads();
    }
#undef APF_TILE_OPERANDS
#undef APF_PRODUCTS
    // The end of the list.  A row with ONE candidate needs no exact key when only the assignment is asked for (the candidates
    // provably include the nearest centroid): the Lloyd iterations and the final assignment of a build pass val = nullptr, and on
    // clustered data three rows in four are such rows.  The chunk buffers are free now (everybody is past the last barrier): per wave
    // a candidate count and the last candidate of every row.
    int *ncand = (int *)smem + wave * (2 * WR), *onlyc = ncand + WR;
    for (int rr = lane; rr < WR; rr += 64) ncand[rr] = 0;
    for (int e = lane; e < cnt; e += 64) {
        const uint32_t pk = cbuf[e];
        const int xl = (int)(pk >> 24), ci = (int)(pk & 0xFFFFFFu);
        if (ci < P.m) {  // (a padding row of the last centroid tile is a candidate only under a non-finite bound)
            atomicAdd(&ncand[xl], 1);
            onlyc[xl] = ci;
        }
    }
    if (!P.want_keys) {
        // keep the pairs of rows with several candidates (or with keys from an earlier flush), in place, batch by batch
        int kept = 0;
        for (int base = 0; base < cnt; base += 64) {
            const int e = base + lane;
            const uint32_t pk = e < cnt ? cbuf[e] : 0u;
            const int xl = (int)(pk >> 24);
            const bool need = e < cnt && (ncand[xl] > 1 || best_w[xl] != ~0ull);
            const uint64_t mk = __ballot(need);
            if (need) cbuf[kept + APF_BELOW(mk)] = pk;
            kept += __popcll(mk);
        }
        cnt = kept;
    }
    apf_exact<L2, true>(P.x, P.c, P.cnorm, P.n, P.m, d, wrow0, cbuf, cnt, best_w, xn_w, P.ids);
    for (int rr = lane; rr < WR; rr += 64) {
        const int64_t row = wrow0 + rr;
        if (row >= P.n) continue;
        const unsigned long long v = best_w[rr];
        const uint32_t o = (uint32_t)(v >> 32);
        const bool only = v == ~0ull && !P.want_keys && ncand[rr] == 1;
        const uint32_t low1 = only ? (P.ids ? (uint32_t)P.ids[onlyc[rr]] : (uint32_t)onlyc[rr]) : 0u;
        if (P.packed_out) P.packed_out[row] = only ? (unsigned long long)low1 : v;
        if (P.assign) P.assign[row] = only ? (int64_t)low1 : v == ~0ull ? -1 : (int64_t)(v & 0xFFFFFFFFull);
        if (P.val) P.val[row] = L2 ? __uint_as_float(o) : ip_from_ord(o);
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
bool qk_assign_pf_supported(int64_t n, int64_t m, int d, int metric) {
    if (qk_env_set("QK_NO_ASSIGN_PF")) return false;
    if (metric != QK_METRIC_L2 && metric != QK_METRIC_IP) return false;
    // (a workgroup is 512 or 1024 rows: with few rows the chip is not filled; under 64 centroids there is nothing to filter)
#ifndef APF_MIN_ROWS
#define APF_MIN_ROWS 40960  /* measured: 32768 rows tie with the fp32 kernel, 49152 win by 7-25 % (scripts/assign_crossover.py) */
#endif
    return d % 8 == 0 && d >= 8 && d <= 128 && m >= 64 && m <= (1 << 24) && n >= APF_MIN_ROWS;
}

template <int NM, int XT, int WAVES>
static int apf_launch(hipStream_t st, const AssignPfParams &p, int metric) {
    constexpr int WR = XT * 16;
    const size_t lds = (size_t)2 * (APF_CH * NM * 1024 + 1024) + (size_t)WAVES * WR * 12 + (size_t)WAVES * APF_CAP * 4;
    const unsigned grid = (unsigned)((p.n + WAVES * WR - 1) / (WAVES * WR));
    if (metric == QK_METRIC_L2) {
        QK_HIP(hipFuncSetAttribute((const void *)k_assign_pf<NM, XT, WAVES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_assign_pf<NM, XT, WAVES, true>), dim3(grid), dim3(64 * WAVES), lds, st, p);
    } else {
        QK_HIP(hipFuncSetAttribute((const void *)k_assign_pf<NM, XT, WAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_assign_pf<NM, XT, WAVES, false>), dim3(grid), dim3(64 * WAVES), lds, st, p);
    }
    QK_HIP(hipGetLastError());
    return QK_OK;
}

size_t qk_assign_pf_scratch_bytes(int64_t m, int d) {
    const int NM = (d + 31) / 32;
    const int64_t nch = ((m + 15) / 16 + APF_CH - 1) / APF_CH;
    return (((size_t)nch * ((size_t)APF_CH * NM * 1024 + 1024) + 255) & ~(size_t)255) + 256;
}

// the launches; scratch: qk_assign_pf_scratch_bytes(m, d) bytes of device memory, 256-byte aligned
int qk_assign_pf_launch(qk_ctx *ctx, const float *x, int64_t n, const float *c, int64_t m, int d, int metric, const float *cnorm,
                        const int64_t *ids, int64_t *assign, float *val, unsigned long long *packed_out, bool want_keys, void *scratch) {
    if (!qk_assign_pf_supported(n, m, d, metric)) QK_FAIL(QK_ERR_UNSUPPORTED, "assign (prefiltered): unsupported shape");
    hipStream_t st = ctx->stream;
    const int NM = (d + 31) / 32;
    const int64_t mt = (m + 15) / 16;
    const int nch = (int)((mt + APF_CH - 1) / APF_CH);
    const int64_t mtp = (int64_t)nch * APF_CH;
    const size_t cbf_bytes = (size_t)nch * ((size_t)APF_CH * NM * 1024 + 1024);
    uint4 *cbf = (uint4 *)scratch;
    unsigned int *ynmax = (unsigned int *)((unsigned char *)scratch + ((cbf_bytes + 255) & ~(size_t)255));
    QK_HIP(hipMemsetAsync(ynmax, 0, 4, st));
    hipLaunchKernelGGL(k_apf_prep, dim3((unsigned)mtp), dim3(64), 0, st, c, cnorm, (int)m, d, NM, metric == QK_METRIC_L2 ? 1 : 0, cbf, ynmax);
    AssignPfParams p;
    p.x = x;
    p.n = n;
    p.d = d;
    p.cbf = cbf;
    p.c = c;
    p.cnorm = cnorm;
    p.ynmax = (const float *)ynmax;
    p.m = (int)m;
    p.nch = nch;
    p.assign = assign;
    p.val = val;
    p.ids = ids;
    p.packed_out = packed_out;
    p.want_keys = (want_keys || val) ? 1 : 0;
    const int num_cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const bool big = n >= (int64_t)num_cus * 1024;  // 1024-row workgroups once they fill the chip
#ifdef APF_V_W16
#define APF_BIG_ 4, 16
#else
#define APF_BIG_ 8, 8
#endif
#define APF_CASE(NM_)                                  \
    case NM_:                                          \
        return big ? apf_launch<NM_, APF_BIG_>(st, p, metric) : apf_launch<NM_, 4, 8>(st, p, metric);
#ifdef APF_PROBE_STATS
    unsigned long long z[4] = {0, 0, 0, 0};
    QK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(apf_stats), z, sizeof(z)));
    int rc_ = QK_ERR_UNSUPPORTED;
    switch (NM) {
        case 1: rc_ = big ? apf_launch<1, APF_BIG_>(st, p, metric) : apf_launch<1, 4, 8>(st, p, metric); break;
        case 2: rc_ = big ? apf_launch<2, APF_BIG_>(st, p, metric) : apf_launch<2, 4, 8>(st, p, metric); break;
        case 3: rc_ = big ? apf_launch<3, APF_BIG_>(st, p, metric) : apf_launch<3, 4, 8>(st, p, metric); break;
        case 4: rc_ = big ? apf_launch<4, APF_BIG_>(st, p, metric) : apf_launch<4, 4, 8>(st, p, metric); break;
    }
    QK_HIP(hipStreamSynchronize(st));
    QK_HIP(hipMemcpyFromSymbol(z, HIP_SYMBOL(apf_stats), sizeof(z)));
    fprintf(stderr, "[apf] n=%lld m=%lld groups=%llu slow=%llu (%.3f) candidates=%llu (%.2f per row)\n", (long long)n, (long long)m, z[0], z[1],
            z[0] ? (double)z[1] / z[0] : 0.0, z[2], (double)z[2] / n);
    return rc_;
#endif
    switch (NM) {
        APF_CASE(1)
        APF_CASE(2)
        APF_CASE(3)
        APF_CASE(4)
    }
#undef APF_CASE
    return QK_ERR_UNSUPPORTED;
}


// nearest centroid of every row of x (k-means); c: [m][d] row-major, cnorm: canonical norms of its rows (device pointers throughout)
int qk_assign_pf_device(qk_ctx *ctx, const float *x, int64_t n, const float *c, int64_t m, int d, int metric, const float *cnorm,
                        int64_t *assign, float *val) {
    const size_t need = qk_assign_pf_scratch_bytes(m, d);
    QK_TRY(qk_ws_reserve(ctx, need + 4096));
    void *scratch = qk_ws_alloc(ctx, need);
    if (!scratch) QK_FAIL(QK_ERR_OOM, "assign (prefiltered): workspace exhausted");
    return qk_assign_pf_launch(ctx, x, n, c, m, d, metric, cnorm, nullptr, assign, val, nullptr, val != nullptr, scratch);
}

// qk_merge.hip -- the merge stage of the partition scan: the records the scan kernels left per (query, list) pair become the
// query's top k (global TopkBuffer::batch_add + output / padding, src/cpp/src/query_coordinator.cpp:752-788), and the cross-rank
// merge of the sharded path (worker_scan's cross-worker merge, :436-460).  Split from qk_scan.hip in round 3 (one translation
// unit of 2500 lines held kernels, host dispatch and probe plumbing of three stages).
#include "qk_internal.h"
#include <vector>
#include <climits>
#include "qk_device.h"
#include "qk_scan_types.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

// ---- merge kernel: one wave per query ---------------------------------------------------------------------

template <int MAXCH>
__device__ __forceinline__ void merge_chain_body(const MergeParams &M) {
    long long ck[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // start, slot line, fetch, bound, consume, final sort, end; [7] = records
    if (M.clock) ck[0] = wall_clock64();
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int k = M.k, Cm = M.Cm;
    int64_t *pool_id = (int64_t *)smem;
    uint32_t *pool_ord = (uint32_t *)(smem + (size_t)Cm * 8);
    uint32_t tau = 0xFFFFFFFFu;
    int cnt = 0;
    // one record: header + first 64 entries are requested together (entries beyond the count are ignored)
    // one record: header, the first 128 entries (two per lane) and the k-th key are requested together, so that nothing in
    // the merge of a record waits for a load that depends on another one (k = 100: 8.4 -> ~4 us per record)
    auto fetch2 = [&](int rr, int2 &h, uint32_t &o, int64_t &dd, uint32_t &o1, int64_t &dd1, uint32_t &okth) {
        h = make_int2(-1, 0);
        o = o1 = okth = 0xFFFFFFFFu;
        dd = dd1 = -1;
        if (rr >= 0 && rr < M.max_recs) {
            h = M.rec_hdr[rr];
            if (lane < k) {
                o = M.rec_ord[(int64_t)rr * k + lane];
                dd = M.rec_id[(int64_t)rr * k + lane];
            }
            if (lane + 64 < k) {
                o1 = M.rec_ord[(int64_t)rr * k + lane + 64];
                dd1 = M.rec_id[(int64_t)rr * k + lane + 64];
            }
            okth = M.rec_ord[(int64_t)rr * k + k - 1];  // (only meaningful for a full record)
        }
    };
    auto fetch = [&](int rr, int2 &h, uint32_t &o, int64_t &dd) {
        uint32_t o1, okth;
        int64_t d1;
        fetch2(rr, h, o, dd, o1, d1, okth);
    };
    auto consume2 = [&](int rec, const int2 hdr, const uint32_t o0, const int64_t d0, const uint32_t o1, const int64_t d1,
                        const uint32_t okth, const bool have1) {
        const int n = hdr.y;
        // a full record is sorted and holds k entries: its k-th key bounds the answer before anything is pooled
        if (n >= k) tau = min(tau, have1 ? okth : (k <= 64 ? (uint32_t)__builtin_amdgcn_readlane(o0, k - 1) : M.rec_ord[(int64_t)rec * k + k - 1]));
        for (int base = 0; base < n; base += 64) {
            const int e = base + lane;
            const bool has = e < n;
            uint32_t o = o0;
            int64_t dd = d0;
            if (base == 64 && have1) {
                o = o1;
                dd = d1;
            } else if (base > 0) {
                o = has ? M.rec_ord[(int64_t)rec * k + e] : 0xFFFFFFFFu;
                dd = has ? M.rec_id[(int64_t)rec * k + e] : -1;
            }
            const bool pass = has && o <= tau;
            const uint64_t m = __ballot(pass);
            if (m) {
                if (pass) {
                    const int sl = cnt + __popcll(m & ((1ull << lane) - 1ull));
                    pool_ord[sl] = o;
                    pool_id[sl] = dd;
                }
                cnt += __popcll(m);
                if (cnt > Cm - 64) {  // (unsorted k best + their bound; the final compaction sorts)
                    uint32_t kth;
                    cnt = select_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane, kth);
                    if (cnt >= k) tau = min(tau, kth);
                }
            }
            // records are sorted ascending: once a valid lane fails the bound, the rest of the record fails too
            if (__popcll(m) < min(64, n - base)) break;
        }
    };
    auto consume = [&](int rec, const int2 hdr, const uint32_t o0, const int64_t d0) { consume2(rec, hdr, o0, d0, 0u, 0, 0u, false); };
    // The slot lines of up to 64 pairs are fetched into LDS in one go (they are contiguous: [q * P, q * P + P) x 32 ints): one
    // memory round trip per 64 pairs instead of one per pair in front of every record fetch.  (Requesting the records of pair
    // r + 1 before merging those of pair r as well was measured SLOWER -- 22 -> 29 us at nprobe 16: the copies of eight
    // records' registers cost more than the round trip they hide.)
    int *s_slots = (int *)(smem + (((size_t)Cm * 12 + 15) & ~(size_t)15));
    for (int r0 = 0; r0 < M.P; r0 += 64) {
    const int pb = min(64, M.P - r0);
    for (int i = lane; i < pb * QK_SLOTS; i += 64) s_slots[i] = M.pair_slots[(q * M.P + r0) * QK_SLOTS + i];
    __syncthreads();
    for (int rb = 0; rb < pb; rb++) {
        const int r = r0 + rb;
        const int64_t pair = q * M.P + r;
        // slot line of the pair: lane 0 = record count, lanes 1..31 = the first records
        const int sv = lane < QK_SLOTS ? s_slots[rb * QK_SLOTS + lane] : 0;
        const int nrecs = __builtin_amdgcn_readlane(sv, 0);
        const int ns = min(nrecs, QK_SLOTS - 1);
        if (M.clock) {
            ck[1] = wall_clock64();
            ck[7] += nrecs;
        }
        int2 hdr, nhdr;
        uint32_t o0, no0;
        int64_t d0, nd0;
        int rec;
        for (int g0 = 0; g0 < ns; g0 += 8) {
            // eight listed records are requested before the first of them is merged: one memory round trip per group
            int recs[8];
            int2 hs[8];
            uint32_t os[8], os1[8], oks[8];
            int64_t ds[8], ds1[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                recs[i] = g0 + i < ns ? __shfl(sv, g0 + i + 1) : -1;
                fetch2(recs[i], hs[i], os[i], ds[i], os1[i], ds1[i], oks[i]);
            }
            // (an exact bound from the fetched keys -- bisection over the key registers -- was tried here: 13 us per use in
            //  this one-wave kernel, more than the select_pool calls it saves at k <= 32; wide k goes to k_merge_wide)
            if (M.clock) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ck[2] = ck[3] = wall_clock64();
            }
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (recs[i] >= 0 && recs[i] < M.max_recs) consume2(recs[i], hs[i], os[i], ds[i], os1[i], ds1[i], oks[i], true);
        }
        if (nrecs > QK_SLOTS - 1) {  // overflow: chained records (header names the next one)
            rec = M.pair_head[pair];
            fetch(rec, hdr, o0, d0);
            while (rec >= 0 && rec < M.max_recs) {
                const int nrec = hdr.x;
                fetch(nrec, nhdr, no0, nd0);
                consume(rec, hdr, o0, d0);
                rec = nrec;
                hdr = nhdr;
                o0 = no0;
                d0 = nd0;
            }
        }
    }
    __syncthreads();
    }
    if (M.clock) ck[4] = wall_clock64();
    cnt = compact_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane);
    if (M.clock) ck[5] = wall_clock64();
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
    if (M.clock && lane == 0) {
        ck[6] = wall_clock64();
        for (int i = 0; i < 8; i++) M.clock[q * 8 + i] = ck[i];
    }
}

template <int MAXCH>
__global__ __launch_bounds__(64) void k_merge(MergeParams M) {
    merge_chain_body<MAXCH>(M);
}

// FLAT form (k <= 32, P <= 64, no chained records): the walk above is a chain of dependent round trips -- per pair: its records,
// then their merge (nprobe 16: 17 of them, 36 us).  Here the records of ALL pairs of the query are listed first (slot lines in
// LDS), the bound comes from the k-th keys of the full records in one round trip, and the entries are read 64 / k records per
// load instruction (lane = (record, entry)), the next batch in flight while the current one is pooled.  Same pooling, same
// final compact_pool under (key, id): the same answer.
template <int MAXCH>
__global__ __launch_bounds__(64) void k_merge_flat(MergeParams M) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int k = M.k, Cm = M.Cm, P = M.P;
    int64_t *pool_id = (int64_t *)smem;
    uint32_t *pool_ord = (uint32_t *)(smem + (size_t)Cm * 8);
    int *s_slots = (int *)(smem + (((size_t)Cm * 12 + 15) & ~(size_t)15));  // [P][32]
    int *s_recs = s_slots + 64 * QK_SLOTS;                                  // [<= P * 31]
    int *s_recn = s_recs + 64 * (QK_SLOTS - 1);
    if (P == 1 && M.pair_slots[q * QK_SLOTS] == 0) {
        // per-pair results (the rounds of a recall-target search): most pairs of a round leave no record at all -- the slot is
        // padding, or the running bound kept everything out -- and their answer is k empty entries
        for (int e2 = lane; e2 < k; e2 += 64) {
            M.out_ids[q * k + e2] = -1;
            if (M.out_dist) M.out_dist[q * k + e2] = M.metric == QK_METRIC_IP ? -INFINITY : INFINITY;
        }
        return;
    }
    for (int i = lane; i < P * QK_SLOTS; i += 64) s_slots[i] = M.pair_slots[q * P * QK_SLOTS + i];
    __syncthreads();
    const int n_p_raw = lane < P ? s_slots[lane * QK_SLOTS] : 0;
    if (__ballot(n_p_raw > QK_SLOTS - 1)) {  // chained records somewhere: the general walk
        __syncthreads();
        merge_chain_body<MAXCH>(M);
        return;
    }
    int inc = n_p_raw;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    const int R = __shfl(inc, 63);
    {
        const int off = inc - n_p_raw;
        for (int j = 0; j < n_p_raw; j++) s_recs[off + j] = s_slots[lane * QK_SLOTS + 1 + j];
    }
    __syncthreads();
    // bound: the smallest k-th key among the full records; record sizes kept for the entry pass
    uint32_t tau = 0xFFFFFFFFu;
    for (int r = lane; r < R; r += 64) {
        const int rec = s_recs[r];
        int n = 0;
        if (rec >= 0 && rec < M.max_recs) {
            n = M.rec_hdr[rec].y;
            if (n >= k) tau = min(tau, M.rec_ord[(int64_t)rec * k + k - 1]);
        }
        s_recn[r] = n;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tau = min(tau, (uint32_t)__shfl_xor((int)tau, o));
    __syncthreads();
    const int rpb = 64 / k;  // records per load instruction
    const int j = lane / k, e = lane - j * k;
    const bool act = j < rpb;
    const int nb = (R + rpb - 1) / rpb;
    int cnt = 0;
    auto load = [&](int b, uint32_t &o, int64_t &dd) -> bool {
        const int r = b * rpb + j;
        bool has = false;
        o = 0xFFFFFFFFu;
        dd = -1;
        if (act && r < R) {
            const int rec = s_recs[r];
            if (e < s_recn[r]) {
                has = true;
                o = M.rec_ord[(int64_t)rec * k + e];
                dd = M.rec_id[(int64_t)rec * k + e];
            }
        }
        return has;
    };
    uint32_t o_cur = 0xFFFFFFFFu, o_nxt;
    int64_t d_cur = -1, d_nxt;
    bool h_cur = nb > 0 ? load(0, o_cur, d_cur) : false, h_nxt;
    for (int b = 0; b < nb; b++) {
        h_nxt = b + 1 < nb ? load(b + 1, o_nxt, d_nxt) : false;
        const bool pass = h_cur && o_cur <= tau;
        const uint64_t m = __ballot(pass);
        if (m) {
            if (pass) {
                const int sl = cnt + __popcll(m & ((1ull << lane) - 1ull));
                pool_ord[sl] = o_cur;
                pool_id[sl] = d_cur;
            }
            cnt += __popcll(m);
            if (cnt > Cm - 64) {
                uint32_t kth;
                cnt = select_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane, kth);
                if (cnt >= k) tau = min(tau, kth);
            }
        }
        o_cur = o_nxt;
        d_cur = d_nxt;
        h_cur = h_nxt;
    }
    cnt = compact_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane);
    for (int e2 = lane; e2 < k; e2 += 64) {
        int64_t oid = -1;
        float od = M.metric == QK_METRIC_IP ? -INFINITY : INFINITY;
        if (e2 < cnt) {
            oid = pool_id[e2];
            const uint32_t o = pool_ord[e2];
            if (M.metric == QK_METRIC_L2) {
                const float d2 = __uint_as_float(o);
                od = M.sqrt_l2 ? sqrtf(d2) : d2;
            } else {
                od = ip_from_ord(o);
            }
        }
        M.out_ids[q * k + e2] = oid;
        if (M.out_dist) M.out_dist[q * k + e2] = od;
    }
}

// ---- merge kernel for wide k: one workgroup of 4 waves per query ------------------------------------------------
// k_merge is one wave per query: its selects and rank sorts are chains of dependent readlane / ballot steps, and with one wave
// per SIMD nothing hides them (QK_MERGE_CLOCK, k = 100, 5 records per query: bound 13 us + final sort 12-32 us of a 50 us
// launch).  Here the records of a query are pooled in LDS by 256 threads -- bounded by the smallest k-th key of the full
// records -- and the pool is sorted under the (key, id) order by a bitonic network; whenever the next round of records might
// not fit, the pool is cut back to its k best first.
#define QK_MW_CAP 2048
__device__ __forceinline__ void mw_sort(uint32_t *keys, int64_t *ids, int n_pad, int tid) {
    for (int size = 2; size <= n_pad; size <<= 1) {
        for (int ls = 31 - __builtin_clz(size) - 1; ls >= 0; ls--) {
            const int stride = 1 << ls;
            for (int t = tid; t < (n_pad >> 1); t += 256) {
                const int lo = ((t >> ls) << (ls + 1)) + (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;  // ascending block
                const uint32_t ka = keys[lo], kb = keys[hi];
                const int64_t ia = ids[lo], ib = ids[hi];
                const bool a_gt_b = ka > kb || (ka == kb && ia > ib);
                if (a_gt_b == up) {
                    keys[lo] = kb;
                    keys[hi] = ka;
                    ids[lo] = ib;
                    ids[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void k_merge_wide(MergeParams M) {
    __shared__ int64_t s_ids[QK_MW_CAP];
    __shared__ uint32_t s_keys[QK_MW_CAP];
    __shared__ int s_rec[256], s_recn[256];
    __shared__ int s_cnt, s_nrec, s_chain, s_qual;
    __shared__ uint32_t s_tau, s_b2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x;
    const int k = M.k;
    long long ck[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // probe: start, record list, bound, entries pooled, -, sorted, end; [7] = records
    if (M.clock) ck[0] = wall_clock64();
    if (tid == 0) {
        s_cnt = 0;
        s_tau = 0xFFFFFFFFu;
    }
    // keep the k best of the pool, sorted; the k-th key becomes the bound
    auto cut = [&]() {
        __syncthreads();
        const int cnt = s_cnt;
        int n_pad = 64;
        while (n_pad < cnt) n_pad <<= 1;
        for (int t = cnt + tid; t < n_pad; t += 256) {
            s_keys[t] = 0xFFFFFFFFu;
            s_ids[t] = LLONG_MAX;
        }
        __syncthreads();
        mw_sort(s_keys, s_ids, n_pad, tid);
        if (tid == 0) {
            const int nn = min(cnt, k);
            s_cnt = nn;
            if (nn >= k) s_tau = min(s_tau, s_keys[k - 1]);
        }
        __syncthreads();
    };
    // pool the records listed in s_rec[0 .. s_nrec)
    auto pool_records = [&]() {
        if (tid == 0) {
            s_qual = 0;
            s_b2 = 0;
        }
        __syncthreads();
        const int nrec = s_nrec;
        if (M.clock) {
            ck[1] = wall_clock64();
            ck[7] += nrec;
        }
        // bounds (records are sorted): the k-th key of every full record; and, with c = ceil(k / records), the largest c-th key
        // -- the first c entries of every record are >= k entries that do not exceed it.  The second one is what keeps the
        // pool near k entries when a partition was cut into several segments (each record then holds the best of a part)
        const int c2 = nrec > 0 ? (k + nrec - 1) / nrec : k;
        for (int i = tid; i < nrec; i += 256) {
            const int rec = s_rec[i];
            const uint32_t kth = M.rec_ord[(int64_t)rec * k + k - 1];  // (requested with the header; meaningful for a full record)
            const uint32_t cth = M.rec_ord[(int64_t)rec * k + c2 - 1];
            const int n = min(M.rec_hdr[rec].y, k);
            s_recn[i] = n;
            if (n >= k) atomicMin(&s_tau, kth);
            if (n >= c2) {
                atomicAdd(&s_qual, 1);
                atomicMax(&s_b2, cth);
            }
        }
        __syncthreads();
        if (tid == 0 && (long long)s_qual * c2 >= k) s_tau = min(s_tau, s_b2);
        __syncthreads();
        if (M.clock) ck[2] = wall_clock64();
        const int per_round = max(1, (QK_MW_CAP - k) / k);  // records that fit next to k kept entries
        for (int r0 = 0; r0 < nrec; r0 += per_round) {
            const int r1 = min(nrec, r0 + per_round);
            if (s_cnt + (r1 - r0) * k > QK_MW_CAP) cut();  // (uniform: s_cnt is read after a barrier)
            const uint32_t tau = s_tau;
            for (int i = r0 + wave; i < r1; i += 4) {  // one wave per record; records are sorted, so a wave stops at the bound
                const int rec = s_rec[i], n = s_recn[i];
                // the first two chunks are requested together (k <= 128: the whole record in one round trip)
                uint32_t oA = 0xFFFFFFFFu, oB = 0xFFFFFFFFu;
                int64_t dA = -1, dB = -1;
                if (lane < n) {
                    oA = M.rec_ord[(int64_t)rec * k + lane];
                    dA = M.rec_id[(int64_t)rec * k + lane];
                }
                if (lane + 64 < n) {
                    oB = M.rec_ord[(int64_t)rec * k + lane + 64];
                    dB = M.rec_id[(int64_t)rec * k + lane + 64];
                }
                for (int base = 0; base < n; base += 64) {
                    const int e = base + lane;
                    uint32_t o = base == 0 ? oA : oB;
                    int64_t dd = base == 0 ? dA : dB;
                    if (base >= 128) {
                        o = 0xFFFFFFFFu;
                        dd = -1;
                        if (e < n) {
                            o = M.rec_ord[(int64_t)rec * k + e];
                            dd = M.rec_id[(int64_t)rec * k + e];
                        }
                    }
                    const bool pass = e < n && o <= tau;
                    const uint64_t m = __ballot(pass);
                    if (m) {
                        int slot0 = 0;
                        if (lane == 0) slot0 = atomicAdd(&s_cnt, __popcll(m));
                        slot0 = __builtin_amdgcn_readfirstlane(slot0);
                        if (pass) {
                            const int sl = slot0 + __popcll(m & ((1ull << lane) - 1ull));
                            s_keys[sl] = o;
                            s_ids[sl] = dd;
                        }
                    }
                    if (__popcll(m) < min(64, n - base)) break;
                }
            }
            __syncthreads();
        }
    };
    for (int r0 = 0; r0 < M.P; r0 += 8) {
        // slot lines of 8 pairs: entry 0 = record count, entries 1..31 = the first records
        if (tid == 0) {
            s_nrec = 0;
            s_chain = 0;
        }
        __syncthreads();
        const int pr = tid >> 5, sl = tid & 31;
        const bool pv = r0 + pr < M.P;
        const int64_t pair = q * M.P + r0 + pr;
        const int sv = pv ? M.pair_slots[pair * QK_SLOTS + sl] : 0;
        const int nrecs = __shfl(sv, lane & 32);
        if (pv && sl >= 1 && sl - 1 < min(nrecs, QK_SLOTS - 1) && sv >= 0 && sv < M.max_recs) s_rec[atomicAdd(&s_nrec, 1)] = sv;
        if (pv && sl == 0 && nrecs > QK_SLOTS - 1) atomicOr(&s_chain, 1 << pr);
        pool_records();
        // overflow chains (more than 31 records of one pair): walked by one thread, pooled 256 at a time
        int chain = s_chain;
        while (chain) {
            const int pc = __ffs(chain) - 1;
            chain &= chain - 1;
            int rec = M.pair_head[q * M.P + r0 + pc];
            while (rec >= 0 && rec < M.max_recs) {
                __syncthreads();
                if (tid == 0) {
                    int n = 0;
                    while (rec >= 0 && rec < M.max_recs && n < 256) {
                        s_rec[n++] = rec;
                        rec = M.rec_hdr[rec].x;
                    }
                    s_nrec = n;
                    s_chain = rec;
                }
                __syncthreads();
                rec = s_chain;
                pool_records();
            }
        }
        __syncthreads();
    }
    if (M.clock) ck[3] = ck[4] = wall_clock64();
    cut();
    if (M.clock) ck[5] = wall_clock64();
    const int cnt = s_cnt;
    for (int e = tid; e < k; e += 256) {
        int64_t oid = -1;
        float od = M.metric == QK_METRIC_IP ? -INFINITY : INFINITY;
        if (e < cnt) {
            oid = s_ids[e];
            const uint32_t o = s_keys[e];
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
    if (M.clock && tid == 0) {
        ck[6] = wall_clock64();
        for (int i = 0; i < 8; i++) M.clock[q * 8 + i] = ck[i];
    }
}

// ---- cross-rank merge: [G][Q][k] per-rank results -> [Q][k] ---------------------------------------
// in_key are SQUARED L2 distances / inner products (what qk_search returns with qk_ctx_set_squared_l2), so the
// merge runs on the same (key, id) order as the single-GPU path; sqrt is applied to the output.
// rank r's results start at in_ids + r * id_stride / in_key + r * key_stride (BYTE strides: the plain form passes Q*k*8 / Q*k*4,
// the packed form of the one-all-to-all exchange the block size for both, qk_topk_block_bytes)
template <int MAXCH>
__global__ __launch_bounds__(64) void k_merge_ranks(const int64_t *__restrict__ in_ids0, const float *__restrict__ in_key0, int G,
                                                    size_t id_stride, size_t key_stride, int64_t Q, int k, int Cm, int metric,
                                                    int sqrt_l2, int64_t *out_ids, float *out_dist) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    int64_t *pool_id = (int64_t *)smem;
    uint32_t *pool_ord = (uint32_t *)(smem + (size_t)Cm * 8);
    uint32_t tau = 0xFFFFFFFFu;
    int cnt = 0;
    for (int r = 0; r < G; r++) {
        const int64_t base0 = q * k;
        const int64_t *__restrict__ in_ids = (const int64_t *)((const unsigned char *)in_ids0 + (size_t)r * id_stride);
        const float *__restrict__ in_key = (const float *)((const unsigned char *)in_key0 + (size_t)r * key_stride);
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
                    uint32_t kth;
                    cnt = select_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane, kth);
                    if (cnt >= k) tau = min(tau, kth);
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

// The same merge for k beyond the LDS pools (k > 960; the single-store search serves k up to 8192, so must a device group and the
// RCCL exchange): every rank's k entries arrive SORTED under the (key, id) order with the -1 padding behind them, so the merged
// position of an entry is its own position plus, for every other rank, the number of that rank's entries in front of it -- a
// binary search each (merge path), no pool, no selection.  One workgroup per query; entry (r, i) goes straight to its place when
// that is below k.  An id held by two ranks (the stores do not enforce unique ids) keeps the lower rank's copy in front.
__global__ __launch_bounds__(256) void k_merge_ranks_large(const int64_t *__restrict__ in_ids0, const float *__restrict__ in_key0, int G,
                                                           size_t id_stride, size_t key_stride, int64_t Q, int k, int metric, int sqrt_l2,
                                                           int64_t *out_ids, float *out_dist) {
    __shared__ int nvalid[64];
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x, base0 = q * k;
    auto ids_of = [&](int r) { return (const int64_t *)((const unsigned char *)in_ids0 + (size_t)r * id_stride) + base0; };
    auto key_of = [&](int r) { return (const float *)((const unsigned char *)in_key0 + (size_t)r * key_stride) + base0; };
    if (tid < G) {  // valid entries of rank tid: the first -1 id (padding is a suffix)
        const int64_t *ids = ids_of(tid);
        int lo = 0, hi = k;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ids[mid] >= 0) lo = mid + 1; else hi = mid;
        }
        nvalid[tid] = lo;
    }
    __syncthreads();
    int total = 0;
    for (int r = 0; r < G; r++) total += nvalid[r];
    for (int64_t e = tid; e < (int64_t)G * k; e += 256) {
        const int r = (int)(e / k), i = (int)(e - (int64_t)r * k);
        if (i >= nvalid[r]) continue;
        const int64_t id = ids_of(r)[i];
        const float v = key_of(r)[i];
        const uint32_t o = metric == QK_METRIC_L2 ? ord_from_l2(v) : ord_from_ip(v);
        int rank = i;
        for (int r2 = 0; r2 < G && rank < k; r2++) {
            if (r2 == r) continue;
            const int64_t *ids2 = ids_of(r2);
            const float *key2 = key_of(r2);
            int lo = 0, hi = nvalid[r2];
            while (lo < hi) {  // entries of r2 in front of (o, id): strictly smaller, or equal when r2 is the lower rank
                const int mid = (lo + hi) >> 1;
                const uint32_t o2 = metric == QK_METRIC_L2 ? ord_from_l2(key2[mid]) : ord_from_ip(key2[mid]);
                const int64_t id2 = ids2[mid];
                const bool front = o2 < o || (o2 == o && (id2 < id || (id2 == id && r2 < r)));
                if (front) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_ids[base0 + rank] = id;
            if (out_dist) {
                float od;
                if (metric == QK_METRIC_L2) {
                    const float d2 = __uint_as_float(o);
                    od = sqrt_l2 ? sqrtf(d2) : d2;
                } else {
                    od = ip_from_ord(o);
                }
                out_dist[base0 + rank] = od;
            }
        }
    }
    for (int e = min(total, k) + tid; e < k; e += 256) {
        out_ids[base0 + e] = -1;
        if (out_dist) out_dist[base0 + e] = metric == QK_METRIC_IP ? -INFINITY : INFINITY;
    }
}

static int launch_merge_ranks(qk_ctx *ctx, const int64_t *in_ids, const float *in_key, int G, size_t id_stride, size_t key_stride,
                              int64_t Q, int k, int metric, int64_t *out_ids, float *out_dist, bool sqrt_l2) {
    if (Q <= 0) return QK_OK;
    const int Cm = qk_round_up(k + 64, 64);
    if (Cm > 1024) {
        if (G > 64) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_merge_topk: k=%d with %d ranks (the sorted-run merge holds 64 ranks)", k, G);
        hipLaunchKernelGGL(k_merge_ranks_large, dim3((unsigned)Q), dim3(256), 0, ctx->stream, in_ids, in_key, G, id_stride, key_stride, Q, k,
                           metric, sqrt_l2 ? 1 : 0, out_ids, out_dist);
        QK_HIP(hipGetLastError());
        return QK_OK;
    }
    const size_t lds = (size_t)Cm * 12;
    const int mc = Cm <= 128 ? 2 : Cm <= 256 ? 4 : Cm <= 512 ? 8 : 16;
    hipStream_t st = ctx->stream;
    const int sq = sqrt_l2 ? 1 : 0;
    switch (mc) {
        case 2: hipLaunchKernelGGL((k_merge_ranks<2>), dim3((unsigned)Q), dim3(64), lds, st, in_ids, in_key, G, id_stride, key_stride, Q, k, Cm, metric, sq, out_ids, out_dist); break;
        case 4: hipLaunchKernelGGL((k_merge_ranks<4>), dim3((unsigned)Q), dim3(64), lds, st, in_ids, in_key, G, id_stride, key_stride, Q, k, Cm, metric, sq, out_ids, out_dist); break;
        case 8: hipLaunchKernelGGL((k_merge_ranks<8>), dim3((unsigned)Q), dim3(64), lds, st, in_ids, in_key, G, id_stride, key_stride, Q, k, Cm, metric, sq, out_ids, out_dist); break;
        default: hipLaunchKernelGGL((k_merge_ranks<16>), dim3((unsigned)Q), dim3(64), lds, st, in_ids, in_key, G, id_stride, key_stride, Q, k, Cm, metric, sq, out_ids, out_dist); break;
    }
    QK_HIP(hipGetLastError());
    return QK_OK;
}

int qk_merge_topk_device(qk_ctx *ctx, const int64_t *in_ids, const float *in_key, int G, int64_t Q, int k, int metric,
                         int64_t *out_ids, float *out_dist, bool sqrt_l2) {
    return launch_merge_ranks(ctx, in_ids, in_key, G, (size_t)Q * k * 8, (size_t)Q * k * 4, Q, k, metric, out_ids, out_dist, sqrt_l2);
}

// ---- the one-all-to-all exchange of the sharded search (quake_amd/sharded.py) ------------------------------------------------
// A rank's [Q][k] ids + keys become G blocks, block j = the results of queries [j*per, (j+1)*per) as ONE 12-byte-per-entry unit:
// per*k int64 ids followed by per*k float keys, padded to 16 bytes -- the send buffer of a single all_to_all_single; the
// receive buffer ([G] blocks, block r = rank r's results for THIS rank's queries) is merged in place by k_merge_ranks.
size_t qk_topk_block_bytes_(int64_t per, int k) { return (((size_t)per * k * 12) + 15) & ~(size_t)15; }

__global__ __launch_bounds__(256) void k_pack_topk(const int64_t *__restrict__ ids, const float *__restrict__ key, int64_t per, int k,
                                                   size_t blk, int64_t total, unsigned char *__restrict__ packed) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // entry number in [Q][k]
    if (i >= total) return;
    const int64_t pk = per * k;
    const int64_t j = i / pk, e = i - j * pk;
    unsigned char *b = packed + (size_t)j * blk;
    ((int64_t *)b)[e] = ids[i];
    ((float *)(b + (size_t)pk * 8))[e] = key[i];
}

int qk_pack_topk_device(qk_ctx *ctx, const int64_t *ids, const float *key, int G, int64_t per, int k, void *packed) {
    const int64_t total = (int64_t)G * per * k;
    if (total <= 0) return QK_OK;
    hipLaunchKernelGGL(k_pack_topk, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ids, key, per, k,
                       qk_topk_block_bytes_(per, k), total, (unsigned char *)packed);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

int qk_merge_topk_packed_device(qk_ctx *ctx, const void *packed, int G, int64_t per, int k, int metric, int64_t *out_ids,
                                float *out_dist, bool sqrt_l2) {
    const size_t blk = qk_topk_block_bytes_(per, k);
    const unsigned char *b = (const unsigned char *)packed;
    return launch_merge_ranks(ctx, (const int64_t *)b, (const float *)(b + (size_t)per * k * 8), G, blk, blk, per, k, metric, out_ids,
                              out_dist, sqrt_l2);
}


// the merge launch of qk_scan_device: one wave (k <= 32: the flat form; else the chained walk) or one workgroup (k >= 33) per query
int qk_launch_merge(qk_ctx *ctx, MergeParams mp, dim3 mgrid) {
    hipStream_t st = ctx->stream;
    const int k = mp.k, Cm = mp.Cm;
    const int maxch_m = Cm <= 128 ? 2 : Cm <= 256 ? 4 : Cm <= 512 ? 8 : 16;
    const size_t lds_merge = (((size_t)Cm * 12 + 15) & ~(size_t)15) + (size_t)64 * QK_SLOTS * 4;  // pool + 64 pair slot lines
    static const bool merge_clock = qk_env_set("QK_MERGE_CLOCK");
    static long long *d_mclock = nullptr;
    mp.clock = nullptr;
    if (merge_clock && (int64_t)mgrid.x * 64 <= ((int64_t)1 << 24)) {
        if (!d_mclock) QK_HIP(hipMalloc((void **)&d_mclock, (size_t)1 << 24));
        mp.clock = d_mclock;
    }
    static const int merge_wide_min_k = qk_env_int("QK_MERGE_WIDE_MIN_K", 33);
    static const bool merge_flat = qk_env_int("QK_MERGE_FLAT", 1) != 0;
    if (k >= merge_wide_min_k) {
        hipLaunchKernelGGL(k_merge_wide, mgrid, dim3(256), 0, st, mp);
    } else
    if (mp.P <= 64 && k <= 32 && !mp.clock && merge_flat) {
        const size_t lds_flat = lds_merge + (size_t)2 * 64 * (QK_SLOTS - 1) * 4;  // + record list, record sizes
        if (maxch_m == 2) hipLaunchKernelGGL((k_merge_flat<2>), mgrid, dim3(64), lds_flat, st, mp);
        else hipLaunchKernelGGL((k_merge_flat<4>), mgrid, dim3(64), lds_flat, st, mp);
    } else
    switch (maxch_m) {
        case 2: hipLaunchKernelGGL((k_merge<2>), mgrid, dim3(64), lds_merge, st, mp); break;
        case 4: hipLaunchKernelGGL((k_merge<4>), mgrid, dim3(64), lds_merge, st, mp); break;
        case 8: hipLaunchKernelGGL((k_merge<8>), mgrid, dim3(64), lds_merge, st, mp); break;
        default: hipLaunchKernelGGL((k_merge<16>), mgrid, dim3(64), lds_merge, st, mp); break;
    }
    QK_HIP(hipGetLastError());
#ifdef QK_PROBES
    if (mp.clock) {  // debug probe: where a merge wave spends its time (mean / max over the queries, 100 MHz ticks)
        std::vector<long long> h((size_t)mgrid.x * 8);
        QK_HIP(hipMemcpyAsync(h.data(), d_mclock, h.size() * 8, hipMemcpyDeviceToHost, st));
        QK_HIP(hipStreamSynchronize(st));
        const char *names[6] = {"slot line", "fetch", "bound", "consume", "final sort", "output"};
        long long t0 = LLONG_MAX, t1 = 0;
        double mean[6] = {0}, recs = 0;
        long long mx[6] = {0}, mxrec = 0;
        for (size_t i = 0; i < (size_t)mgrid.x; i++) {
            const long long *c = &h[8 * i];
            t0 = std::min(t0, c[0]);
            t1 = std::max(t1, c[6]);
            long long prev = c[0];
            for (int ph = 0; ph < 6; ph++) {
                const long long cur = c[ph + 1] ? c[ph + 1] : prev;
                mean[ph] += (double)(cur - prev);
                mx[ph] = std::max(mx[ph], cur - prev);
                prev = cur;
            }
            recs += (double)c[7];
            mxrec = std::max(mxrec, c[7]);
        }
        fprintf(stderr, "[k_merge waves] n=%u span=%lld ticks, records per query mean=%.1f max=%lld;", mgrid.x, t1 - t0, recs / mgrid.x, mxrec);
        for (int ph = 0; ph < 6; ph++) fprintf(stderr, " %s mean=%.0f max=%lld;", names[ph], mean[ph] / mgrid.x, mx[ph]);
        fprintf(stderr, "\n");
    }
#endif
    return QK_OK;
}

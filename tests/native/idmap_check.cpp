// QkIdMap (quake_amd/csrc/qk_idmap.h) against std::unordered_map on a random stream of set / set_if_absent / erase / erase_if /
// take / prefetch / find / clear / reserve.  Built and run by tests/test_idmap_host.py (g++, no GPU).
#include "qk_idmap.h"

#include <cstdio>
#include <random>
#include <unordered_map>

int main() {
    std::mt19937_64 rng(7);
    QkIdMap m;
    std::unordered_map<int64_t, int32_t> r;
    for (int it = 0; it < 3000000; it++) {
        const int64_t key = (int64_t)(rng() % 200000) - 1000;  // negative ids too
        const int32_t val = (int32_t)(rng() % 5000);
        switch (rng() % 7) {
            case 0:
                m.set(key, val);
                r[key] = val;
                break;
            case 1:
                m.set_if_absent(key, val);
                r.emplace(key, val);
                break;
            case 2:
                m.erase(key);
                r.erase(key);
                break;
            case 3: {
                m.erase_if(key, val);
                auto f = r.find(key);
                if (f != r.end() && f->second == val) r.erase(f);
                break;
            }
            case 4: {  // take = find + erase (prefetch first, as the store does)
                m.prefetch(key);
                auto f = r.find(key);
                const int32_t want = f == r.end() ? -1 : f->second;
                if (m.take(key) != want) {
                    printf("TAKE MISMATCH at step %d\n", it);
                    return 1;
                }
                if (f != r.end()) r.erase(f);
                break;
            }
            default: {
                auto f = r.find(key);
                const int32_t want = f == r.end() ? -1 : f->second;
                if (m.find(key) != want) {
                    printf("MISMATCH at step %d\n", it);
                    return 1;
                }
            }
        }
        if (it % 250000 == 0 && m.size() != r.size()) {
            printf("SIZE MISMATCH at step %d\n", it);
            return 1;
        }
        if (it == 1500000) {
            m.clear();
            r.clear();
        }
        if (it == 1600000) m.reserve(300000);
    }
    for (auto &kv : r)
        if (m.find(kv.first) != kv.second) {
            printf("FINAL MISMATCH\n");
            return 1;
        }
    // erased slots are recycled: a long insert / erase churn on a small live set must not grow the table without bound
    QkIdMap c;
    for (int64_t i = 0; i < 2000000; i++) {
        c.set(i, 1);
        if (i >= 100) c.erase(i - 100);
    }
    if (c.size() != 100 || c.slots.size() > 4096) {
        printf("CHURN: size %zu slots %zu\n", c.size(), c.slots.size());
        return 1;
    }
    // bulk build from segments (the store's index after a bulk load): first segment wins for a key held twice; the built table
    // then takes the single-key operations like any other
    for (int threads : {1, 3, 4, 16}) {
        std::mt19937_64 g2(100 + threads);
        const size_t nseg = 700;
        std::vector<std::vector<int64_t>> segs(nseg);
        std::vector<const int64_t *> keys(nseg);
        std::vector<int64_t> lens(nseg);
        std::vector<int32_t> vals(nseg);
        std::unordered_map<int64_t, int32_t> want;
        for (size_t j = 0; j < nseg; j++) {
            const size_t n = j % 50 == 7 ? 0 : 1000 + (size_t)(g2() % 3000);
            for (size_t i = 0; i < n; i++) {
                // mostly fresh ids, some repeats of earlier ones (within and across segments), negative ids too
                const int64_t key = (g2() % 16 == 0) ? (int64_t)(g2() % 5000) - 2500 : (int64_t)(j * 1000003 + i * 7919 + 10000);
                segs[j].push_back(key);
                want.emplace(key, (int32_t)(j + 5));
            }
            keys[j] = segs[j].data();
            lens[j] = (int64_t)segs[j].size();
            vals[j] = (int32_t)(j + 5);
        }
        QkIdMap b;
        b.set(12345, 1);  // (a build starts from scratch)
        b.build_from_segments(keys.data(), lens.data(), vals.data(), nseg, threads);
        if (b.size() != want.size()) {
            printf("BULK SIZE MISMATCH threads %d: %zu vs %zu\n", threads, b.size(), want.size());
            return 1;
        }
        for (auto &kv : want)
            if (b.find(kv.first) != kv.second) {
                printf("BULK MISMATCH threads %d\n", threads);
                return 1;
            }
        if (b.find(-999999) != -1 || b.used != b.live) {
            printf("BULK STATE threads %d\n", threads);
            return 1;
        }
        for (int it = 0; it < 200000; it++) {  // and goes on as a map
            const int64_t key = (int64_t)(g2() % 40000) - 3000;
            if (g2() & 1) {
                b.set(key, 3);
                want[key] = 3;
            } else {
                auto f = want.find(key);
                const int32_t w = f == want.end() ? -1 : f->second;
                if (b.take(key) != w) {
                    printf("BULK TAKE MISMATCH threads %d\n", threads);
                    return 1;
                }
                if (f != want.end()) want.erase(f);
            }
        }
        if (b.size() != want.size()) {
            printf("BULK CHURN SIZE threads %d\n", threads);
            return 1;
        }
    }
    printf("ok\n");
    return 0;
}

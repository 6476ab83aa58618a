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
    printf("ok\n");
    return 0;
}

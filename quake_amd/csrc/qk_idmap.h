// Host-side id -> list number map of the store (qk_store.hip).  Plain C++, no HIP: tests/test_idmap_host.py compiles it with g++
// and checks it against std::unordered_map.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <thread>
#include <utility>
#include <vector>

// fn(0) ... fn(T - 1) on T threads: T - 1 new ones and the caller's own; shares whose thread could not be started (resource limits)
// run on the caller's thread too, so the work is always done
template <class F>
static inline void qk_run_shares(int T, F &&fn) {
    std::vector<std::thread> th;
    th.reserve((size_t)std::max(0, T - 1));
    int started = 0;
    for (; started < T - 1; started++) {
        try {
            th.emplace_back(fn, started);
        } catch (...) {
            break;
        }
    }
    for (int t = started; t < T; t++) fn(t);
    for (auto &x : th) x.join();
}

// id -> list number: open addressing, linear probing, power-of-two capacity, 16-byte slots, load <= 0.5 (live + erased).  The store
// inserts / erases hundreds of thousands of ids per add / remove call: a node-based std::unordered_map spent more time in
// its allocator than the device spent on the rows.
struct QkIdMap {
    static constexpr int32_t EMPTY = -1, ERASED = -2;  // list numbers are >= 0
    struct Slot {  // key and value side by side: one cache line per probe
        int64_t key;
        int32_t val;
        int32_t pad;
    };
    // the slot array: plain storage whose first touch (page faults of a table of gigabytes) can be shared out over threads
    struct SlotArray {
        Slot *p = nullptr;
        size_t n = 0;
        SlotArray() = default;
        SlotArray(const SlotArray &) = delete;
        SlotArray &operator=(const SlotArray &) = delete;
        ~SlotArray() { free(p); }
        size_t size() const { return n; }
        bool empty() const { return n == 0; }
        Slot *data() { return p; }
        Slot &operator[](size_t i) { return p[i]; }
        const Slot &operator[](size_t i) const { return p[i]; }
        void clear() {
            free(p);
            p = nullptr;
            n = 0;
        }
        void swap(SlotArray &o) {
            std::swap(p, o.p);
            std::swap(n, o.n);
        }
        void assign_empty(size_t cap, int threads) {  // cap slots, all EMPTY
            clear();
            p = (Slot *)malloc(cap * sizeof(Slot));
            if (!p) throw std::bad_alloc();
            n = cap;
            const int T = (int)std::min<size_t>((size_t)std::max(1, threads), std::max<size_t>(1, cap >> 20));
            auto fill = [this](size_t a, size_t b) {
                for (size_t i = a; i < b; i++) p[i] = Slot{0, EMPTY, 0};
            };
            if (T <= 1) {
                fill(0, cap);
                return;
            }
            qk_run_shares(T, [&](int t) { fill(cap * (size_t)t / (size_t)T, cap * (size_t)(t + 1) / (size_t)T); });
        }
    };
    SlotArray slots;
    size_t live = 0, used = 0;  // used = live + erased
    static inline uint64_t mix(uint64_t x) {
        x ^= x >> 30;
        x *= 0xbf58476d1ce4e5b9ull;
        x ^= x >> 27;
        x *= 0x94d049bb133111ebull;
        x ^= x >> 31;
        return x;
    }
    size_t size() const { return live; }
    void clear() {
        slots.clear();
        live = used = 0;
    }
    void rehash(size_t want_live, int threads = 1) {
        size_t cap = 64;
        while (cap < want_live * 2 + 16) cap <<= 1;
        SlotArray old;
        old.swap(slots);
        slots.assign_empty(cap, threads);
        live = used = 0;
        for (size_t i = 0; i < old.size(); i++)
            if (old[i].val >= 0) set(old[i].key, old[i].val);
    }
    void reserve(size_t n, int threads = 1) {
        if (slots.size() < n * 2 + 16) rehash(n, threads);
    }
    // slot of `key`, or of the first free slot of its probe sequence (erased slots are reused)
    inline size_t probe(int64_t key, bool &found) const {
        const size_t mask = slots.size() - 1;
        size_t i = (size_t)mix((uint64_t)key) & mask, first_free = (size_t)-1;
        for (;; i = (i + 1) & mask) {
            const int32_t v = slots[i].val;
            if (v == EMPTY) {
                found = false;
                return first_free != (size_t)-1 ? first_free : i;
            }
            if (v == ERASED) {
                if (first_free == (size_t)-1) first_free = i;
            } else if (slots[i].key == key) {
                found = true;
                return i;
            }
        }
    }
    // the cache line a later find / set / erase of `key` starts at: the store walks hundreds of thousands of random ids per call, one
    // DRAM round trip each when the table is larger than the caches -- requested a few iterations ahead they overlap
    inline void prefetch(int64_t key) const {
        if (!slots.empty()) __builtin_prefetch(&slots[(size_t)mix((uint64_t)key) & (slots.size() - 1)]);
    }
    int32_t find(int64_t key) const {  // list number or -1
        if (slots.empty()) return -1;
        bool f;
        const size_t i = probe(key, f);
        return f ? slots[i].val : -1;
    }
    // the value of a key that is PRESENT is replaced in place (returns false, and does nothing, when it is absent): no slot changes its
    // state, so calls for different keys may run on different threads at the same time
    bool overwrite_present(int64_t key, int32_t val) {
        if (slots.empty()) return false;
        bool f;
        const size_t i = probe(key, f);
        if (f) slots[i].val = val;
        return f;
    }
    void put(int64_t key, int32_t val, bool overwrite) {
        if ((used + 1) * 2 > slots.size()) rehash(std::max<size_t>(live + 1, live * 2));
        bool f;
        const size_t i = probe(key, f);
        if (f) {
            if (overwrite) slots[i].val = val;
            return;
        }
        if (slots[i].val == EMPTY) used++;
        slots[i].key = key;
        slots[i].val = val;
        live++;
    }
    // Bulk build from segments (every key of segment j maps to vals[j]; a key met again keeps its FIRST segment): the table of
    // a 50M-id store took 1.8 s to fill from one thread -- the first remove / get_vector after a bulk build paid it.  The table
    // is cut into `threads` slot ranges; a first sweep (segments shared out) notes every key's owner = the range its probe
    // sequence starts in; then every thread walks the owner bytes in segment order and inserts its own keys, which keeps the
    // first-segment rule per key without any locking.  A probe that would leave its range is put aside and inserted at the end.
    void build_from_segments(const int64_t *const *keys, const int64_t *lens, const int32_t *vals, size_t nseg, int threads) {
        size_t total = 0;
        std::vector<size_t> seg_off(nseg + 1, 0);
        for (size_t j = 0; j < nseg; j++) {
            total += (size_t)std::max<int64_t>(0, lens[j]);
            seg_off[j + 1] = total;
        }
        clear();
        reserve(total + 16, threads);
        if (slots.empty()) rehash(16);
        const size_t cap = slots.size(), mask = cap - 1;
        const int T = (int)std::min<size_t>((size_t)std::max(1, std::min(threads, 255)), std::max<size_t>(1, total >> 16));
        if (T <= 1) {
            for (size_t j = 0; j < nseg; j++)
                for (int64_t i = 0; i < lens[j]; i++) set_if_absent(keys[j][i], vals[j]);
            return;
        }
        // range t = slots [cap * t / T, cap * (t + 1) / T)
        auto owner_of = [&](size_t h) { return (uint8_t)(((unsigned __int128)h * (unsigned)T) / cap); };
        std::vector<uint8_t> owner(total);
        qk_run_shares(T, [&](int t) {
            for (size_t j = (size_t)t; j < nseg; j += (size_t)T) {
                uint8_t *o = owner.data() + seg_off[j];
                const int64_t *kj = keys[j];
                for (int64_t i = 0; i < lens[j]; i++) o[i] = owner_of((size_t)mix((uint64_t)kj[i]) & mask);
            }
        });
        std::vector<std::vector<std::pair<int64_t, int32_t>>> aside((size_t)T);
        std::vector<size_t> made((size_t)T, 0);
        {
            qk_run_shares(T, [&](int t) {
                    const size_t hi = (size_t)(((unsigned __int128)cap * (unsigned)(t + 1)) / (unsigned)T);
                    Slot *sl = slots.data();
                    size_t n_made = 0;
                    constexpr int AHEAD = 16;
                    struct Pend { int64_t key; size_t h; int32_t val; };
                    Pend ring[AHEAD];
                    int head = 0, fill = 0;
                    auto insert = [&](const Pend &pe) {
                        size_t i = pe.h;
                        for (; i < hi; i++) {
                            if (sl[i].val == EMPTY) {
                                sl[i].key = pe.key;
                                sl[i].val = pe.val;
                                n_made++;
                                return;
                            }
                            if (sl[i].key == pe.key) return;  // met before: the first segment stays
                        }
                        aside[(size_t)t].emplace_back(pe.key, pe.val);
                    };
                    for (size_t j = 0; j < nseg; j++) {
                        const uint8_t *o = owner.data() + seg_off[j];
                        const int64_t *kj = keys[j];
                        for (int64_t i = 0; i < lens[j]; i++) {
                            if (o[i] != (uint8_t)t) continue;
                            Pend pe{kj[i], (size_t)mix((uint64_t)kj[i]) & mask, vals[j]};
                            __builtin_prefetch(&sl[pe.h], 1);
                            if (fill == AHEAD) {  // the slot asked for AHEAD keys ago is (likely) here by now
                                insert(ring[head]);
                                ring[head] = pe;
                                head = (head + 1) % AHEAD;
                            } else {
                                ring[(head + fill) % AHEAD] = pe;
                                fill++;
                            }
                        }
                    }
                    for (int r = 0; r < fill; r++) insert(ring[(head + r) % AHEAD]);
                    made[(size_t)t] = n_made;
                });
        }
        for (int t = 0; t < T; t++) {
            live += made[(size_t)t];
            used += made[(size_t)t];
        }
        for (int t = 0; t < T; t++)
            for (auto &kv : aside[(size_t)t]) set_if_absent(kv.first, kv.second);
    }
    void set(int64_t key, int32_t val) { put(key, val, true); }
    void set_if_absent(int64_t key, int32_t val) { put(key, val, false); }
    void erase(int64_t key) {
        if (slots.empty()) return;
        bool f;
        const size_t i = probe(key, f);
        if (f) {
            slots[i].val = ERASED;
            live--;
        }
    }
    int32_t take(int64_t key) {  // find + erase in one probe: the value the key had, or -1
        if (slots.empty()) return -1;
        bool f;
        const size_t i = probe(key, f);
        if (!f) return -1;
        const int32_t v = slots[i].val;
        slots[i].val = ERASED;
        live--;
        return v;
    }
    void erase_if(int64_t key, int32_t val) {  // only while it still maps to `val`
        if (slots.empty()) return;
        bool f;
        const size_t i = probe(key, f);
        if (f && slots[i].val == val) {
            slots[i].val = ERASED;
            live--;
        }
    }
};


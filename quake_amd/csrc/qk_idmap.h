// Host-side id -> list number map of the store (qk_store.hip).  Plain C++, no HIP: tests/test_idmap_host.py compiles it with g++
// and checks it against std::unordered_map.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

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
    std::vector<Slot> slots;
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
    void rehash(size_t want_live) {
        size_t cap = 64;
        while (cap < want_live * 2 + 16) cap <<= 1;
        std::vector<Slot> old;
        old.swap(slots);
        slots.assign(cap, Slot{0, EMPTY, 0});
        live = used = 0;
        for (const Slot &o : old)
            if (o.val >= 0) set(o.key, o.val);
    }
    void reserve(size_t n) {
        if (slots.size() < n * 2 + 16) rehash(n);
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


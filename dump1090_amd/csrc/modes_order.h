// Host side of the record hand-off: the demod kernel leaves its records in the slots it reserved, in
// completion order and with unused slots in between (block == invalid); modes_host_resolve needs them in
// stream order, ascending (buffer, offset).  Header-only so that the CPU tests can drive it
// (tests/native/order_shim.cpp).
//
// One call covers at most 2^15 buffers, so (buffer - first_buffer, offset) is a 32-bit key; key and slot
// index travel as one u64.  Small lists: one std::sort and a gather.  Large lists (a message-dense
// capture leaves ~750,000 records = 49 MB per GiB; a gather in sorted order would miss the cache on
// every record): T threads partition the RECORDS by the high bits of their keys (count, prefix,
// scatter - three streaming passes), then sort bucket by bucket, each bucket in a core's L2.
#ifndef MODES_ORDER_H
#define MODES_ORDER_H
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/modes_gfx950.h"

struct modes_order_scratch {
    std::vector<uint64_t> keys;
    std::vector<uint32_t> hist;                 // [threads][buckets]
    std::vector<size_t> bucket_off, piece;
    std::vector<modes_record> bucketed;
};

// slots[0..nslots): records and invalid slots; first_block: the smallest buffer index of the call.
// Writes the valid records in ascending (block, j) to out (capacity >= number of valid records) and
// returns their number.  `threads` <= 1, or a short list, takes the single-thread path.
inline size_t modes_order_records(const modes_record *slots, size_t nslots, uint32_t invalid_block, uint32_t first_block,
                                  modes_record *out, modes_order_scratch &sc, int threads) {
    constexpr size_t kParallelFrom = 1u << 15;
    auto key32 = [&](const modes_record &r) -> uint32_t { return ((r.block - first_block) << 17) | r.j; };
    if (threads <= 1 || nslots < kParallelFrom) {
        std::vector<uint64_t> &k = sc.keys;
        k.clear();
        for (size_t i = 0; i < nslots; i++)
            if (slots[i].block != invalid_block) k.push_back(((uint64_t)key32(slots[i]) << 32) | (uint64_t)i);
        std::sort(k.begin(), k.end());
        for (size_t i = 0; i < k.size(); i++) out[i] = slots[(uint32_t)k[i]];
        return k.size();
    }
    const int T = std::min(threads, 64);
    constexpr int kBucketBits = 10, kBuckets = 1 << kBucketBits;
    // buckets are key ranges (concatenating the sorted buckets gives the sorted list): high bits of the
    // largest key that can occur
    uint32_t max_block = 0;
    for (size_t i = 0; i < nslots; i++)
        if (slots[i].block != invalid_block) max_block = std::max(max_block, slots[i].block - first_block);
    int shift = 0;
    while (shift < 32 && ((((uint64_t)max_block + 1) << 17) >> shift) > (uint64_t)kBuckets) shift++;
    auto bucket_of = [&](uint32_t key) -> uint32_t { return key >> shift; };

    sc.hist.assign((size_t)T * kBuckets, 0u);
    auto slice = [&](int t, size_t *lo, size_t *hi) { *lo = nslots * (size_t)t / T; *hi = nslots * (size_t)(t + 1) / T; };
    auto run = [&](auto &&fn) {
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(fn, t);
        fn(0);
        for (auto &x : th) x.join();
    };
    run([&](int t) {                                                         // 1. histogram of this slice
        size_t lo, hi;
        slice(t, &lo, &hi);
        uint32_t *h = &sc.hist[(size_t)t * kBuckets];
        for (size_t i = lo; i < hi; i++)
            if (slots[i].block != invalid_block) h[bucket_of(key32(slots[i]))]++;
    });
    sc.bucket_off.assign(kBuckets + 1, 0);                                   // 2. where every (bucket, thread) piece goes
    sc.piece.resize((size_t)T * kBuckets);
    size_t total = 0;
    for (int b = 0; b < kBuckets; b++) {
        sc.bucket_off[b] = total;
        for (int t = 0; t < T; t++) { sc.piece[(size_t)t * kBuckets + b] = total; total += sc.hist[(size_t)t * kBuckets + b]; }
    }
    sc.bucket_off[kBuckets] = total;
    if (sc.bucketed.size() < total) sc.bucketed.resize(total);
    run([&](int t) {                                                         // 3. scatter the records
        size_t lo, hi;
        slice(t, &lo, &hi);
        size_t *dst = &sc.piece[(size_t)t * kBuckets];
        for (size_t i = lo; i < hi; i++)
            if (slots[i].block != invalid_block) sc.bucketed[dst[bucket_of(key32(slots[i]))]++] = slots[i];
    });
    run([&](int t) {                                                         // 4. sort bucket by bucket (dealt round-robin)
        std::vector<uint64_t> k;
        for (int b = t; b < kBuckets; b += T) {
            const size_t off = sc.bucket_off[b], n = sc.bucket_off[b + 1] - off;
            const modes_record *src = sc.bucketed.data() + off;
            k.resize(n);
            for (size_t i = 0; i < n; i++) k[i] = ((uint64_t)key32(src[i]) << 32) | (uint64_t)i;
            std::sort(k.begin(), k.end());
            for (size_t i = 0; i < n; i++) out[off + i] = src[(uint32_t)k[i]];
        }
    });
    return total;
}
#endif

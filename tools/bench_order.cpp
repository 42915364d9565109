// Host-only timing of the record ordering (dump1090_amd/csrc/modes_order.h) on a message-dense list:
//   g++ -O2 -std=c++17 -pthread -Iinclude tools/bench_order.cpp -o /tmp/bench_order && /tmp/bench_order
#include "../dump1090_amd/csrc/modes_order.h"
#include <chrono>
#include <cstdio>
#include <random>
int main() {
    const size_t n = 780000;
    std::vector<modes_record> slots(n), out(n);
    std::mt19937_64 rng(1);
    for (size_t i = 0; i < n; i++) {
        memset(&slots[i], 0, sizeof slots[i]);
        if (i % 40 == 39) { slots[i].block = 0xFFFFFFFFu; continue; }
        uint64_t pos = rng() % (4096ull * 131070);
        slots[i].block = (uint32_t)(pos / 131072); slots[i].j = (uint32_t)(pos % 131072);
    }
    modes_order_scratch sc;
    for (int threads : {1, 1, 2, 4, 8, 16, 32, 8}) {
        auto a = std::chrono::steady_clock::now();
        size_t m = modes_order_records(slots.data(), n, 0xFFFFFFFFu, 0, out.data(), sc, threads);
        auto b = std::chrono::steady_clock::now();
        printf("threads %d: %zu records, %.2f ms\n", threads, m, std::chrono::duration<double, std::milli>(b - a).count());
    }
}

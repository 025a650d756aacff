#include <algorithm>
#include <cstdint>
#include <random>
#include <set>
#include <vector>
// variant fields: order(0 rows first,1 costs first) | dup(0 elem,1 row) <<1 | draw(0 dist size_t,1 modulo,2 dist via double) <<2 | costrng(0 same,1 fresh same seed,2 seed+1,3 default mt19937(seed) 32-bit) <<4 | rowrng32 <<6
extern "C" int gen(uint64_t V, uint64_t B, uint64_t k, uint64_t seed, int variant, uint64_t* rows, double* costs)
{
    const int order = variant & 1, dup = (variant >> 1) & 1, drawm = (variant >> 2) & 3, crng = (variant >> 4) & 3, r32 = (variant >> 6) & 1;
    std::mt19937_64 rng(seed);
    std::mt19937 rng32((uint32_t)seed);
    std::mt19937_64 rc_same(seed), rc_p1(seed + 1);
    std::mt19937 rc32((uint32_t)seed);
    auto draw_costs = [&]() {
        std::uniform_real_distribution<double> d(1.0, 10.0);
        for (uint64_t v = 0; v < V; ++v) {
            switch (crng) { case 0: costs[v] = r32 ? d(rng32) : d(rng); break; case 1: costs[v] = d(rc_same); break; case 2: costs[v] = d(rc_p1); break; default: costs[v] = d(rc32); }
        }
    };
    if (order) draw_costs();
    std::uniform_int_distribution<size_t> ds(0, V - 1);
    std::uniform_real_distribution<double> du(0.0, 1.0);
    auto draw = [&]() -> uint64_t {
        if (r32) { if (drawm == 1) return rng32() % V; return ds(rng32); }
        if (drawm == 1) return rng() % V;
        if (drawm == 2) return (uint64_t)(du(rng) * V);
        return ds(rng);
    };
    std::vector<uint64_t> r;
    for (uint64_t b = 0; b < B; ++b) {
        r.clear();
        if (dup) {
            for (;;) { r.clear(); for (uint64_t i = 0; i < k; ++i) r.push_back(draw()); std::sort(r.begin(), r.end()); if (std::adjacent_find(r.begin(), r.end()) == r.end()) break; }
        } else {
            while (r.size() < k) { const uint64_t v = draw(); if (std::find(r.begin(), r.end(), v) == r.end()) r.push_back(v); }
            std::sort(r.begin(), r.end());
        }
        for (uint64_t i = 0; i < k; ++i) rows[b * k + i] = r[i];
    }
    if (!order) draw_costs();
    return 0;
}

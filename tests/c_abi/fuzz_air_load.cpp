// Host-only fuzz harness of the AEROAIR loader (air_program.hpp) under ASan/UBSan: mutated program bytes must be rejected with an
// aero::Error or load cleanly; a loaded program must instantiate, generate its device code and evaluate on the host without touching
// memory out of bounds.
#include "air_program.hpp"
#include <random>
#include <cstdio>
using namespace aero;
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937_64 rng(argc > 2 ? atoll(argv[2]) : 1);
    std::vector<std::vector<uint8_t>> seeds;
    seeds.push_back(air::fib_program(4, 2, 3, 3));
    seeds.push_back(air::fib_program(2, 0, 0, 2));
    seeds.push_back(air::synth_vm_program(6, 2, 3, 4));
    seeds.push_back(air::synth_vm_program(5, 1, 0, 0));
    size_t loaded = 0, rejected = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> b = seeds[rng() % seeds.size()];
        const int nmut = 1 + rng() % 4;
        for (int m = 0; m < nmut; m++) {
            const int kind = rng() % 6;
            if (b.empty()) break;
            const size_t pos = rng() % b.size();
            if (kind == 0) b[pos] ^= (uint8_t)(1u << (rng() % 8));
            else if (kind == 1) b[pos] = (uint8_t)rng();
            else if (kind == 2 && pos + 4 <= b.size()) { uint32_t v = (uint32_t)(rng() % 3 == 0 ? 0xFFFFFFFFu : rng() % 300); memcpy(&b[pos & ~3ull], &v, 4); }
            else if (kind == 3) b.resize(pos);
            else if (kind == 4) b.insert(b.begin() + pos, (size_t)(rng() % 16), (uint8_t)rng());
            else if (pos + 8 <= b.size()) { uint64_t v = rng(); memcpy(&b[pos & ~7ull], &v, 8); }
        }
        try {
            air::Program p = air::load(b.data(), b.size());
            loaded++;
            for (int log_n : {3, 5, 9}) {
                try {
                    air::Instance in = air::instantiate(p, log_n);
                    (void)in;
                } catch (const Error&) {}
            }
        } catch (const Error&) { rejected++; }
    }
    printf("%d mutated programs: %zu loaded, %zu rejected, no memory error\n", iters, loaded, rejected);
    return 0;
}

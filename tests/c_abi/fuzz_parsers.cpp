// Host-only fuzz harness (g++ -fsanitize=address,undefined) of the byte parsers that face a host: the SDK's worker messages
// (worker_messages.hpp: bincode HashingWorkItem / ConstraintComputeWorkItem), the proof layout and the Miden public inputs
// (proof_format.hpp). argv: hashing_item constraint_item proof miden_inputs iterations seed. Mutated inputs must end in an
// aero::Error (FormatError) or parse cleanly - never in a memory error.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "proof_format.hpp"
#include "worker_messages.hpp"
using namespace aero;
static std::vector<uint8_t> read_file(const char* p) {
    FILE* f = fopen(p, "rb"); if (!f) { printf("cannot open %s\n", p); exit(2); }
    std::vector<uint8_t> b; uint8_t buf[4096]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f); return b;
}
int main(int argc, char** argv) {
    if (argc < 7) { printf("usage: fuzz_parsers hashing_item constraint_item proof miden_inputs iters seed\n"); return 64; }
    const std::vector<uint8_t> seeds[4] = {read_file(argv[1]), read_file(argv[2]), read_file(argv[3]), read_file(argv[4])};
    const int iters = atoi(argv[5]);
    std::mt19937_64 rng(atoll(argv[6]));
    size_t ok[4] = {0, 0, 0, 0}, bad[4] = {0, 0, 0, 0};
    for (int it = -4; it < iters; it++) {
        const int which = it < 0 ? it + 4 : (int)(rng() % 4);
        std::vector<uint8_t> b = seeds[which];
        const int nmut = it < 0 ? 0 : 1 + (int)(rng() % 3);
        for (int m = 0; m < nmut && !b.empty(); m++) {
            const int kind = rng() % 6;
            const size_t pos = rng() % b.size();
            if (kind == 0) b[pos] ^= (uint8_t)(1u << (rng() % 8));
            else if (kind == 1) b[pos] = (uint8_t)rng();
            else if (kind == 2) b.resize(pos);
            else if (kind == 3) b.insert(b.begin() + pos, (size_t)(rng() % 24), (uint8_t)rng());
            else if (pos + 8 <= b.size()) { uint64_t v = rng() % 4 == 0 ? ~0ull : rng() % 4 == 1 ? (rng() % 4096) : rng(); memcpy(&b[pos & ~7ull], &v, 8); }
        }
        try {
            if (which == 0) { std::vector<uint64_t> offs; (void)wm::scan_hashing_work_item(b.data(), b.size(), offs); }
            else if (which == 1) { const wm::ConstraintWorkItem w = wm::parse_constraint_work_item(b.data(), b.size()); (void)w; }
            else if (which == 2) {
                const fmt::Parsed p = fmt::parse(b.data(), b.size());
                (void)p;
            } else { const fmt::MidenInputs mi = fmt::parse_miden_inputs(b.data(), b.size()); (void)mi; }
            ok[which]++;
        } catch (const Error&) { bad[which]++; }
        if (it < 0 && bad[which]) { printf("seed %d does not parse\n", which); return 1; }
    }
    printf("%d mutated inputs: hashing %zu/%zu, constraint item %zu/%zu, proof %zu/%zu, public inputs %zu/%zu parsed/rejected, no memory error\n", iters,
           ok[0], bad[0], ok[1], bad[1], ok[2], bad[2], ok[3], bad[3]);
    return 0;
}

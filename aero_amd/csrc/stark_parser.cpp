// bin/stark_parser: the command line of the reference's `miden-to-cairo-parser` (src/main.rs:12-113) over libaero_stark.so, so that
// the Cairo side's hint glue (src/stark_verifier/utils.py:4-6,33-41, tests/integration/utils.py:5-24: `subprocess.run([
// 'bin/stark_parser', path, command, ...])`) works unchanged on proofs this backend emits:
//
//   stark_parser <container.bin> proof
//   stark_parser <container.bin> public-inputs
//   stark_parser <container.bin> trace-queries '[i0, i1, ...]'        (constraint-queries / fri-queries alike)
//   stark_parser <container.bin> interpolate-poly '["<8 bytes LE hex>", ...]' '[...]'
//
// The first five print the Cairo-memory JSON array (aero_cairo_memory) followed by the newline `println!` adds; interpolate-poly
// prints the coefficients of the interpolant the way main.rs:103-109 folds them (", c0, c1, ..."). Host code only, no GPU.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/aero_stark.h"

namespace {
typedef unsigned __int128 u128;
const uint64_t P = 0xFFFFFFFF00000001ull;
uint64_t addm(uint64_t a, uint64_t b) { const u128 s = (u128)a + b; return (uint64_t)(s >= P ? s - P : s); }
uint64_t subm(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (P - b); }
uint64_t mulm(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % P); }
uint64_t powm(uint64_t a, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = mulm(r, a); a = mulm(a, a); e >>= 1; } return r; }
uint64_t invm(uint64_t a) { return powm(a, P - 2); }

[[noreturn]] void die(const std::string& why) { fprintf(stderr, "stark_parser: %s\n", why.c_str()); exit(1); }

// '[1, 2, 3]' -> numbers; '["0100000000000000", ...]' -> strings
std::vector<std::string> json_items(const std::string& s) {
    std::vector<std::string> out;
    size_t i = s.find('[');
    if (i == std::string::npos) die("expected a JSON array");
    std::string cur;
    bool in_str = false, any = false;
    for (i++; i < s.size(); i++) {
        const char c = s[i];
        if (in_str) { if (c == '"') in_str = false; else cur += c; continue; }
        if (c == '"') { in_str = true; any = true; continue; }
        if (c == ',' || c == ']') { if (any) out.push_back(cur); cur.clear(); any = false; if (c == ']') return out; continue; }
        if (c == ' ' || c == '\t' || c == '\n') continue;
        cur += c; any = true;
    }
    die("unterminated JSON array");
}
std::vector<uint64_t> felt_array(const std::string& s) {        // decode_felt_array (main.rs:115-126): 8 bytes, little endian, hex
    std::vector<uint64_t> v;
    for (const std::string& it : json_items(s)) {
        if (it.size() != 16) die("a field element is 16 hex digits (8 bytes, little endian)");
        uint64_t x = 0;
        for (int b = 0; b < 8; b++) x |= (uint64_t)strtoul(it.substr(2 * b, 2).c_str(), nullptr, 16) << (8 * b);
        v.push_back(x % P);
    }
    return v;
}
// winter_math::polynom::interpolate(xs, ys, false): coefficients (low to high) of the degree < n polynomial through the points
std::vector<uint64_t> interpolate(const std::vector<uint64_t>& xs, const std::vector<uint64_t>& ys) {
    const size_t n = xs.size();
    if (ys.size() != n || n == 0) die("interpolate-poly: x and y value counts differ");
    std::vector<uint64_t> root(n + 1, 0);                          // prod (x - x_i)
    root[0] = 1;
    for (size_t i = 0; i < n; i++) {
        for (size_t k = i + 1; k-- > 0;) { root[k + 1] = addm(root[k + 1], root[k]); root[k] = mulm(root[k], subm(0, xs[i])); }
    }
    std::vector<uint64_t> res(n, 0), q(n);
    for (size_t i = 0; i < n; i++) {
        // q = root / (x - x_i) by synthetic division, then scale by y_i / q(x_i)
        uint64_t carry = 0;
        for (size_t k = n; k-- > 0;) { carry = addm(root[k + 1], mulm(carry, xs[i])); q[k] = carry; }
        uint64_t denom = 0;
        for (size_t k = n; k-- > 0;) denom = addm(mulm(denom, xs[i]), q[k]);
        if (denom == 0) die("interpolate-poly: repeated x value");
        const uint64_t c = mulm(ys[i], invm(denom));
        for (size_t k = 0; k < n; k++) res[k] = addm(res[k], mulm(q[k], c));
    }
    return res;
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) die("usage: stark_parser <container.bin> proof|public-inputs|trace-queries|constraint-queries|fri-queries [indexes] | interpolate-poly <x> <y>");
    const std::string cmd = argv[2];
    if (cmd == "interpolate-poly") {
        if (argc < 5) die("interpolate-poly needs x_values and y_values");
        std::string line;
        for (uint64_t c : interpolate(felt_array(argv[3]), felt_array(argv[4]))) line += ", " + std::to_string(c);
        printf("%s\n", line.c_str());
        return 0;
    }
    // BinaryProofData::from_file (lib.rs:25-39): bincode ProofData { input_bytes, proof_bytes }
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) die(std::string("cannot open ") + argv[1]);
    const std::vector<uint8_t> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto u64 = [&](size_t o) { if (blob.size() < o + 8) die("truncated container"); uint64_t v; memcpy(&v, blob.data() + o, 8); return v; };
    const uint64_t n = u64(0);
    if (n > blob.size() - 8) die("truncated container");
    const uint64_t m = u64(8 + n);
    if (m != blob.size() - 16 - n) die("container length mismatch");
    const uint8_t *inputs = blob.data() + 8, *proof = blob.data() + 16 + n;
    uint32_t what;
    if (cmd == "proof") what = AERO_CAIRO_PROOF;
    else if (cmd == "public-inputs") what = AERO_CAIRO_PUBLIC_INPUTS;
    else if (cmd == "trace-queries") what = AERO_CAIRO_TRACE_QUERIES;
    else if (cmd == "constraint-queries") what = AERO_CAIRO_CONSTRAINT_QUERIES;
    else if (cmd == "fri-queries") what = AERO_CAIRO_FRI_QUERIES;
    else die("unknown command " + cmd);
    std::vector<uint64_t> indexes;
    if (what >= AERO_CAIRO_TRACE_QUERIES) {
        if (argc < 4) die(cmd + " needs the JSON array of query positions");
        for (const std::string& it : json_items(argv[3])) indexes.push_back(strtoull(it.c_str(), nullptr, 10));
    }
    char* json = nullptr;
    size_t len = 0;
    char err[512] = {0};
    const int32_t rc = aero_cairo_memory(what, proof, (size_t)m, inputs, (size_t)n, indexes.empty() ? nullptr : indexes.data(), (uint32_t)indexes.size(), &json, &len,
                                         err, sizeof err);
    if (rc != AERO_OK) die(err);
    fwrite(json, 1, len, stdout);
    fputc('\n', stdout);
    aero_free(json);
    return 0;
}

// poseidon_constants.cpp — start-up derivation of the Poseidon round constants (host).
// Upstream plonky2 generated ALL_ROUND_CONSTANTS with ChaCha8Rng::seed_from_u64(0) and
// Rng::gen_range(0..ORDER) (rand 0.8: widening multiply, reject when the low word exceeds the zone).
#include <cstring>
#include <mutex>
#include "poseidon.hpp"

namespace poseidon {
namespace {
struct ChaCha8 {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t block[16];
    int pos = 16;
    static uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
    static void quarter(uint32_t *x, int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    }
    explicit ChaCha8(uint64_t seed) {
        // rand_core::SeedableRng::seed_from_u64: PCG32 stream fills the 32-byte key
        uint64_t st = seed;
        for (int i = 0; i < 8; i++) {
            st = st * 6364136223846793005ULL + 11634580027462260723ULL;
            uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27), rot = (uint32_t)(st >> 59);
            key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
        }
    }
    void refill() {
        uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        std::memcpy(in + 4, key, sizeof key);
        in[12] = (uint32_t)counter; in[13] = (uint32_t)(counter >> 32); in[14] = 0; in[15] = 0;
        uint32_t x[16];
        std::memcpy(x, in, sizeof x);
        for (int dr = 0; dr < 4; dr++) {
            quarter(x, 0, 4, 8, 12); quarter(x, 1, 5, 9, 13); quarter(x, 2, 6, 10, 14); quarter(x, 3, 7, 11, 15);
            quarter(x, 0, 5, 10, 15); quarter(x, 1, 6, 11, 12); quarter(x, 2, 7, 8, 13); quarter(x, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) block[i] = x[i] + in[i];
        counter++; pos = 0;
    }
    uint32_t next32() { if (pos == 16) refill(); return block[pos++]; }
    uint64_t next64() { uint64_t lo = next32(); uint64_t hi = next32(); return (hi << 32) | lo; }
    uint64_t below(uint64_t range) {
        const uint64_t zone = (range << __builtin_clzll(range)) - 1;
        for (;;) {
            unsigned __int128 m = (unsigned __int128)next64() * range;
            if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
        }
    }
};
u64 g_rc[ROUNDS * WIDTH];
std::once_flag g_once;
}  // namespace

void derive_round_constants(u64 *out360) {
    ChaCha8 rng(0);
    for (int i = 0; i < ROUNDS * WIDTH; i++) out360[i] = rng.below(gl::P);
}
const u64 *host_round_constants() {
    std::call_once(g_once, [] { derive_round_constants(g_rc); });
    return g_rc;
}
}  // namespace poseidon

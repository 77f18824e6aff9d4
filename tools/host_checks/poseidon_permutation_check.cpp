// poseidon_permutation_check.cpp — the library's Poseidon permutation (poseidon.hpp: full rounds + the 22 partial rounds as one
// spectral run with renormalisation) against its layer-wise form and the textbook schedule, on the host (the same GL_HD code the
// kernels compile): random states and the extremes of the value range (all ones, p - 1 - small, 32-bit values, multiples of
// 2^32), loose (non-canonical) inputs included. Also Poseidon2: the multiplication-free external layer against the general one,
// and the 192-bit accumulator of the quotient kernels against reduce-every-term. Built and run by tests/test_host_checks.py.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "poseidon.hpp"
#include "poseidon_mfma.hpp"
using gl::u64;
int check_poseidon() {
    const u64 *rc = poseidon::host_hash_round_constants(), *rcp = poseidon::host_round_constants();
    u64 seed = 12345; auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    int bad = 0;
    for (int t = 0; t < 200000; t++) {
        u64 a[12], b[12], c[12];
        for (int i = 0; i < 12; i++) {
            u64 v = rnd();
            if (t % 7 == 0) v = (t % 14 == 0) ? 0xFFFFFFFFFFFFFFFFull : gl::P - 1 - (v & 3);      // extremes: loose all-ones, near p
            if (t % 11 == 0) v &= 0xFFFFFFFFull;
            if (t % 13 == 0) v = v << 32;
            a[i] = b[i] = c[i] = v;
        }
        poseidon::permute(a, rc); poseidon::permute_layerwise(b, rc);
        for (int i = 0; i < 12; i++) c[i] = gl::canon(c[i]);
        poseidon::permute_textbook(c, rcp);
        for (int i = 0; i < 12; i++) if (a[i] != b[i] || a[i] != c[i]) bad++;
    }
    printf("poseidon permutation: mismatches %d\n", bad);
    return bad;
}
int check_poseidon2() {
    const poseidon2::Params &P = poseidon2::qp_params();
    u64 seed = 99; auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    int bad = 0;
    for (int t = 0; t < 50000; t++) {
        u64 a[12], b[12];
        for (int i = 0; i < 12; i++) { u64 v = rnd(); if (t % 5 == 0) v = ~0ull - (v & 7); a[i] = b[i] = v; }
        poseidon2::permute(a, P); poseidon2::permute_qp(b, P);
        for (int i = 0; i < 12; i++) bad += a[i] != b[i];
    }
    printf("poseidon2 permute_qp: mismatches %d\n", bad);
    return bad;
}
int check_acc() {
    u64 seed = 7; auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    int bad = 0;
    for (int t = 0; t < 20000; t++) {
        gl::Acc192 acc = gl::acc_zero(); u64 ref = 0;
        const int n = 1 + (int)(rnd() % 300);
        for (int i = 0; i < n; i++) { u64 a = rnd(), b = rnd(); if (t % 3 == 0) { a = ~0ull; b = ~0ull - (rnd() & 3); } gl::acc_mul(acc, a, b); ref = gl::add(ref, gl::mul(a, b)); }
        if (gl::canon(gl::acc_reduce(acc)) != gl::canon(ref)) bad++;
    }
    printf("192-bit accumulator: mismatches %d\n", bad);
    return bad;
}
// the matrix form of the partial rounds (poseidon_mfma.hpp): the integer emulation of the device schedule, computed from the very
// table bytes the kernels load (digits, accumulator start values, recombination), against the permutation; a limb outside
// [0, 2^24) fails the emulation
int check_matrix_form() {
    const u64 *rc = poseidon::host_hash_round_constants();
    std::vector<unsigned char> tab(pmf::TABLE_BYTES);
    if (!pmf::build_tables(rc, tab.data())) { printf("matrix form: table construction failed\n"); return 1; }
    u64 seed = 4242; auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    int bad = 0;
    for (int t = 0; t < 20000; t++) {
        u64 a[12], b[12];
        for (int i = 0; i < 12; i++) {
            u64 v = rnd();
            if (t % 7 == 0) v = (t % 14 == 0) ? 0xFFFFFFFFFFFFFFFFull : gl::P - 1 - (v & 3);
            if (t % 11 == 0) v &= 0xFFFFFFFFull;
            if (t % 13 == 0) v = v << 32;
            if (t % 17 == 0) v = 0;
            a[i] = b[i] = v;
        }
        poseidon::permute(a, rc);
        if (!pmf::host::emu_permute(b, rc, tab.data())) { bad += 12; continue; }
        for (int i = 0; i < 12; i++) bad += a[i] != b[i];
    }
    // digit conversion at the edges of its two cases
    const u64 edge[] = {0, 1, 0x7F7F7F7F7F7F7F7Full, 0x7F7F7F7F7F7F7F80ull, 0x7F7F7F7F7F7F7F81ull, gl::P - 1, gl::P, ~0ull, 0x8000000000000000ull};
    for (u64 v : edge) {
        const u64 t = pmf::to_digits(v);
        __int128 sum = 0;
        for (int bq = 7; bq >= 0; bq--) sum = sum * 256 + (signed char)(t >> (8 * bq));
        const __int128 want = (__int128)(v % gl::P);
        __int128 got = sum % (__int128)gl::P; if (got < 0) got += gl::P;
        bad += got != want;
    }
    printf("matrix form of the partial rounds: mismatches %d\n", bad);
    return bad;
}
// the same for qp-poseidon-core's Poseidon2 (internal rounds, the external layer before them, the constants behind them): the
// emulation against the GENERAL form of the permutation (caller-supplied block products), 20 000 states
int check_matrix_form_p2() {
    const poseidon2::Params &P = poseidon2::qp_params();
    std::vector<unsigned char> tab(pmf::TABLE_BYTES);
    if (!pmf::build_tables_p2(P, tab.data())) { printf("Poseidon2 matrix form: table construction failed\n"); return 1; }
    u64 seed = 777; auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    int bad = 0;
    for (int t = 0; t < 20000; t++) {
        u64 a[12], b[12];
        for (int i = 0; i < 12; i++) {
            u64 v = rnd();
            if (t % 7 == 0) v = (t % 14 == 0) ? 0xFFFFFFFFFFFFFFFFull : gl::P - 1 - (v & 3);
            if (t % 11 == 0) v &= 0xFFFFFFFFull;
            if (t % 13 == 0) v = v << 32;
            if (t % 17 == 0) v = 0;
            a[i] = b[i] = v;
        }
        poseidon2::permute(a, P);
        if (!pmf::host::emu_permute_p2(b, P, tab.data())) { bad += 12; continue; }
        for (int i = 0; i < 12; i++) bad += a[i] != b[i];
    }
    printf("Poseidon2 matrix form of the internal rounds: mismatches %d\n", bad);
    return bad;
}
// the table with the PLAIN round constants (twelve lanes in every partial round: the schedule PoseidonGate's constraints are written
// against) against the textbook permutation, through the generic builder
int check_matrix_form_gate() {
    const u64 *rcp = poseidon::host_round_constants();
    std::vector<unsigned char> tab(pmf::TABLE_BYTES);
    if (!pmf::build_tables_gate(rcp, tab.data())) { printf("gate table: construction failed\n"); return 1; }
    const int bad = pmf::host_selfcheck_gate(rcp, tab.data(), 5000) ? 0 : 1;
    printf("matrix form with the plain round constants (gate table): mismatches %d\n", bad);
    return bad;
}
int main() { return (check_poseidon() | check_poseidon2() | check_acc() | check_matrix_form() | check_matrix_form_p2() | check_matrix_form_gate()) != 0; }

// poseidon_mfma_microbench.hip — the 22 partial rounds of the Poseidon permutation as ONE constant matrix applied on the matrix
// pipe (v_mfma_i32_32x32x32_i8), measured against the library's spectral form (round-3 experiment; record under profiles/).
//
// Idea. In a partial round only lane 0 is non-linear. With N = M P (M the MDS matrix, P = zero lane 0), m0 = M e0 and y_k the S-box
// output of round k, the S-box INPUT of round k and the state after the last round are affine in (s, y_0 .. y_{k-1}):
//     x_k  = (e0^T N^k)(s + c') + sum_{j<k} (e0^T N^{k-1-j} m0) y_j + c_k,      out = N^22 (s + c') + sum_j y_j N^{21-j} m0 + c''
// so the whole linear part is one 34-column matrix of dense 64-bit field constants times the vector (s[0..11], y_0 .. y_21): 21 + 12
// output rows instead of 22 MDS layers. A 64-bit modular matrix product is an int8 GEMM on byte digits: inputs as 8 signed base-256
// digits (in the lane's own registers: the state of lane n IS column n of the B operand, 8 digits per element along K), constants as
// signed digits Toeplitz-expanded over 12 output limbs (limbs 12..15 folded with 2^96 = -1 into limbs 0..3 at no cost: the two index
// sets are disjoint), accumulators initialised with 2^23 + the additive constant's bytes so every limb comes out in [0, 2^24) and the
// recombination is byte permutes + one 128-bit add chain + one reduce128 (28 vector instructions per output against ~270 for an MDS
// layer in the spectral form). The y_j of the running group of four rounds are not in the B operand yet: their three coefficients
// are small integers (25, 14 882, 6 935 649) and are applied on the VALU.
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -I qp-zk-circuits_amd/csrc tools/poseidon_mfma_microbench.hip -o tools/scratch_bin/poseidon_mfma_microbench
#define POSEIDON_GROUPED_SBOX 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "poseidon_mfma.hpp"

using gl::u32;
using gl::u64;

// ---------- operand layout probe ----------
__global__ void probe_kernel(const int *a, const int *b, int *d) {
    const int l = threadIdx.x;
    pmf::v4i av, bv;
    pmf::v16i c;
    for (int i = 0; i < 4; i++) { av[i] = a[l * 4 + i]; bv[i] = b[l * 4 + i]; }
    for (int i = 0; i < 16; i++) c[i] = 0;
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
    for (int i = 0; i < 16; i++) d[l * 16 + i] = c[i];
}
static bool probe_layout() {
    // hypothesis: lane l holds A[row l&31][k = 16 (l>>5) + j] and B[k = 16 (l>>5) + j][col l&31] in byte j of its 16-byte fragment;
    // D[row (i&3) + 8 (i>>2) + 4 (l>>5)][col l&31] in register i
    std::vector<int8_t> A(32 * 32), B(32 * 32);
    srand(7);
    for (auto &v : A) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : B) v = (int8_t)(rand() % 256 - 128);
    std::vector<int> fa(64 * 4), fb(64 * 4), hd(64 * 16);
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 16; j++) {
            const int k = 16 * (l >> 5) + j;
            ((int8_t *)&fa[l * 4])[j] = A[(l & 31) * 32 + k];
            ((int8_t *)&fb[l * 4])[j] = B[k * 32 + (l & 31)];
        }
    int *da, *db, *dd;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 4096);
    hipMemcpy(da, fa.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, fb.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd.data(), dd, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 16; i++) {
            const int m = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), n = l & 31;
            int ref = 0;
            for (int k = 0; k < 32; k++) ref += (int)A[m * 32 + k] * (int)B[k * 32 + n];
            if (ref != hd[l * 16 + i]) bad++;
        }
    printf("layout probe (i8 32x32x32, k = 16 h + j, D row = (i&3) + 8 (i>>2) + 4 h): %s (%d mismatches of 1024)\n", bad ? "FAIL" : "ok", bad);
    hipFree(da); hipFree(db); hipFree(dd);
    return bad == 0;
}

// ---------- kernels ----------
#ifndef PMF_WPE
#define PMF_WPE 4
#endif
template <int WG, int MODE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(PMF_WPE, PMF_WPE))) void perm_kernel(u64 *out, const u64 *rc, const uint4 *tables, int iters, const poseidon2::Params *p2 = nullptr) {
    extern __shared__ uint4 lds[];
    if (MODE == 1 || MODE == 3) {
        for (int i = threadIdx.x; i < pmf::TABLE_BYTES / 16; i += WG) lds[i] = tables[i];
        __syncthreads();
    }
    u64 s[12];
    const u64 t = threadIdx.x + blockIdx.x * (u64)blockDim.x;
    for (int i = 0; i < 12; i++) s[i] = t * 0x9E3779B97F4A7C15ull + i;
    for (int it = 0; it < iters; it++) {
        if (MODE == 1) pmf::permute(s, rc, (const unsigned char *)lds);
        else if (MODE == 2) poseidon2::permute_qp(s, *p2);          // Poseidon2, qp-poseidon-core's set, multiplication-free external layers
        else if (MODE == 3) pmf::permute_p2qp(s, *p2, (const unsigned char *)lds);
        else if (MODE == 4) poseidon2::permute(s, *p2);             // Poseidon2 through the general parameter plug
        else poseidon::permute(s, rc);
    }
    if (iters == 1) {
        for (int i = 0; i < 12; i++) out[t * 12 + i] = s[i];
    } else {
        u64 x = 0;
        for (int i = 0; i < 12; i++) x ^= s[i];
        out[t] = x;
    }
}

template <int WG, int MODE>
static void run(const char *name, int iters, const u64 *rc, const uint4 *tables, int waves_per_simd_note, const poseidon2::Params *p2 = nullptr) {
    const int total = 256 * 16 * 256;
    const int blocks = (total + WG - 1) / WG;
    u64 *out;
    hipMalloc(&out, (size_t)blocks * WG * 8);
    const size_t shm = (MODE == 1 || MODE == 3) ? pmf::TABLE_BYTES : 0;
    if (shm) hipFuncSetAttribute((const void *)perm_kernel<WG, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((perm_kernel<WG, MODE>), dim3(blocks), dim3(WG), shm, 0, out, rc, tables, 2, p2);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(err)); return; }
    float best = 1e9; u64 chk = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((perm_kernel<WG, MODE>), dim3(blocks), dim3(WG), shm, 0, out, rc, tables, iters, p2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(&chk, out + 12345, 8, hipMemcpyDeviceToHost);
    printf("%-44s %8.3f ms  %8.3f G/s  chk %016llx\n", name, best, (double)blocks * WG * iters / best / 1e6, (unsigned long long)chk);
    hipFree(out);
}

template <int WG>
static bool check(const u64 *rc_dev, const u64 *rc_host, const uint4 *tables) {
    const int blocks = 4;
    u64 *out; hipMalloc(&out, (size_t)blocks * WG * 12 * 8);
    hipFuncSetAttribute((const void *)perm_kernel<WG, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pmf::TABLE_BYTES);
    hipLaunchKernelGGL((perm_kernel<WG, 1>), dim3(blocks), dim3(WG), pmf::TABLE_BYTES, 0, out, rc_dev, tables, 1);
    std::vector<u64> h((size_t)blocks * WG * 12);
    hipError_t err = hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    if (err != hipSuccess) { printf("check: %s\n", hipGetErrorString(err)); return false; }
    int bad = 0;
    std::vector<int> wrong_lanes(blocks * WG / 64, 0);
    for (int t = 0; t < blocks * WG; t++) {
        u64 s[12];
        for (int i = 0; i < 12; i++) s[i] = (u64)t * 0x9E3779B97F4A7C15ull + i;
        poseidon::permute(s, rc_host);
        for (int i = 0; i < 12; i++)
            if (s[i] != h[(size_t)t * 12 + i]) { if (i == 0) wrong_lanes[t / 64]++; if (bad < 4) printf("  mismatch thread %d lane %d: %016llx vs %016llx\n", t, i, (unsigned long long)h[(size_t)t * 12 + i], (unsigned long long)s[i]); bad++; }
    }
    printf("MFMA permutation vs host permutation, %d states (WG %d): %s (%d mismatching words)\n", blocks * WG, WG, bad ? "FAIL" : "bit-exact", bad);
    if (bad) { printf("  wrong lanes per wave:"); for (size_t w = 0; w < wrong_lanes.size(); w++) printf(" %d", wrong_lanes[w]); printf("\n"); }
    hipFree(out);
    return bad == 0;
}

template <int WG>
static bool check_p2(const poseidon2::Params *p2_dev, const poseidon2::Params &p2, const uint4 *tables) {
    const int blocks = 4;
    u64 *out; hipMalloc(&out, (size_t)blocks * WG * 12 * 8);
    hipFuncSetAttribute((const void *)perm_kernel<WG, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pmf::TABLE_BYTES);
    hipLaunchKernelGGL((perm_kernel<WG, 3>), dim3(blocks), dim3(WG), pmf::TABLE_BYTES, 0, out, (const u64 *)nullptr, tables, 1, p2_dev);
    std::vector<u64> h((size_t)blocks * WG * 12);
    if (hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("check_p2: copy failed\n"); return false; }
    int bad = 0;
    for (int t = 0; t < blocks * WG; t++) {
        u64 s[12];
        for (int i = 0; i < 12; i++) s[i] = (u64)t * 0x9E3779B97F4A7C15ull + i;
        poseidon2::permute(s, p2);                          // the general form on the host
        for (int i = 0; i < 12; i++) bad += s[i] != h[(size_t)t * 12 + i];
    }
    printf("Poseidon2 matrix form vs host poseidon2::permute, %d states (WG %d): %s (%d mismatching words)\n", blocks * WG, WG, bad ? "FAIL" : "bit-exact", bad);
    hipFree(out);
    return bad == 0;
}

int main(int argc, char **argv) {
    u64 h[360];
    for (int i = 0; i < 360; i++) h[i] = (0x123456789ABCDEFull * (i + 1)) % 0xFFFFFFFF00000001ull;
    u64 *rc; hipMalloc(&rc, sizeof h); hipMemcpy(rc, h, sizeof h, hipMemcpyHostToDevice);
    if (!probe_layout()) return 2;
    std::vector<unsigned char> tab(pmf::TABLE_BYTES);
    if (!pmf::build_tables(h, tab.data())) { printf("table construction failed\n"); return 3; }
    if (!pmf::host_selfcheck(h, tab.data())) { printf("host self-check of the tables failed\n"); return 4; }
    uint4 *tables; hipMalloc(&tables, tab.size()); hipMemcpy(tables, tab.data(), tab.size(), hipMemcpyHostToDevice);
    bool ok = check<64>(rc, h, tables);
    ok = check<256>(rc, h, tables) && ok;
    ok = check<1024>(rc, h, tables) && ok;
    ok = check<512>(rc, h, tables) && ok;
    if (!ok && !(argc > 1 && !strcmp(argv[1], "force"))) return 5;
    // Poseidon2 with qp-poseidon-core's parameters
    const poseidon2::Params &P2 = poseidon2::qp_params();
    std::vector<unsigned char> tab2(pmf::TABLE_BYTES);
    if (!pmf::build_tables_p2(P2, tab2.data()) || !pmf::host_selfcheck_p2(P2, tab2.data())) { printf("Poseidon2 table construction / self-check failed\n"); return 6; }
    uint4 *tables2; hipMalloc(&tables2, tab2.size()); hipMemcpy(tables2, tab2.data(), tab2.size(), hipMemcpyHostToDevice);
    poseidon2::Params *p2d; hipMalloc(&p2d, sizeof P2); hipMemcpy(p2d, &P2, sizeof P2, hipMemcpyHostToDevice);
    ok = check_p2<512>(p2d, P2, tables2) && ok;
    ok = check_p2<64>(p2d, P2, tables2) && ok;
    if (!ok && !(argc > 1 && !strcmp(argv[1], "force"))) return 5;
    run<256, 4>("Poseidon2 permute, general plug (WG 256)", 64, rc, tables2, 0, p2d);
    run<256, 2>("Poseidon2 permute_qp (WG 256)", 64, rc, tables2, 0, p2d);
    run<512, 3>("Poseidon2, internal rounds on MFMA (WG 512)", 64, rc, tables2, 0, p2d);
    run<256, 0>("permute, lib spectral (WG 256)", 64, rc, tables, 0);
    run<256, 1>("permute, partial rounds on MFMA (WG 256)", 64, rc, tables, 0);
    run<512, 1>("permute, partial rounds on MFMA (WG 512)", 64, rc, tables, 0);
    run<1024, 1>("permute, partial rounds on MFMA (WG 1024)", 64, rc, tables, 0);

    return 0;
}

// valu_rates.hip — per-instruction issue rates on gfx950 for the integer ops the field code uses.
// Each kernel runs 8 independent chains per thread, 256 iterations x 8 x UNROLL instrs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32; typedef unsigned long long u64;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k(u32 *out, int iters, u64 *clk) {
    u32 a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 7 + i; b[i] = blockIdx.x + 13 * i + 1; }
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 1) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (OP == 2) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(a[(i + 1) & 7]) : "vcc");
                if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : );
                if (OP == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 5) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 6) { u64 r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a[i]), "v"(b[i]) : "vcc"); a[i] = (u32)r; b[i] ^= (u32)(r >> 32); }
                if (OP == 7) { u64 r = ((u64)b[i] << 32) | a[i]; asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(r)); a[i] = (u32)r; b[i] = (u32)(r >> 32); }
                if (OP == 8) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 10) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 11) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 12) asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (OP == 13) { u64 r = ((u64)b[i] << 32) | a[i]; asm volatile("v_lshlrev_b64 %0, 5, %0" : "+v"(r)); a[i] = (u32)r; b[i] = (u32)(r >> 32); }
                if (OP == 14) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            }
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    u32 acc = 0;
    for (int i = 0; i < 8; i++) acc ^= a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}
template <int OP> void run(const char *name, int instr_per_slot, int waves_per_simd) {
    u32 *out; u64 *clk; hipMalloc(&out, 1 << 24); hipMalloc(&clk, 8);
    int blocks = 256 * waves_per_simd, threads = 256, iters = 2048;  // 256 thr = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 8, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    u64 c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    double winstr = (double)iters * 32 * instr_per_slot;           // wave-instructions per wave
    double cyc_per_instr_wave = (double)c / winstr;                 // cycles per instr as seen by one wave (memtime ticks at 100MHz? see ratio)
    double total_winstr = winstr * blocks * 4;                      // all waves
    double per_simd_per_s = total_winstr / 1024 / (ms * 1e-3);
    printf("%-34s waves/SIMD %d  %7.3f ms  %7.1f M wave-instr/s/SIMD  => %5.2f cycles/instr @2.4GHz (memtime ticks/instr/wave %.2f)\n",
           name, waves_per_simd, ms, per_simd_per_s / 1e6, 2.4e9 / per_simd_per_s, cyc_per_instr_wave);
    hipFree(out); hipFree(clk);
}
int main() {
    for (int w : {1, 4, 8}) {
        run<0>("v_add_u32", 1, w);
        run<1>("v_add_co_u32 (vcc out)", 1, w);
        run<2>("v_add_co + v_addc_co pair", 2, w);
        run<12>("v_subb_co_u32 (vcc in/out)", 1, w);
        run<3>("v_cndmask_b32", 1, w);
        run<14>("v_cmp + v_cndmask", 2, w);
        run<4>("v_mul_lo_u32", 1, w);
        run<5>("v_mul_hi_u32", 1, w);
        run<6>("v_mad_u64_u32", 1, w);
        run<7>("v_lshl_add_u64", 1, w);
        run<13>("v_lshlrev_b64", 1, w);
        run<8>("v_alignbit_b32", 1, w);
        run<9>("v_mad_u32_u24", 1, w);
        run<10>("v_xor_b32", 1, w);
        run<11>("v_add3_u32", 1, w);
        printf("\n");
    }
    return 0;
}

// microbench.hip — throughput of the Goldilocks primitives on gfx950 (ALU roofline inputs for DESIGN.md).
// Build: hipcc --offload-arch=gfx950 -O3 -I qp-zk-circuits_amd/csrc tools/microbench.hip -o /tmp/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gl64.hpp"
using gl::u64;

template <int OP>
__global__ void k(u64 *out, u64 seed, int iters) {
    u64 a = seed + threadIdx.x + blockIdx.x * blockDim.x, b = a * 0x9E3779B97F4A7C15ull + 1, c = b ^ 0x1234567, d = c + 99;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { a = gl::mul(a, b); b = gl::mul(b, c); c = gl::mul(c, d); d = gl::mul(d, a); }
        if (OP == 1) { a = gl::add(a, b); b = gl::add(b, c); c = gl::add(c, d); d = gl::add(d, a); }
        if (OP == 2) { a = gl::sub(a, b); b = gl::sub(b, c); c = gl::sub(c, d); d = gl::sub(d, a); }
        if (OP == 3) { a = gl::mul_pow2<24>(a) + 1; b = gl::mul_pow2<66>(b) + 1; c = gl::mul_pow2<6>(c) + 1; d = gl::mul_pow2<48>(d) + 1; }
        if (OP == 4) { a = a * b + c; b = b * c + d; c = c * d + a; d = d * a + b; }   // 64-bit mad lo
        if (OP == 5) { a = __umul64hi(a, b) + c; b = __umul64hi(b, c) + d; c = __umul64hi(c, d) + a; d = __umul64hi(d, a) + b; }
        if (OP == 6) { unsigned x = a, y = b, z = c, w = d; x = x * y + z; y = y * z + w; z = z * w + x; w = w * x + y; a = x; b = y; c = z; d = w; }
        if (OP == 7) { u64 x2 = gl::sqr(a), x4 = gl::sqr(x2), x3 = gl::mul(a, x2); a = gl::mul(x3, x4) + b; b += 1; }
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = a ^ b ^ c ^ d;
}
__global__ void copy_k(const ulonglong2 *in, ulonglong2 *out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
template <int OP> void run(const char *name, int per_iter) {
    u64 *out; hipMalloc(&out, 1024 * 1024 * 8 * 8);
    int blocks = 256 * 8, threads = 256, iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * threads * iters * per_iter;
    printf("%-28s %8.3f ms  %8.2f Gop/s\n", name, ms, ops / ms / 1e6);
    hipFree(out);
}
int main() {
    run<0>("gl::mul (4 indep chains)", 4);
    run<1>("gl::add", 4);
    run<2>("gl::sub", 4);
    run<3>("gl::mul_pow2 (24,66,6,48)", 4);
    run<4>("u64 mad lo", 4);
    run<5>("u64 mulhi", 4);
    run<6>("u32 mad", 4);
    run<7>("sbox x^7 (4 mul)", 1);
    size_t bytes = 1ull << 30; ulonglong2 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(copy_k, dim3(256 * 8), dim3(256), 0, 0, a, b, bytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy 1 GiB: %.3f ms  %.1f GB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e6);
    }
    return 0;
}

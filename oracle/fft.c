/*
 * oracle/fft.c — CPU restatement of plonky2::field::fft (qp-plonky2-field 1.5.5, un-vendored;
 * reference call site: every `prove` in /root/reference, e.g. wormhole/prover/src/lib.rs:171-175,
 * reaches PolynomialBatch::from_values / from_coeffs which call these transforms).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/gl.h header).
 *
 * Conventions restated:
 *   fft(coeffs)[i]  = P(w_n^i), natural order in and out, w_n = primitive_root_of_unity(log n)
 *   ifft(values)[i] = n^-1 * FFT(values)[(n - i) mod n]
 *   coset_fft(c, s) = fft(c[i] * s^i);   coset_ifft = ifft then * s^-i
 *   lde(rate_bits)  = zero-pad coefficients to n << rate_bits
 * Field elements have one canonical form, so the output does not depend on the butterfly schedule.
 */
#include "gl.h"
#include <stdlib.h>
#include <string.h>

static void bitrev_permute(gl_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev32((uint32_t)i, log_n);
        if (i < j) { gl_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

/* In-place, natural -> natural. Bit-reverse then decimation-in-time. */
void orc_fft(gl_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    if (log_n == 0) return;
    bitrev_permute(a, log_n);
    gl_t *tw = (gl_t *)malloc(sizeof(gl_t) * (n / 2 ? n / 2 : 1));
    gl_t w = gl_root_of_unity(log_n);
    tw[0] = 1;
    for (size_t i = 1; i < n / 2; i++) tw[i] = gl_mul(tw[i - 1], w);
    for (unsigned s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, half = m >> 1, step = n >> s;
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; j++) {
                gl_t t = gl_mul(tw[j * step], a[k + j + half]);
                gl_t u = a[k + j];
                a[k + j] = gl_add(u, t);
                a[k + j + half] = gl_sub(u, t);
            }
    }
    free(tw);
}

void orc_ifft(gl_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    orc_fft(a, log_n);
    gl_t n_inv = gl_inv((gl_t)n % GL_P);
    if (n == 1) return;
    a[0] = gl_mul(a[0], n_inv);
    a[n / 2] = gl_mul(a[n / 2], n_inv);
    for (size_t i = 1; i < n / 2; i++) {
        size_t j = n - i;
        gl_t ci = gl_mul(a[j], n_inv), cj = gl_mul(a[i], n_inv);
        a[i] = ci; a[j] = cj;
    }
}

void orc_coset_fft(gl_t *a, unsigned log_n, gl_t shift) {
    size_t n = (size_t)1 << log_n;
    gl_t s = 1;
    for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], s); s = gl_mul(s, shift); }
    orc_fft(a, log_n);
}

void orc_coset_ifft(gl_t *a, unsigned log_n, gl_t shift) {
    size_t n = (size_t)1 << log_n;
    orc_ifft(a, log_n);
    gl_t si = gl_inv(shift), s = 1;
    for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], s); s = gl_mul(s, si); }
}

/* out (n << rate_bits) = coset_fft(zero-padded coeffs, shift) */
void orc_lde(const gl_t *coeffs, unsigned log_n, unsigned rate_bits, gl_t shift, gl_t *out) {
    size_t n = (size_t)1 << log_n, big = n << rate_bits;
    memcpy(out, coeffs, n * sizeof(gl_t));
    memset(out + n, 0, (big - n) * sizeof(gl_t));
    orc_coset_fft(out, log_n + rate_bits, shift);
}

/* O(n^2) definition-level DFT, used only to pin orc_fft on small sizes. */
void orc_dft_naive(const gl_t *in, gl_t *out, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    gl_t w = gl_root_of_unity(log_n);
    for (size_t i = 0; i < n; i++) {
        gl_t wi = gl_pow(w, i), acc = 0, x = 1;
        for (size_t j = 0; j < n; j++) { acc = gl_add(acc, gl_mul(in[j], x)); x = gl_mul(x, wi); }
        out[i] = acc;
    }
}

/* Batched column-major helpers (column c at data + c*n). Threads: OpenMP if compiled with it. */
void orc_fft_batch(gl_t *data, unsigned log_n, size_t batch, int inverse) {
    size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)batch; c++) {
        if (inverse) orc_ifft(data + (size_t)c * n, log_n);
        else orc_fft(data + (size_t)c * n, log_n);
    }
}

/* values (n per column) -> coset-LDE values (n << rate_bits per column); PolynomialBatch::from_values
 * without the Merkle step. coeffs_out (optional) receives ifft(values). */
void orc_lde_batch(const gl_t *values, unsigned log_n, unsigned rate_bits, size_t batch, gl_t shift,
                   gl_t *coeffs_out, gl_t *lde_out) {
    size_t n = (size_t)1 << log_n, big = n << rate_bits;
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)batch; c++) {
        gl_t *tmp = (gl_t *)malloc(n * sizeof(gl_t));
        memcpy(tmp, values + (size_t)c * n, n * sizeof(gl_t));
        orc_ifft(tmp, log_n);
        if (coeffs_out) memcpy(coeffs_out + (size_t)c * n, tmp, n * sizeof(gl_t));
        orc_lde(tmp, log_n, rate_bits, shift, lde_out + (size_t)c * big);
        free(tmp);
    }
}

/* scalar field helpers exported for the Python tests */
gl_t orc_gl_mul(gl_t a, gl_t b) { return gl_mul(a, b); }
gl_t orc_gl_add(gl_t a, gl_t b) { return gl_add(a, b); }
gl_t orc_gl_sub(gl_t a, gl_t b) { return gl_sub(a, b); }
gl_t orc_gl_inv(gl_t a) { return gl_inv(a); }
gl_t orc_gl_pow(gl_t a, uint64_t e) { return gl_pow(a, e); }
gl_t orc_gl_root(unsigned log_n) { return gl_root_of_unity(log_n); }
void orc_gl2_mul(const gl_t *a, const gl_t *b, gl_t *out) {
    gl2_t r = gl2_mul(gl2_make(a[0], a[1]), gl2_make(b[0], b[1]));
    out[0] = r.c[0]; out[1] = r.c[1];
}
void orc_gl2_inv(const gl_t *a, gl_t *out) {
    gl2_t r = gl2_inv(gl2_make(a[0], a[1]));
    out[0] = r.c[0]; out[1] = r.c[1];
}

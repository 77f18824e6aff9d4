/* oracle/challenger.h — duplex-sponge challenger state (see poseidon.c). TEST INFRASTRUCTURE ONLY. */
#ifndef ORACLE_CHALLENGER_H
#define ORACLE_CHALLENGER_H
#include "gl.h"
typedef struct {
    gl_t state[12];
    gl_t in[8]; int n_in;
    gl_t out[8]; int n_out;
} orc_challenger;
void orc_challenger_init(orc_challenger *c);
void orc_challenger_observe(orc_challenger *c, const gl_t *x, size_t n);
gl_t orc_challenger_get(orc_challenger *c);
void orc_challenger_get_n(orc_challenger *c, gl_t *out, size_t n);
gl_t orc_challenger_pow_response(const orc_challenger *c, gl_t nonce);
#endif

/*
 * oracle/poseidon2.h — parameter block of the parametric Poseidon2 in oracle/poseidon2.c (TEST INFRASTRUCTURE ONLY, see
 * oracle/gl.h), shared with the Poseidon2 gate restatement in oracle/poseidon2_gate.c.
 */
#ifndef ORACLE_POSEIDON2_H
#define ORACLE_POSEIDON2_H
#include "gl.h"
typedef struct {
    gl_t rc_ext[8][12];
    gl_t rc_int[22];
    gl_t diag_m1[12];   /* internal matrix = J + diag(diag_m1) */
    gl_t m4[4][4];
    int absorb_add;     /* 0: overwrite rate lanes, 1: add into rate lanes */
} orc_p2_params;
void orc_p2_qp_params(orc_p2_params *p);
void orc_p2_permute(const orc_p2_params *p, gl_t s[12]);
void orc_p2_hash_pad10(const orc_p2_params *p, const gl_t *in, size_t n, gl_t out[4]);
#endif

#!/usr/bin/env python3
"""Derives plonky2's fast-partial-round tables for Poseidon over Goldilocks (width 12) from the MDS matrix and the
round constants, following the published HADES optimisation (move round constants up through the linear layer;
factor M = M' * M'' with M'' sparse). Checks: (1) the fast permutation equals the textbook one, (2) recalled upstream
anchors of the generated tables."""
import json, sys
from chacha import ChaChaRng, gen_range
P = 0xFFFFFFFF00000001
T, RF, RP = 12, 8, 22
r = ChaChaRng(0, 8)
RC = [gen_range(r, P) for _ in range(360)]
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
DIAG = [8] + [0] * 11
# column-vector convention: out[r] = sum_c M[r][c] x[c]; mds_row_shf: out[r] = sum_i x[(i+r)%12] CIRC[i] + x[r] DIAG[r]
M = [[(CIRC[(c - r) % 12] + (DIAG[r] if r == c else 0)) % P for c in range(12)] for r in range(12)]

def matmul(A, B): return [[sum(A[i][k] * B[k][j] for k in range(len(B))) % P for j in range(len(B[0]))] for i in range(len(A))]
def matvec(A, x): return [sum(A[i][k] * x[k] for k in range(len(x))) % P for i in range(len(A))]
def transpose(A): return [list(r) for r in zip(*A)]
def inverse(A):
    n = len(A); a = [list(A[i]) + [1 if i == j else 0 for j in range(n)] for i in range(n)]
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r][c] % P)
        a[c], a[piv] = a[piv], a[c]
        inv = pow(a[c][c], P - 2, P); a[c] = [v * inv % P for v in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]; a[r] = [(v - f * w) % P for v, w in zip(a[r], a[c])]
    return [row[n:] for row in a]

def sbox(x): return pow(x, 7, P)
def perm_naive(s):
    s = list(s); rc = 0
    for _ in range(4):
        s = [sbox((s[i] + RC[rc * 12 + i]) % P) for i in range(12)]; s = matvec(M, s); rc += 1
    for _ in range(22):
        s = [(s[i] + RC[rc * 12 + i]) % P for i in range(12)]; s[0] = sbox(s[0]); s = matvec(M, s); rc += 1
    for _ in range(4):
        s = [sbox((s[i] + RC[rc * 12 + i]) % P) for i in range(12)]; s = matvec(M, s); rc += 1
    return s

# ---- equivalent constants: move constants of partial round k+1 up behind the S-box of round k ----
Minv = inverse(M)
C = [RC[(4 + k) * 12:(5 + k) * 12] for k in range(22)]      # constants of the partial rounds
post = [0] * 22                                             # scalar added to element 0 after the S-box of round k
for k in range(20, -1, -1):
    w = matvec(Minv, C[k + 1])
    C[k] = [C[k][0]] + [(C[k][i] + w[i]) % P for i in range(1, 12)]
    post[k] = w[0]
    C[k + 1] = None
FIRST = C[0]                                               # FAST_PARTIAL_FIRST_ROUND_CONSTANT
ROUND_CONSTANTS = post[:21]                                # FAST_PARTIAL_ROUND_CONSTANTS (+ a trailing 0 upstream)

# ---- equivalent matrices (hadeshash calc_equivalent_matrices, written with column vectors) ----
# x <- M x with M = M' M'', M' = diag(1, Mhat), M'' = [[m00, v^T],[what, I]], what = Mhat^-1 w; M' moves into the previous round.
VS, WHATS = [], []
Mmul = M
for i in range(RP - 1, -1, -1):
    # Mmul = M'' * M' with M' = diag(1, B) (moves in front of this round's S-box, i.e. into the previous round's
    # linear layer) and M'' = [[m00, what^T],[c, I]] sparse
    B = [row[1:] for row in Mmul[1:]]
    c = [Mmul[r][0] for r in range(1, 12)]      # first column below the corner  -> "V"
    rr = Mmul[0][1:]                            # first row right of the corner
    what = matvec(inverse(transpose(B)), rr)    # what^T = rr^T B^-1              -> "W_HAT"
    VS.append(c); WHATS.append(what)
    Mi = [[1 if r == cc else 0 for cc in range(12)] for r in range(12)]
    for r in range(1, 12):
        for cc in range(1, 12): Mi[r][cc] = B[r - 1][cc - 1]
    Mmul = matmul(Mi, M)
INIT = [row[1:] for row in Mi[1:]]              # applied to elements 1..11 before the first partial round
VS.reverse(); WHATS.reverse()
M00 = M[0][0]

def perm_fast(s):
    s = list(s); rc = 0
    for _ in range(4):
        s = [sbox((s[i] + RC[rc * 12 + i]) % P) for i in range(12)]; s = matvec(M, s); rc += 1
    s = [(s[i] + FIRST[i]) % P for i in range(12)]
    s = [s[0]] + matvec(INIT, s[1:])
    for k in range(22):
        s[0] = sbox(s[0])
        if k < 21: s[0] = (s[0] + ROUND_CONSTANTS[k]) % P
        # sparse layer: new0 = m00 s0 + <row, s[1:]>; new_i = s_i + s0 * col_i
        col, row = VS[k], WHATS[k]
        d = (M00 * s[0] + sum(row[i] * s[1 + i] for i in range(11))) % P
        s = [d] + [(s[1 + i] + s[0] * col[i]) % P for i in range(11)]
    rc += 22
    for _ in range(4):
        s = [sbox((s[i] + RC[rc * 12 + i]) % P) for i in range(12)]; s = matvec(M, s); rc += 1
    return s

if __name__ == "__main__":
    import random
    random.seed(1)
    ok = True
    for _ in range(5):
        x = [random.randrange(P) for _ in range(12)]
        ok &= perm_naive(x) == perm_fast(x)
    print("fast == naive:", ok)
    print("FIRST[0..2]        ", [hex(v) for v in FIRST[:3]])
    print("ROUND_CONSTANTS[0..2]", [hex(v) for v in ROUND_CONSTANTS[:3]])
    print("VS[0][0..2]        ", [hex(v) for v in VS[0][:3]], "(recalled 0x94877900674181c3, 0xc6c67cc37a2a2bbd)")
    print("W_HATS[0][0..2]    ", [hex(v) for v in WHATS[0][:3]], "(recalled 0x3d999c961b7c63b0, 0x814e82efcd172529)")
    print("INIT[0][0..2]      ", [hex(v) for v in INIT[0][:3]])
    print("INIT^T[0][0..2]    ", [hex(v) for v in transpose(INIT)[0][:3]])

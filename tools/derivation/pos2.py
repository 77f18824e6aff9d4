from grain import consts
import itertools, struct
P = 0xFFFFFFFF00000001
C = consts(12, 8, 22, 8*12+22)
RC_B = [C[i*12:(i+1)*12] for i in range(4)]
RC_I = C[48:70]
RC_E = [C[70+i*12:70+(i+1)*12] for i in range(4)]
DIAG = [0xc3b6c08e23ba9300,0xd84b5de94a324fb6,0x0d0c371c5b35b84f,0x7964f570e7188037,0x5daf18bbd996604b,0x6743bc47b9595257,
        0x5528b9362c59bb70,0xac45e25b7127b68b,0xa2077d7dfbb606b5,0xf3faac6faee378ae,0x0c6388b51545e883,0xd27dbb6944917b60]
M4_HL = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
M4_P3 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
def ext(s, M4):
    t = []
    for b in range(3):
        x = s[4*b:4*b+4]
        t += [sum(M4[i][j]*x[j] for j in range(4)) % P for i in range(4)]
    sums = [(t[i]+t[4+i]+t[8+i]) % P for i in range(4)]
    return [(t[i] + sums[i%4]) % P for i in range(12)]
def internal(s, diag_plus):
    sm = sum(s) % P
    return [(s[i]*(DIAG[i]+diag_plus) + sm) % P for i in range(12)]
def perm(s, M4=M4_HL, diag_plus=0, init_lin=True):
    s = list(s)
    if init_lin: s = ext(s, M4)
    for r in range(4):
        s = [pow((s[i]+RC_B[r][i])%P, 7, P) for i in range(12)]; s = ext(s, M4)
    for r in range(22):
        s[0] = pow((s[0]+RC_I[r])%P, 7, P); s = internal(s, diag_plus)
    for r in range(4):
        s = [pow((s[i]+RC_E[r][i])%P, 7, P) for i in range(12)]; s = ext(s, M4)
    return s
def hash_pad(inp, mode="overwrite", **kw):
    x = list(inp) + [1]
    while len(x) % 8: x.append(0)
    st = [0]*12
    for i in range(0, len(x), 8):
        for j in range(8):
            st[j] = x[i+j] if mode=="overwrite" else (st[j] + x[i+j]) % P
        st = perm(st, **kw)
    return st[:4]
def bytes_to_u64s(b):
    b = bytes(b) + b"\x01"
    while len(b) % 4: b += b"\x00"
    return [int.from_bytes(b[i:i+4], "little") for i in range(0, len(b), 4)]
def digest_bytes(d): return b"".join(struct.pack("<Q", x) for x in d)
def b2d(b): return [int.from_bytes(b[i:i+8], "little") for i in range(0, 32, 8)]
if __name__ == "__main__":
    secret = bytes.fromhex("4c8587bd422e01d961acdc75e7d66f6761b7af7c9b1864a492f369c9d6724f05")
    want = "de68c6fcb3e38d6736b79a010e4504b98c6321f1e4d11cd8484f67c187ca090e"
    pre = bytes_to_u64s(b"wormhole") + b2d(secret)
    print(len(pre), pre[:3])
    for M4, dp, il in itertools.product((M4_HL, M4_P3), (0, 1, -1), (True, False)):
        kw = dict(M4=M4, diag_plus=dp, init_lin=il)
        h = hash_pad(hash_pad(pre, **kw), **kw)
        got = digest_bytes(h).hex()
        print("HL" if M4 is M4_HL else "P3", dp, il, got, got == want)

from chacha import *
P = 0xFFFFFFFF00000001
r = ChaChaRng(0, 8)
RC = [gen_range(r, P) for _ in range(360)]
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
DIAG = [8] + [0]*11
def mds(s):
    out = []
    for rr in range(12):
        acc = 0
        for i in range(12):
            acc += s[(i + rr) % 12] * CIRC[i]
        acc += s[rr] * DIAG[rr]
        out.append(acc % P)
    return out
def perm(s):
    s = list(s); rc = 0
    def full():
        nonlocal s, rc
        s = [(s[i] + RC[rc*12+i]) % P for i in range(12)]; rc += 1
        s = [pow(x, 7, P) for x in s]; s = mds(s)
    def part():
        nonlocal s, rc
        s = [(s[i] + RC[rc*12+i]) % P for i in range(12)]; rc += 1
        s[0] = pow(s[0], 7, P); s = mds(s)
    for _ in range(4): full()
    for _ in range(22): part()
    for _ in range(4): full()
    return s
if __name__ == "__main__":
    print([hex(x) for x in perm([0]*12)])
    print([hex(x) for x in perm(list(range(12)))])
    print(max(RC) < 0xfffeeac900011537)

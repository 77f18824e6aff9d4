# Grain-LFSR parameter generation (Poseidon reference procedure), for candidate validation only.
P = 0xFFFFFFFF00000001
def grain(field, sbox, n, t, RF, RP):
    bits = []
    def app(v, w): bits.extend(int(b) for b in bin(v)[2:].zfill(w))
    app(field,2); app(sbox,4); app(n,12); app(t,12); app(RF,10); app(RP,10)
    bits.extend([1]*30)
    assert len(bits)==80
    def step():
        nb = bits[62]^bits[51]^bits[38]^bits[23]^bits[13]^bits[0]
        bits.pop(0); bits.append(nb); return nb
    for _ in range(160): step()
    def gen():
        while True:
            nb = step()
            while nb == 0:
                step(); nb = step()
            yield step()
    return gen()
def consts(t, RF, RP, count, n=64, sbox=0):
    g = grain(1, sbox, n, t, RF, RP)
    out = []
    while len(out) < count:
        v = 0
        for _ in range(n): v = (v<<1) | next(g)
        if v < P: out.append(v)
    return out
if __name__ == "__main__":
    c = consts(12, 8, 22, 360)
    print([hex(x) for x in c[:12]])

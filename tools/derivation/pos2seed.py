import itertools
from chacha import ChaChaRng
import pos2
from pos2 import *
def sample(rng):
    while True:
        v = rng.next_u64()
        if v < P: return v
secret = bytes.fromhex("4c8587bd422e01d961acdc75e7d66f6761b7af7c9b1864a492f369c9d6724f05")
want = "de68c6fcb3e38d6736b79a010e4504b98c6321f1e4d11cd8484f67c187ca090e"
pre = bytes_to_u64s(b"wormhole") + b2d(secret)
for seed in (0,1,2,42,12345,0xdeadbeef,1337):
  for rounds in (8,12,20):
    for order in ("ext_int","seq"):
        rng = ChaChaRng(seed, rounds)
        if order == "ext_int":
            b = [[sample(rng) for _ in range(12)] for _ in range(4)]
            e = [[sample(rng) for _ in range(12)] for _ in range(4)]
            i = [sample(rng) for _ in range(22)]
        else:
            b = [[sample(rng) for _ in range(12)] for _ in range(4)]
            i = [sample(rng) for _ in range(22)]
            e = [[sample(rng) for _ in range(12)] for _ in range(4)]
        pos2.RC_B, pos2.RC_I, pos2.RC_E = b, i, e
        for M4, dp in itertools.product((M4_HL, M4_P3), (0,1)):
            kw = dict(M4=M4, diag_plus=dp, init_lin=True)
            h = hash_pad(hash_pad(pre, **kw), **kw)
            if digest_bytes(h).hex() == want: print("MATCH", seed, rounds, order, dp)
print("done")

from grain import *
import itertools
target = 0xb585f766f2144405
for field in (0,1,2):
  for sbox in (0,1,2,7):
    for n in (63,64,65):
      for t in (8,12,16):
        for RF in (8,):
          for RP in (22,21,23):
            g = grain(field, sbox, n, t, RF, RP)
            # scan first 2000 bits for the target at any alignment
            v = 0
            bits=[]
            for i in range(3000):
                v = ((v<<1)|next(g)) & ((1<<64)-1)
                if i>=63 and v==target:
                    print("FOUND", field,sbox,n,t,RF,RP,"bit offset",i-63)
print("done")

#!/usr/bin/env python3
"""Identity of the compiled NTT / hashing kernels: a hash over the sources that determine their code. The counter summaries
under profiles/ (rocprofv3 --pmc runs, collected outside bench.py) carry the identity of the library they were measured on;
bench.py reports their figures only while it still matches the library it runs, so that a kernel change cannot leave stale
counters next to freshly measured durations.  usage: python tools/kernel_id.py [ntt|hash|hash_mx]"""
import hashlib
import os
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qp-zk-circuits_amd", "csrc")
GROUPS = {
    "ntt": ["gl64.hpp", "ntt_pass.hpp", "ntt_kernel_impl.hpp", "ntt_kernels.hip", "ntt_inst_0.hip", "ntt_inst_1.hip", "ntt_inst_2.hip", "ntt_inst_3.hip", "Makefile"],
    # the matrix-pipe build of the hashing kernels alone (the routing in merkle_kernels.hip does not change their code)
    "hash_mx": ["gl64.hpp", "poseidon.hpp", "poseidon_mfma.hpp", "merkle.hpp", "merkle_kernels_mx.hip", "Makefile"],
    "hash": ["gl64.hpp", "poseidon.hpp", "poseidon_mfma.hpp", "merkle.hpp", "merkle_hash_impl.hpp", "merkle_kernels.hip", "merkle_kernels_tp.hip", "merkle_kernels_mx.hip", "Makefile"],
}


def kernel_source_id(group="ntt"):
    h = hashlib.sha256()
    for name in GROUPS[group]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_id(sys.argv[1] if len(sys.argv) > 1 else "ntt"))

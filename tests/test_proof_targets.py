"""Inner proofs as witness assignments of a recursive wrapper (include/qpgpu_batch.h "inner-proof targets"): the reference's
fill_private_batch_witness (wormhole/aggregator/src/private_batch/prover/witness.rs:15-77) and its shape preflight
ensure_proof_shape_matches_targets (wormhole/aggregator/src/common/utils.rs:295-540), message for message. Proofs come from
the CPU oracle, so this file needs no GPU."""
import ctypes

import numpy as np
import pytest

from oracle_binding import OracleCircuit

P = 0xFFFFFFFF00000001
ERR = 400


@pytest.fixture(scope="module")
def lib(pkg):
    L = pkg.load_library()
    vp, sz, cp = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p
    L.qpgpu_proof_target_shape.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), cp]
    L.qpgpu_proof_shape_of_bytes.argtypes = [vp, sz, cp, sz, vp, sz, ctypes.POINTER(sz), cp]
    L.qpgpu_ensure_proof_shape_matches_targets.argtypes = [vp, sz, vp, sz, sz, cp, cp]
    L.qpgpu_proof_target_count.argtypes = [vp, sz]; L.qpgpu_proof_target_count.restype = sz
    L.qpgpu_proof_target_values.argtypes = [vp, sz, cp, sz, sz, cp, vp, sz, ctypes.POINTER(sz), cp]
    L.qpgpu_batch_fill_proof_targets.argtypes = [vp, sz, vp, vp, sz, sz, vp, sz, sz, cp, vp, vp, sz, ctypes.POINTER(sz), cp]
    return L


@pytest.fixture(scope="module")
def inner(pkg, orc):
    """a small inner circuit with 21 public inputs and two proofs of it"""
    pack, wires, pis = pkg.synth_circuit(6, seed=61, poseidon=True, base_sum=True)
    oc = OracleCircuit(orc, pack)
    proofs = [oc.prove(wires, pis)]
    w2 = wires.copy()
    proofs.append(oc.prove(w2, pis))
    oc.close()
    return pack, pis, proofs


def target_shape(lib, pack):
    n = ctypes.c_size_t(); err = ctypes.create_string_buffer(ERR)
    assert lib.qpgpu_proof_target_shape(pack.ctypes.data, pack.size, None, 0, ctypes.byref(n), err) == 0
    out = np.zeros(n.value, dtype=np.uint32)
    assert lib.qpgpu_proof_target_shape(pack.ctypes.data, pack.size, out.ctypes.data, out.size, ctypes.byref(n), err) == 0
    return out


def bytes_shape(lib, pack, proof):
    n = ctypes.c_size_t(); err = ctypes.create_string_buffer(ERR)
    out = np.zeros(4096, dtype=np.uint32)
    rc = lib.qpgpu_proof_shape_of_bytes(pack.ctypes.data, pack.size, proof, len(proof), out.ctypes.data, out.size, ctypes.byref(n), err)
    return rc, out[:n.value].copy(), err.value.decode()


def ensure(lib, t, p, slot=3, label=b"leaf proof"):
    err = ctypes.create_string_buffer(ERR)
    rc = lib.qpgpu_ensure_proof_shape_matches_targets(t.ctypes.data, t.size, p.ctypes.data, p.size, slot, label, err)
    return rc, err.value.decode()


def test_target_shape_and_count(pkg, lib, inner):
    pack, pis, proofs = inner
    h = pkg.pack_header(pack)
    t = target_shape(lib, pack)
    ncs = h["num_selectors"] + h["num_constants"] + h["num_routed_wires"]
    assert t[:13].tolist() == [21, 16, 16, 16, h["num_selectors"] + h["num_constants"], 80, 135, 2, 2, 18, 16, 0, 0]
    n_caps = int(t[13]); assert n_caps == h["num_arity_rounds"] and t[14:14 + n_caps].tolist() == [16] * n_caps
    assert int(t[14 + n_caps]) == 28
    first_round = t[15 + n_caps:]
    L = h["degree_bits"] + 3
    assert first_round[:9].tolist() == [4, ncs, L - 4, 135, L - 4, 20, L - 4, 16, L - 4]
    rc, s, _ = bytes_shape(lib, pack, proofs[0])
    assert rc == 0 and np.array_equal(s, t)                       # an honest proof has exactly the target's shape
    # T = every field element of the proof: its bytes minus the one-byte Merkle path lengths
    n_paths = 28 * (4 + h["num_arity_rounds"])
    assert lib.qpgpu_proof_target_count(pack.ctypes.data, pack.size) == (len(proofs[0]) - n_paths) // 8


def test_values_follow_the_documented_order(pkg, lib, inner):
    pack, pis, proofs = inner
    T = lib.qpgpu_proof_target_count(pack.ctypes.data, pack.size)
    vals = np.zeros(T, dtype=np.uint64); n = ctypes.c_size_t(); err = ctypes.create_string_buffer(ERR)
    assert lib.qpgpu_proof_target_values(pack.ctypes.data, pack.size, proofs[0], len(proofs[0]), 0, b"leaf proof", vals.ctypes.data, T, ctypes.byref(n), err) == 0
    assert n.value == T
    raw = np.frombuffer(proofs[0][:3 * 16 * 32], dtype="<u8")
    assert vals[:21].tolist() == [int(x) for x in pis]                          # public inputs first
    assert np.array_equal(vals[21:21 + 192], raw)                               # then the three caps, as in the bytes
    h = pkg.pack_header(pack)
    ncs = h["num_selectors"] + h["num_constants"] + 80
    ob = 3 * 16 * 32
    def opening(start_ext, count_ext):
        return np.frombuffer(proofs[0][ob + 16 * start_ext: ob + 16 * (start_ext + count_ext)], dtype="<u8")
    at = 21 + 192
    # zeta batch: constants + sigmas, wires, plonk_zs, partial_products, quotient_polys; plonk_zs_next comes after them
    assert np.array_equal(vals[at:at + 2 * (ncs + 135 + 2)], opening(0, ncs + 135 + 2))
    at += 2 * (ncs + 135 + 2)
    assert np.array_equal(vals[at:at + 2 * 18], opening(ncs + 135 + 4, 18))     # partial_products (bytes: behind zs_next)
    at += 2 * 18
    assert np.array_equal(vals[at:at + 2 * 16], opening(ncs + 135 + 4 + 18, 16))
    at += 2 * 16
    assert np.array_equal(vals[at:at + 4], opening(ncs + 135 + 2, 2))           # plonk_zs_next
    at += 4
    pow_witness = int.from_bytes(proofs[0][-8 * 21 - 8:-8 * 21], "little")
    assert int(vals[at]) == pow_witness                                         # then pow_witness, final polynomial, ...
    assert sorted(vals.tolist()) == sorted(np.frombuffer(bytes(b for b in _strip_path_bytes(pkg, pack, proofs[0])), dtype="<u8").tolist())


def _strip_path_bytes(pkg, pack, proof):
    """the proof's bytes without the one-byte Merkle path lengths"""
    h = pkg.pack_header(pack)
    ncs = h["num_selectors"] + h["num_constants"] + h["num_routed_wires"]
    L = h["degree_bits"] + h["rate_bits"]
    arity = [int(x) for x in pack[18:18 + h["num_arity_rounds"]]]
    pos = 3 * 16 * 32 + 16 * (ncs + 135 + 4 + 18 + 16) + len(arity) * 16 * 32
    out = bytearray(proof[:pos])
    for _ in range(28):
        for w in (ncs, 135, 20, 16):
            out += proof[pos:pos + 8 * w]; pos += 8 * w
            assert proof[pos] == L - 4; pos += 1
            out += proof[pos:pos + 32 * (L - 4)]; pos += 32 * (L - 4)
        lvl = L
        for ab in arity:
            lvl -= ab
            out += proof[pos:pos + 16 * (1 << ab)]; pos += 16 * (1 << ab)
            assert proof[pos] == lvl - 4; pos += 1
            out += proof[pos:pos + 32 * (lvl - 4)]; pos += 32 * (lvl - 4)
    out += proof[pos:]
    return bytes(out)


def test_every_malformed_shape_the_reference_names(pkg, lib, inner):
    """ensure_proof_shape_matches_targets' checks in its order, each with its message (common/utils.rs:333-540)."""
    pack, pis, proofs = inner
    t = target_shape(lib, pack)
    assert ensure(lib, t, t) == (0, "")
    n_caps = int(t[13])
    r0 = 15 + n_caps                        # first query round
    round_len = 1 + 2 * 4 + 1 + 2 * n_caps
    names = ["public inputs", "wires_cap", "plonk_zs_partial_products_cap", "quotient_polys_cap", "openings.constants", "openings.plonk_sigmas",
             "openings.wires", "openings.plonk_zs", "openings.plonk_zs_next", "openings.partial_products", "openings.quotient_polys",
             "openings.lookup_zs", "openings.lookup_zs_next"]
    cases = [(i, nm) for i, nm in enumerate(names)]
    cases += [(14, "opening_proof.commit_phase_merkle_caps[0]"),
              (r0 + 1, "opening_proof.query_round_proofs[0].initial_trees_proof.evals_proofs[0].evals"),
              (r0 + 2, "opening_proof.query_round_proofs[0].initial_trees_proof.evals_proofs[0].siblings"),
              (r0 + 7, "opening_proof.query_round_proofs[0].initial_trees_proof.evals_proofs[3].evals"),
              (r0 + 10, "opening_proof.query_round_proofs[0].steps[0].evals"),
              (r0 + 11, "opening_proof.query_round_proofs[0].steps[0].merkle_proof.siblings"),
              (r0 + 5 * round_len + 4, "opening_proof.query_round_proofs[5].initial_trees_proof.evals_proofs[1].siblings"),
              (t.size - 1, "opening_proof.final_poly")]
    for idx, what in cases:
        p = t.copy(); p[idx] += 1
        rc, msg = ensure(lib, t, p)
        assert rc == -1 and msg == f"leaf proof at slot 3 is malformed: {what} has length {int(p[idx])}, but the circuit expects {int(t[idx])}", (idx, msg)
    # list counts: fewer commit caps, fewer query rounds, fewer initial oracles, fewer steps
    def drop(words, at, n):
        return np.concatenate([words[:at], words[at + n:]])
    p = drop(t, 14, 1); p[13] -= 1
    assert ensure(lib, t, p)[1] == f"leaf proof at slot 3 is malformed: opening_proof.commit_phase_merkle_caps has length {n_caps - 1}, but the circuit expects {n_caps}"
    p = drop(t, r0, round_len); p[r0 - 1] -= 1
    assert ensure(lib, t, p)[1] == "leaf proof at slot 3 is malformed: opening_proof.query_round_proofs has length 27, but the circuit expects 28"
    p = drop(t, r0 + 1, 2); p[r0] -= 1
    assert ensure(lib, t, p)[1] == "leaf proof at slot 3 is malformed: opening_proof.query_round_proofs[0].initial_trees_proof.evals_proofs has length 3, but the circuit expects 4"
    p = drop(t, r0 + 10, 2); p[r0 + 9] -= 1
    assert ensure(lib, t, p)[1] == f"leaf proof at slot 3 is malformed: opening_proof.query_round_proofs[0].steps has length {n_caps - 1}, but the circuit expects {n_caps}"
    assert ensure(lib, t, t[:-1])[0] == -1 and ensure(lib, t[:-1], t)[0] == -1           # descriptors that are not shapes
    assert ensure(lib, t, p, slot=9, label=b"private batch proof")[1].startswith("private batch proof at slot 9 is malformed")


def test_malformed_bytes(pkg, lib, inner):
    pack, pis, proofs = inner
    t = target_shape(lib, pack)
    good = proofs[0]
    # a public input too many: the layout still parses, the shape check names it
    rc, s, _ = bytes_shape(lib, pack, good + bytes(8))
    assert rc == 0 and ensure(lib, t, s)[1] == "leaf proof at slot 3 is malformed: public inputs has length 22, but the circuit expects 21"
    rc, s, _ = bytes_shape(lib, pack, good[:-8])
    assert rc == 0 and int(s[0]) == 20
    # truncated inside a vector, a non-canonical element
    assert bytes_shape(lib, pack, good[:-3])[0] == -1
    bad = bytearray(good); bad[8:16] = (P + 1).to_bytes(8, "little")
    rc, _, msg = bytes_shape(lib, pack, bytes(bad))
    assert rc == -1 and "non-canonical" in msg
    # a Merkle path that claims another length shifts everything behind it: rejected one way or the other
    h = pkg.pack_header(pack)
    ncs = h["num_selectors"] + h["num_constants"] + 80
    at = 3 * 512 + 16 * (ncs + 135 + 4 + 18 + 16) + h["num_arity_rounds"] * 512 + 8 * ncs
    bad = bytearray(good); assert bad[at] == h["degree_bits"] + 3 - 4; bad[at] -= 1
    rc, s, msg = bytes_shape(lib, pack, bytes(bad))
    assert rc == -1 or ensure(lib, t, s)[0] == -1
    n = ctypes.c_size_t(); err = ctypes.create_string_buffer(ERR)
    rc = lib.qpgpu_proof_target_values(pack.ctypes.data, pack.size, bytes(bad), len(bad), 2, b"leaf proof", None, 0, ctypes.byref(n), err)
    assert rc == -1 and err.value.decode().startswith("leaf proof at slot 2 is malformed")


def fill(lib, pack, proofs, n_targets, pre, n_pre_targets, want_out=True):
    arr = (ctypes.c_char_p * len(proofs))(*proofs)
    lens = (ctypes.c_size_t * len(proofs))(*[len(p) for p in proofs])
    pre = np.ascontiguousarray(pre, dtype=np.uint64)
    n = ctypes.c_size_t(); err = ctypes.create_string_buffer(ERR)
    T = lib.qpgpu_proof_target_count(pack.ctypes.data, pack.size)
    cap = len(proofs) * (T + 4)
    ids = np.zeros(cap, dtype=np.uint32); vals = np.zeros(cap, dtype=np.uint64)
    rc = lib.qpgpu_batch_fill_proof_targets(pack.ctypes.data, pack.size, arr, lens, len(proofs), n_targets, pre.ctypes.data, pre.size // 4, n_pre_targets,
                                            b"leaf proof", ids.ctypes.data if want_out else None, vals.ctypes.data if want_out else None, cap, ctypes.byref(n), err)
    return rc, ids[:n.value], vals[:n.value], err.value.decode(), n.value


def test_fill_private_batch_witness(pkg, lib, inner):
    pack, pis, proofs = inner
    T = lib.qpgpu_proof_target_count(pack.ctypes.data, pack.size)
    pre = np.arange(1, 9, dtype=np.uint64).reshape(2, 4)
    rc, ids, vals, msg, n = fill(lib, pack, proofs, 2, pre, 2)
    assert rc == 0 and n == 2 * (T + 4)
    assert ids.tolist() == list(range(2 * T + 8))                   # slot-major proofs, then the preimages
    assert vals[:21].tolist() == [int(x) for x in pis] and vals[T:T + 21].tolist() == [int(x) for x in pis]
    assert vals[2 * T:].tolist() == list(range(1, 9))
    assert fill(lib, pack, proofs, 2, pre, 2, want_out=False)[4] == 2 * (T + 4)
    # the three count checks, with the reference's messages (witness.rs:23-45)
    assert fill(lib, pack, proofs, 3, pre, 3)[3] == "proof count mismatch: got 2, but circuit expects 3 leaf proofs"
    assert fill(lib, pack, proofs, 2, pre, 3)[3] == "target layout is inconsistent: dummy_nullifier_pre_image target count 3 != leaf proof target count 2"
    assert fill(lib, pack, proofs, 2, pre[:1], 2)[3] == "dummy nullifier preimage count mismatch: got 1, but circuit expects 2"
    # a malformed proof in slot 1 is named with its slot; a non-canonical preimage element too
    rc, _, _, msg, _ = fill(lib, pack, [proofs[0], proofs[1] + bytes(8)], 2, pre, 2)
    assert rc == -1 and msg == "leaf proof at slot 1 is malformed: public inputs has length 22, but the circuit expects 21"
    bad = pre.copy(); bad[1, 2] = P
    rc, _, _, msg, _ = fill(lib, pack, proofs, 2, bad, 2)
    assert rc == -1 and msg.startswith("failed to set dummy nullifier preimage target at slot 1, limb 2")

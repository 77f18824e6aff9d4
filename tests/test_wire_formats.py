"""Wire formats (include/qpgpu_wire.h): proof hex, config.json with the legacy key, artifact names, and the pack validator's
diagnostics. Host only. Reference: wormhole/aggregator/src/config.rs:20-88 (and its tests :90-170),
wormhole/tests/src/aggregator/aggregator_tests.rs:350-394, artifact names in the three build.rs files."""
import ctypes

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib(pkg):
    L = pkg.load_library()
    L.qpgpu_hex_encode.restype = ctypes.c_size_t
    L.qpgpu_hex_encode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    L.qpgpu_hex_decode.restype = ctypes.c_size_t
    L.qpgpu_hex_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    L.qpgpu_bins_config_parse.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_char_p]
    L.qpgpu_bins_config_write.restype = ctypes.c_size_t
    L.qpgpu_bins_config_write.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.qpgpu_artifact_name.restype = ctypes.c_char_p
    L.qpgpu_artifact_name.argtypes = [ctypes.c_int, ctypes.c_int]
    L.qpgpu_pack_validate.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p]
    return L


class Cfg(ctypes.Structure):
    _fields_ = [("num_leaf_proofs", ctypes.c_uint64), ("has_priv", ctypes.c_int), ("num_private_batch_proofs", ctypes.c_uint64)]


def parse(lib, text):
    c = Cfg(); err = ctypes.create_string_buffer(200)
    rc = lib.qpgpu_bins_config_parse(text.encode(), len(text.encode()), ctypes.byref(c), err)
    return rc, (c.num_leaf_proofs, c.num_private_batch_proofs if c.has_priv else None), err.value.decode()


def test_proof_hex_round_trip(lib):
    rng = np.random.default_rng(4)
    for n in (0, 1, 2, 31, 133440):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        out = ctypes.create_string_buffer(2 * n + 1)
        assert lib.qpgpu_hex_encode(data, n, out, 2 * n + 1) == 2 * n
        assert out.value.decode() == data.hex()                      # hex::encode: lowercase, no prefix
        back = ctypes.create_string_buffer(max(n, 1))
        assert lib.qpgpu_hex_decode(out.value, 2 * n, back, n) == n and back.raw[:n] == data
        up = out.value.upper()
        assert lib.qpgpu_hex_decode(up, 2 * n, back, n) == n and back.raw[:n] == data   # hex::decode takes either case
    bad = 2**64 - 1
    b = ctypes.create_string_buffer(16)
    assert lib.qpgpu_hex_decode(b"abc", 3, b, 16) == bad             # odd length
    assert lib.qpgpu_hex_decode(b"0x12", 4, b, 16) == bad            # no prefix
    assert lib.qpgpu_hex_decode(b"12 4", 4, b, 16) == bad
    assert lib.qpgpu_hex_encode(b"ab", 2, b, 4) == 0                 # no room for the terminator


def test_config_json(lib):
    # what CircuitBinsConfig::save writes (serde_json::to_string_pretty), and back
    for leaf, priv in ((8, 8), (7, None), (64, 1), (1, 64)):
        c = Cfg(leaf, 0 if priv is None else 1, priv or 0)
        out = ctypes.create_string_buffer(200)
        n = lib.qpgpu_bins_config_write(ctypes.byref(c), out, 200)
        want = '{\n  "num_leaf_proofs": %d,\n  "num_private_batch_proofs": %s\n}' % (leaf, "null" if priv is None else priv)
        assert n == len(want) and out.value.decode() == want
        assert parse(lib, want) == (0, (leaf, priv), "")
    # the legacy key of older config.json files (serde alias), compact spacing, unknown keys ignored, missing Option = None
    assert parse(lib, '{"num_leaf_proofs":8,"num_layer0_proofs":4}')[:2] == (0, (8, 4))
    assert parse(lib, ' { "extra" : {"a":[1,2,{"b":"}"}]}, "num_leaf_proofs" : 16 , "flag": true } ')[:2] == (0, (16, None))
    assert parse(lib, '{"num_private_batch_proofs": null, "num_leaf_proofs": 2}')[:2] == (0, (2, None))
    # rejections: the reference's validate() bounds (1..=64) and serde's type / duplicate / missing-field errors
    for text, why in (('{"num_leaf_proofs": 0}', "must be > 0"), ('{"num_leaf_proofs": 65}', "exceeds maximum allowed (64)"),
                      ('{"num_leaf_proofs": 8, "num_private_batch_proofs": 0}', "num_private_batch_proofs must be > 0"),
                      ('{"num_leaf_proofs": 8, "num_private_batch_proofs": 1025}', "exceeds maximum"),
                      ('{"num_private_batch_proofs": 8}', "missing field"), ('{"num_leaf_proofs": "8"}', "invalid type"),
                      ('{"num_leaf_proofs": 8, "num_private_batch_proofs": 2, "num_layer0_proofs": 2}', "duplicate field"),
                      ('{"num_leaf_proofs": 8.0}', "bad value"), ('{"num_leaf_proofs": 8} x', "trailing"), ('[8]', "expected an object"),
                      ('{"num_leaf_proofs": -1}', "invalid type"), ('{"num_leaf_proofs": 08}', "bad value")):
        rc, _, err = parse(lib, text)
        assert rc != 0 and why in err, (text, err)
    assert lib.qpgpu_bins_config_write(ctypes.byref(Cfg(0, 0, 0)), ctypes.create_string_buffer(200), 200) == 0


def test_artifact_names(lib):
    n = lambda l, k: (lib.qpgpu_artifact_name(l, k) or b"").decode()
    assert [n(0, k) for k in range(3)] == ["common.bin", "verifier.bin", "dummy_proof.bin"]
    assert [n(1, k) for k in range(3)] == ["private_batch_common.bin", "private_batch_verifier.bin", "dummy_private_batch_proof.bin"]
    assert [n(2, k) for k in range(3)] == ["public_batch_common.bin", "public_batch_verifier.bin", ""]
    assert [n(l, 3) for l in range(3)] == ["prover_pack.qpcp", "private_batch_prover_pack.qpcp", "public_batch_prover_pack.qpcp"]
    assert n(1, 4) == "config.json" and n(5, 0) == "" and n(0, 9) == ""


def test_pack_validator_explains_refusals(lib, pkg):
    pack, wires, pis = pkg.synth_circuit(6, num_wires=135, num_routed=80, num_public_inputs=5, seed=21, poseidon=True, base_sum=True, hints=True)
    hdr = pkg.pack_header(pack)
    n = 1 << hdr["degree_bits"]
    base = 18 + hdr["num_arity_rounds"] + 8 * hdr["num_gates"]
    cs0 = base + hdr["num_routed_wires"] + 4
    sig0 = cs0 + (hdr["num_selectors"] + hdr["num_constants"]) * n

    def check(p):
        err = ctypes.create_string_buffer(200)
        p = np.ascontiguousarray(p, dtype=np.uint64)
        return lib.qpgpu_pack_validate(p.ctypes.data, p.size, err), err.value.decode()

    assert check(pack) == (0, "")
    cases = []
    b = pack.copy(); b[0] ^= 1; cases.append((b, "bad magic"))
    cases.append((pack[:cs0 + 10], "truncated constants_sigmas"))
    b = pack.copy(); b[10] = 99; cases.append((b, "rate_bits"))
    b = pack.copy(); b[12] = 70; cases.append((b, "proof_of_work_bits"))
    b = pack.copy(); b[18 + hdr["num_arity_rounds"]] = 77; cases.append((b, "unknown gate type"))
    b = pack.copy(); b[18 + hdr["num_arity_rounds"] + 8 * 1 + 5] += 1; cases.append((b, "group"))            # a group end moved
    b = pack.copy(); b[cs0 + 3] = 55; cases.append((b, "names no gate"))                                           # selector value of row 3
    b = pack.copy(); b[sig0 + 7] = b[sig0 + 8]; cases.append((b, "two cells map to"))                              # sigma collision
    b = pack.copy(); b[sig0 + 5] = 12345; cases.append((b, "outside every wire coset"))
    b = pack.copy(); b[base + 1] = b[base]; cases.append((b, "same coset"))                                        # k_is[1] = k_is[0]
    b = pack.copy(); b[17] = hdr["num_arity_rounds"]; b[18] = 5; cases.append((b, "unsupported FRI arity"))
    b = pack.copy(); b[-1] = np.uint64(135 * n + 3); cases.append((b, "public-input cell is not a routed wire"))
    b = pack.copy(); b[-(2 + 5) - 8 + 1] = np.uint64(135 * n); cases.append((b, "hint cell is not a routed wire"))
    for p, why in cases:
        rc, err = check(p)
        assert rc != 0 and why in err, (why, err)
    # a cap taller than the last FRI tree
    p2, _, _ = pkg.synth_circuit(5, num_wires=24, num_routed=16, num_public_inputs=0, seed=2)
    h2 = pkg.pack_header(p2)
    if h2["num_arity_rounds"]:
        b = p2.copy(); b[11] = 7
        rc, err = check(b)
        assert rc != 0 and ("cap height" in err or "cap_height" in err), err


# ---- CircuitConfig policy (common/src/circuit.rs:378-571, tests :589-675) ----

class CircuitConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("num_wires", "num_routed_wires", "num_constants", "security_bits", "num_challenges", "max_quotient_degree_factor")] + \
               [("use_base_arithmetic_gate", ctypes.c_int), ("zero_knowledge", ctypes.c_int)] + \
               [(n, ctypes.c_uint64) for n in ("rate_bits", "cap_height", "proof_of_work_bits", "num_query_rounds", "reduction_arity_bits", "reduction_final_poly_bits")]


def _lib():
    import __graft_entry__ as ge
    return ge.load_package().load_library()


def _config(level):
    lib = _lib()
    c = CircuitConfig()
    assert lib.qpgpu_wormhole_circuit_config(level, ctypes.byref(c)) == 0
    return c


def _validate(c):
    err = ctypes.create_string_buffer(400)
    rc = _lib().qpgpu_validate_circuit_config(ctypes.byref(c), err)
    return rc, err.value.decode()


def test_canonical_wormhole_configs_pass():
    for level in (0, 1, 2):
        assert _validate(_config(level))[0] == 0
    leaf, priv, pub = _config(0), _config(1), _config(2)
    assert (leaf.num_wires, leaf.num_routed_wires, leaf.zero_knowledge, leaf.rate_bits, leaf.cap_height, leaf.num_query_rounds) == (135, 80, 0, 3, 4, 28)
    assert (priv.num_wires, priv.num_routed_wires, priv.zero_knowledge) == (135, 60, 1)
    assert (pub.num_routed_wires, pub.zero_knowledge) == (80, 0)


def test_structural_floors_are_enforced():
    for field, value, needle in (("num_wires", 134, "num_wires"), ("num_routed_wires", 36, "num_routed_wires"), ("num_routed_wires", 136, "prefix"),
                                 ("max_quotient_degree_factor", 6, "max_quotient_degree_factor")):
        c = _config(1); setattr(c, field, value)
        rc, msg = _validate(c)
        assert rc != 0 and needle in msg, msg


def test_fri_exponent_ceilings_are_enforced():
    for bits in (9, 20, 63, (1 << 64) - 1):
        c = _config(1); c.rate_bits = bits
        assert "rate_bits" in _validate(c)[1]
        c = _config(1); c.cap_height = bits
        assert "cap_height" in _validate(c)[1]
    c = _config(1); c.rate_bits = 8; c.cap_height = 8
    assert _validate(c)[0] == 0


def test_rate_bits_below_quotient_degree_are_rejected():
    for bits in (1, 2):
        c = _config(1); c.rate_bits = bits
        rc, msg = _validate(c)
        assert rc != 0 and "proving time" in msg


def test_zero_knobs_are_rejected():
    for field, needle in (("num_challenges", "num_challenges"), ("num_query_rounds", "num_query_rounds"), ("security_bits", "security_bits")):
        c = _config(1); setattr(c, field, 0)
        rc, msg = _validate(c)
        assert rc != 0 and needle in msg


def test_pack_config_is_canonical(pkg):
    lib = _lib()
    lib.qpgpu_pack_config_is_canonical.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p]
    err = ctypes.create_string_buffer(400)
    leaf, _, _ = pkg.synth_circuit(6, num_wires=135, num_routed=80, num_public_inputs=21, seed=1, poseidon=True)
    assert lib.qpgpu_pack_config_is_canonical(leaf.ctypes.data, leaf.size, 0, err) == 0, err.value
    assert lib.qpgpu_pack_config_is_canonical(leaf.ctypes.data, leaf.size, 2, err) == 0
    assert lib.qpgpu_pack_config_is_canonical(leaf.ctypes.data, leaf.size, 1, err) != 0 and b"private-batch circuit config does not match" in err.value
    priv, _, _ = pkg.synth_circuit(6, num_wires=135, num_routed=60, num_public_inputs=21, seed=1, poseidon=True)
    priv[14] = 1
    assert lib.qpgpu_pack_config_is_canonical(priv.ctypes.data, priv.size, 1, err) == 0, err.value
    priv[13] = 20
    assert lib.qpgpu_pack_config_is_canonical(priv.ctypes.data, priv.size, 1, err) != 0 and b"num_query_rounds loaded=20, expected=28" in err.value

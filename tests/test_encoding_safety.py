"""The reference's encoding-safety checks (wormhole/tests/tests/encoding_safety.rs, the differential side of
formal/WormholeSpec/Encoding.lean) against this library's codecs (include/qpgpu_leaf.h): the 4-bytes-per-element edge encoding is
lossless and injective unconditionally and never needs a field reduction; the 8-bytes-per-element digest encoding round-trips and
is injective on canonical limbs only — the {w, w + p} collision that makes the canonical-input precondition load-bearing is
exhibited, as the reference exhibits it."""
import ctypes

import numpy as np
import pytest
from hypothesis import assume, given, settings, strategies as st

P = 0xFFFFFFFF00000001
limb = st.integers(min_value=0, max_value=P - 1)
digest = st.tuples(limb, limb, limb, limb)
CASES = settings(max_examples=80, deadline=None)


@pytest.fixture(scope="module")
def lib(pkg):
    L = pkg.load_library()
    c = ctypes
    L.qpgpu_bytes_to_felts.argtypes = [c.c_char_p, c.c_size_t, c.c_void_p, c.c_size_t]; L.qpgpu_bytes_to_felts.restype = c.c_size_t
    L.qpgpu_felts_to_bytes.argtypes = [c.c_void_p, c.c_size_t, c.c_char_p, c.c_size_t]; L.qpgpu_felts_to_bytes.restype = c.c_size_t
    L.qpgpu_bytes_to_digest.argtypes = [c.c_char_p, c.c_void_p]; L.qpgpu_bytes_to_digest.restype = None
    L.qpgpu_digest_to_bytes.argtypes = [c.c_void_p, c.c_char_p]; L.qpgpu_digest_to_bytes.restype = None
    L.qpgpu_bytes_digest_is_canonical.argtypes = [c.c_char_p]
    return L


def b32(limbs):
    return b"".join(int(v).to_bytes(8, "little") for v in limbs)


def to_felts(lib, data):
    out = np.zeros(len(data) // 4 + 1, dtype=np.uint64)
    assert lib.qpgpu_bytes_to_felts(bytes(data), len(data), out.ctypes.data, out.size) == out.size
    return out


def to_digest(lib, b):
    out = np.zeros(4, dtype=np.uint64)
    lib.qpgpu_bytes_to_digest(b, out.ctypes.data)
    return out.tolist()


def from_digest(lib, felts):
    x = np.array(felts, dtype=np.uint64); out = ctypes.create_string_buffer(32)
    lib.qpgpu_digest_to_bytes(x.ctypes.data, out)
    return out.raw


def test_digest_decode_collides_and_round_trip_fails_off_canonical(lib):
    canonical, non_canonical = b32([0, 0, 0, 0]), b32([P, 0, 0, 0])
    assert canonical != non_canonical and to_digest(lib, canonical) == to_digest(lib, non_canonical)      # limbs 0 and p decode to the same digest
    assert from_digest(lib, to_digest(lib, non_canonical)) == canonical != non_canonical                  # folded to the canonical representative
    assert lib.qpgpu_bytes_digest_is_canonical(canonical) == 1 and lib.qpgpu_bytes_digest_is_canonical(non_canonical) == 0      # BytesDigest::try_from refuses it


@CASES
@given(st.binary(max_size=95))
def test_edge_encoding_round_trips_with_32_bit_limbs(lib, data):
    felts = to_felts(lib, data)
    assert int(felts.max()) < 1 << 32                                          # no field reduction at the edges
    out = ctypes.create_string_buffer(128)
    n = lib.qpgpu_felts_to_bytes(felts.ctypes.data, felts.size, out, 128)
    assert n == len(data) and out.raw[:n] == data


@CASES
@given(st.binary(max_size=95), st.binary(max_size=95))
def test_edge_encoding_injective(lib, x, y):
    assume(x != y)
    assert to_felts(lib, x).tolist() != to_felts(lib, y).tolist()


@CASES
@given(digest)
def test_digest_round_trips_on_canonical(lib, d):
    assert from_digest(lib, to_digest(lib, b32(d))) == b32(d)                  # digest_round_trips_on_canonical
    assert to_digest(lib, from_digest(lib, d)) == list(d)                      # hash_output_digest_round_trips


@CASES
@given(digest, digest)
def test_digest_decode_injective_on_canonical(lib, x, y):
    assume(x != y)
    assert to_digest(lib, b32(x)) != to_digest(lib, b32(y))

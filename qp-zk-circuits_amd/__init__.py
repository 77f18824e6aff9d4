"""qp-zk-circuits_amd — MI355X (gfx950) backend for the qp-wormhole proving hot path.

The product is libqpgpu.so (hand-written HIP kernels behind the C ABI of include/qpgpu.h).
This Python package is only the ctypes harness used by tests/ and bench.py; it never falls back
to a CPU implementation: if the shared library or a GPU is missing, calls raise.

The directory name carries a hyphen (it mirrors the reference repository's name), so it is
imported through `__graft_entry__.load_package()` under the module name `qp_zk_circuits_amd`.
"""
from .binding import poseidon_constants, synth_circuit, Circuit, QpGpu, QpGpuError, lib_path, load_library, P, MULT_GEN  # noqa: F401
from .binding import PolyOracle, Challenger, fri_prove, set_hasher_poseidon, set_hasher_poseidon2, ProvingPool  # noqa: F401
from .binding import pack_header, pack_public_input_cells, Verifier, poseidon2_qp_params, synth_p2_sites  # noqa: F401
from .binding import pack_p2_layout, pack_trailers, p2_site_cells, P2_NO_SWAP  # noqa: F401
from . import sharding  # noqa: F401,E402
from . import aggregation  # noqa: F401,E402
from . import leaf  # noqa: F401,E402
from . import recursion  # noqa: F401,E402

"""Short runs of the four fuzz tools as part of the GPU suite (the long runs are recorded in profiles/r04_soak.txt): random and
mutated CircuitInputs through the leaf path (host constraint check, device s1 and oracle agree; witnesses equal), single-bit flips
of a leaf proof through the complete in-circuit verifier (verdict equals the host verifier's), random private batches over fake
leaves (a witness exists iff the host restatement accepts, and the public inputs read out of the witness equal the host's), and the same for
random public batches over stand-in private-batch proofs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,args,keys", [
    ("tests/soak/fuzz_leaf_inputs.py", ["384", "21"], {"inputs": 384}),              # (checks against the oracle: lives under tests/)
    ("tests/soak/fuzz_leaf_inputs.py", ["256", "25", "hints"], {"inputs": 256, "hash_hints": True}),   # the device side with the front-end's hash hints
    ("tools/fuzz_wrapper_tamper.py", ["96", "22"], {"flips": 96, "accepted_by_both": 0}),
    ("tools/fuzz_private_batch.py", ["192", "23"], {"batches": 192}),
    ("tools/fuzz_public_batch.py", ["192", "24"], {"batches": 192}),
])
def test_fuzz_tool(tool, args, keys):
    res = subprocess.run([sys.executable, os.path.join(ROOT, tool)] + args, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    stats = json.loads(res.stdout.strip().splitlines()[-1])
    assert stats["mismatches"] == 0
    for k, v in keys.items():
        assert stats[k] == v
    if tool.endswith(("fuzz_private_batch.py", "fuzz_public_batch.py")):
        assert stats["satisfiable"] > 50 and stats["unsatisfiable"] > 20
    if tool.endswith("fuzz_leaf_inputs.py"):
        assert stats["satisfiable"] > stats["inputs"] // 3 and stats["unsatisfiable"] > stats["inputs"] // 8 and stats["witnesses_compared"] == stats["satisfiable"]


def test_no_dead_witness_cells():
    """tools/dead_cell_lint.py, a short pass: every advice cell a generator writes — in a synthetic circuit with all 15 gate types, the
    leaf circuit, the gadget circuits, two random gadget programs and a private-batch wrapper — changes the verdict of the witness
    check when it is changed: no gate here constrains less than its generator assumes."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dead_cell_lint.py"), "500"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.strip().splitlines()[-1] == "undetected total: 0"

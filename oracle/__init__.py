"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's algorithm for the hot path named in
BASELINE.json (YOLOv2 conv fwd/bwd + the src/pruning masks).  It is the
*checker* for the HIP path and the `cpu_baseline` leg of bench.py.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this package.  Nothing under `modelcompression_amd/` imports it, and the
product raises when its HIP library is missing instead of falling back here.

Parity pin: every function in here is checked against the reference itself
(imported in the build container by tests/golden/gen_golden.py) and against
the golden fixtures that script wrote to tests/golden/.  The reference has no
tests or golden vectors of its own for this path, so the
fixtures generated from the imported reference are the pin.  Version pin:
numpy 2.2.6 / torch 2.10 semantics (np.percentile computes the virtual index in
the array's dtype, see prune_ref.percentile_linear).
"""

"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU fp32 restatement of the reference's hot-path arithmetic, written from the
reference's text (each function cites the file:line it follows) and pinned against the reference
modules themselves by `oracle/gen_golden.py` (fixtures under `tests/golden/`).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
package, and only as the checker / reported baseline.  The product (`computervision_codes_amd`)
never imports it and fails loudly when its HIP library is missing.
"""

"""oracle/ -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package, and only as the checker / reported baseline.  The product path
(`voice-tts_amd/`) never imports it and fails loudly when the HIP library is missing.

Every function cites the reference file:line it restates (paths relative to the
reference repo root).  Arithmetic is fp32 on the host CPU (torch CPU tensors for
matmul/conv so the baseline can use all host cores).

Pinning status: the reference ships NO tests or golden vectors for this path
(SURVEY.md F12), so the oracle is pinned against outputs of the reference's own
module classes, imported in the build container with seeded random weights
(tests/golden/make_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).
"""

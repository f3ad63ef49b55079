"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements (plain torch-CPU functional ops / numpy) of the reference algorithms on the
hot path, each citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; the product (torch_em_amd/) never
does and has no CPU fallback.

Pinning: tests/golden/*.npz were generated in the build container by IMPORTING the reference
itself (tests/golden/gen_golden.py reads /root/reference) -- model outputs, losses and all
parameter gradients for small configurations.  tests/test_oracle_golden.py checks every
function here against those vectors, so the oracle is pinned to the reference's behaviour.
Exceptions (stated in DESIGN.md): BoundaryTransform has no reference test and its arithmetic
lives in un-vendored scikit-image -> "parity unpinned" for label_ref.boundaries (hand-made KATs
only); affinities are pinned by the brute-force definitions the reference's own test holds.
"""

"""CPU oracle for the Semantic-Pyramid hot path.  TEST INFRASTRUCTURE ONLY.

A restatement, in plain torch-CPU functional code driven by state_dicts, of what the reference
computes on this path (generator / discriminator / feature-creation models / ResNet-18 taps /
WGAN-GP train steps / masks).  Each function cites the reference file:line it follows.

Pinning: `oracle/gen_golden.py` imports the reference's own modules from /root/reference in the build
container and writes small fixtures under tests/golden/; `tests/test_oracle_golden.py` checks this
restatement against them (CPU, no GPU needed).  Gaps: the ResNet-18 arithmetic lives in
torchvision (un-vendored, not installed) and no trained classifier weights ship, so the classifier is
"parity unpinned" by reference data — it is pinned op-by-op against torch functional ops instead.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product package never does (tests/test_no_oracle_in_product.py enforces it).
"""

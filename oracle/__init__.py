"""CPU oracle for the DPO/PPO hot path -- TEST INFRASTRUCTURE ONLY.

A plain-PyTorch (fp32, CPU) restatement of the reference's algorithm for the path named in
BASELINE.json, each function citing the reference file:line it follows.  Only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package, and only as the
checker / the reported CPU baseline -- never as the thing measured or shipped.  The product path
(align_anything_amd/) does not import it and has no CPU fallback.

Pinning: the reference ships no golden vectors ("parity unpinned" upstream), so the
oracle is pinned against outputs of the reference ITSELF, run in the build container through
oracle/_shim.py by oracle/gen_golden.py; the resulting fixtures are committed under tests/golden/ and
checked by tests/test_oracle_golden.py (CPU).  The transformer arithmetic lives in the third-party
HuggingFace `transformers` package (installed: 5.15.0; reference pins >= 4.50.0, pyproject.toml:37);
oracle/models.py restates it and is pinned the same way against HF modules run on CPU.
"""

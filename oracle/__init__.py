"""ORACLE — test infrastructure, not product code.

CPU restatement (torch-CPU / numpy ops, written from scratch) of the reference's
diffusion hot path (gms/diffusion/{simple_unet,gaussian_diffusion,diffusion_utils}.py).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this package, and only as the checker / the timed CPU baseline.  The product package
`generative_models_amd` never imports it (tests/test_no_oracle_in_product.py enforces that).

Pinning: the reference's own tests hold no numeric fixtures for this path
(tests/test_models.py:10-14 is exit-status only), so the oracle is pinned by golden
vectors generated in the build container by importing the reference itself
(`oracle/make_golden.py`, outputs committed under `tests/golden/`), plus the
known-answer anchors of SURVEY.md Appendix C.
"""

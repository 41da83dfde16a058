"""CPU oracle for the REFace DDIM inference hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Plain-PyTorch fp32 restatement of the reference algorithm (UNet, KL-VAE, CLIP vision + mapper,
ArcFace IR-SE50, DDIM/CFG), written as pure functions over reference-layout state dicts.  Every
function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and there only as the checker / baseline, never as the thing measured or shipped.
Nothing under ``reface_amd/`` imports it.

Parity pinning: the reference has no tests or golden vectors of its own, so
the oracle is pinned against outputs of the reference itself, imported in the build container by
``tools/gen_golden.py`` and committed as fixtures under ``tests/golden/`` (checked by
``tests/test_oracle_golden.py``).  Third-party arithmetic that is not under /root/reference
(HF ``transformers`` CLIP, pinned 4.19.2 by the reference; here 5.x) is pinned against the
installed transformers version by the same script.
"""

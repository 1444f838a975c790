"""CPU oracle for the k-diffusion sampling hot path  --  TEST INFRASTRUCTURE ONLY.

Nothing in the product path (the ``k-diffusion_amd`` package, ``sample.py``) imports this
package.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may use it, and only as the checker / the timed CPU baseline.

Contents
--------
``hdit.py``       functional torch-CPU restatement of ``ImageTransformerDenoiserModelV2.forward``
                  (/root/reference/k_diffusion/models/image_transformer_v2.py:721-762) and every op
                  below it, including a restated NATTEN ``na2d`` (third-party, absent: see below).
``solvers.py``    restatement of the Karras schedule, ``to_d``, ancestral step, the Karras
                  preconditioner and the sampler loops (sampling.py:17-607, layers.py:70-90).
``brownian.py``   CPU restatement of the counter-based Brownian-interval noise source used by the
                  HIP ``BrownianTreeNoiseSampler`` (the reference delegates to torchsde, absent).
``ref_import.py`` import shim that loads the *real* reference from /root/reference with stubbed
                  third-party modules (works only in the build container).
``make_golden.py`` regenerates ``tests/golden/*.safetensors`` from the real reference.

Parity pinning
--------------
The reference has no tests, fixtures or golden vectors of its own, so the
oracle is pinned against *outputs of the reference itself run in the build container*:
``tests/golden/`` holds those outputs together with the generator script, and
``tests/test_oracle_vs_golden.py`` checks every oracle function against them.

Two pieces are **parity unpinned** because their third-party implementation is not in
/root/reference nor installable: NATTEN ``na2d`` (un-pinned "NATTEN main", README.md:15) and
``torchsde.BrownianTree`` (requirements.txt:14, un-pinned).  For those the oracle *defines* the
semantics (clamped 7x7 window; path-consistent Brownian increments) and the tests check the
properties the reference's call sites rely on.
"""

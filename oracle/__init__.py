"""CPU oracle for the e-OSVOS inner-loop hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain torch-CPU (fp32) restatement of the reference algorithm
for the path named by BASELINE.json:north_star.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the *checker* (or as the timed CPU baseline) -- never as the
product path.  The product (``e-osvos_amd``) never imports this package and fails
loudly when its HIP extension is missing.

Parity status: **pinned by generated fixtures**.  The reference repository has no
tests or golden vectors for this path, so the oracle is
pinned against outputs of the reference itself: ``tests/golden/make_golden.py``
imports the *unmodified* reference classes from /root/reference/src (through a
torchvision stand-in, because torchvision 0.4 is a third-party dependency that is
neither vendored in the reference nor installed in this image) and writes small
input/output fixtures to ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this oracle against them.

Third-party arithmetic restated here (not under /root/reference):
torchvision 0.4 (README.md:23 of the reference) ``models.resnet.resnet50/101``
(Bottleneck v1.5), ``models.segmentation.deeplabv3.ASPP`` and
``models._utils.IntermediateLayerGetter`` -- restated from their published
architecture in ``oracle/topology.py`` / ``oracle/deeplab.py``.
"""

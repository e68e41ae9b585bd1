"""CPU oracle for the TRU-Net hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, with stock CPU ``torch`` / ``numpy`` calls, the
algorithm of the reference hot path (Okrio/tinyrecurrentunet: ``network.py``,
``dataset.ProcessAudio``, ``phm.py``, ``stft_loss.py``, ``util.loss_fn``) in the
canonical repaired composition "R" of SURVEY.md section 0.2.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it -- and there only as the checker / the timed CPU baseline,
never as the thing shipped.  Nothing under ``tinyrecurrentunet_amd/`` imports
this package; the product path fails loudly when the HIP extension is missing.

Pinning: the reference repo holds no tests, fixtures or golden vectors
(SURVEY.md section 4).  The oracle is pinned instead by outputs of the
reference's own importable pieces run in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``): the six block
classes of ``network.py:9-120``, ``stft_loss.py`` (whole), ``dataset.ProcessAudio``
/ ``pcenfunc`` and ``util.LinearWarmupCosineDecay``.  Pieces the reference leaves
broken (``TRUNet`` composition D1-D6, ``PhaseAwareMask`` D8, ``loss_fn`` D9) are
"parity unpinned by the reference"; for those the oracle *is* the definition
(repairs R1-R7), checked stage-wise against the importable pieces.
"""

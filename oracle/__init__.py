"""CPU oracle for the CTC acoustic-model training path -- TEST INFRASTRUCTURE, not product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The shipped path (``chainer-speech-recognition_amd/``) never imports it and fails loudly
when the HIP library is missing.

What pins each restatement (see DESIGN.md section "Oracle"):

* ``oracle.ctc``   Gram-CTC and standard CTC: pinned by golden vectors generated from the reference's
                   own ``asr/loss/gram_ctc.py`` (tests/golden/gram_ctc.npz) and cross-checked with
                   ``torch.nn.functional.ctc_loss`` on CPU.
* ``oracle.fft``   filterbank / log-mel / deltas / running statistics / augmentation: pinned by
                   tests/golden/fft.npz, stats.npz, augment.npz (reference ``asr/fft.py``,
                   ``asr/data/loaders/base.py``).  pre-emphasis / framing / power spectrum live in the
                   absent third-party ``python_speech_features.sigproc`` -> PARITY UNPINNED for those
                   three (restated from the published algorithm, cross-checked with ``numpy.fft``).
* ``oracle.nn``    SRU forward pinned by tests/golden/sru.npz (reference ``forward_cpu``); SRU backward,
                   layer-norm backward and weight-norm pinned by finite differences of the pinned
                   forward (the reference's own test strategy, ``asr/nn/test_layernorm.py:70-74``).
                   conv / max-pool / maxout / GRU / Adam are Chainer's (absent) -> PARITY UNPINNED,
                   torch-CPU fp32 is the stated stand-in.
* ``oracle.text``  tokeniser / greedy decode / CER: pinned by tests/golden/text.json.
"""

"""The C-ABI library loads and exports exactly what include/asr_hip.h declares (no GPU, no compute)."""
import os
import re

import pytest

from conftest import ROOT, PKG


def _header_functions():
    text = open(os.path.join(ROOT, "include", "asr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(asr_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built():
    assert os.path.isfile(os.path.join(PKG, "libasr_hip.so")), "run `make -C chainer-speech-recognition_amd`"


def test_header_and_binding_agree():
    from asr import _lib
    assert _header_functions() == sorted(_lib.SIGNATURES.keys())


def test_every_declared_symbol_is_exported():
    from asr import _lib
    handle = _lib.lib()
    for name in _header_functions():
        assert hasattr(handle, name), name
    assert handle.asr_version() >= 1


def test_workspace_query_runs_on_host():
    from asr import _lib
    n = _lib.lib().asr_ctc_workspace_bytes(1000, 32, 3000, 120, 0)
    assert n > 32 * 1000 * 241 * 8 * 2


def test_no_cpu_fallback():
    import torch
    from asr import _lib
    from asr.loss import connectionist_temporal_classification
    x = torch.zeros(4, 2, 5)
    t = torch.ones(2, 1, dtype=torch.int32)
    with pytest.raises(_lib.AsrHipError):
        connectionist_temporal_classification(x, t, 0)

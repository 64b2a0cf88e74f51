"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/echo_hip.h declares, and its
argument validation works without touching a GPU.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "echo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(echo_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from dsp_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_symbols()


def test_library_exports_every_declared_symbol():
    from dsp_amd import _lib
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.echo_version()
    assert lib.echo_strerror(0) == b"ok" and b"invalid" in lib.echo_strerror(-1)


def test_argument_validation_needs_no_gpu():
    from dsp_amd import _lib
    lib = _lib.load()
    assert lib.echo_create(None, None) == -1
    assert lib.echo_reserve(None, 10) == -1
    assert lib.echo_argmax(None, None, 1, 1, None) == -1
    assert lib.echo_chain(None, None, None, None, None, None, None, 1, 1, None) == -1


def test_struct_layout_matches_header():
    from dsp_amd import _lib
    from dsp_amd.engine import RESULT_DTYPE
    assert ctypes.sizeof(_lib.EchoConfigC) == 12 * 4 + 4 * 8 + 3 * 8 + 5 * 8
    assert RESULT_DTYPE.itemsize == 64
    assert RESULT_DTYPE.fields["peak_idx"][1] == 36 and RESULT_DTYPE.fields["valid_mask"][1] == 60


def test_missing_library_fails_loudly(monkeypatch):
    from dsp_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libechohip.so")
    with pytest.raises(_lib.EchoError):
        _lib.load()


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dsp_amd import _lib
    from dsp_amd.engine import EchoEngine
    with pytest.raises(_lib.EchoError):
        EchoEngine()

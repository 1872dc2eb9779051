"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every symbol
include/hawk.h declares.  No compute calls (no GPU here)."""
import os
import re

import pytest

from crisprhawk_hip import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hawk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hawk_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/hawk.h but not exported"
    assert sorted(_lib.EXPORTS) == syms


def test_status_strings_and_no_device_is_loud():
    assert _lib.strerror(0) == "ok"
    assert "IUPAC" in _lib.strerror(_lib.HAWK_E_IUPAC)
    if _lib.device_count() == 0:
        with pytest.raises(_lib.HawkDeviceError):
            _lib.context(0)

"""CPU: librvcx.so loads and exports every symbol include/rvcx.h declares (no compute without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "rvcx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rvcx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    import polgen_rvc_amd  # noqa: F401
    from polgen_rvc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    assert b"gfx950" in lib.rvcx_version()


def test_no_gpu_means_loud_failure():
    """Without a GPU the product path must raise, never fall back to a CPU implementation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import polgen_rvc_amd  # noqa: F401
    from polgen_rvc_amd import _lib
    with pytest.raises(_lib.RvcxError):
        _lib.Context(0)


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "polgen-rvc_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f

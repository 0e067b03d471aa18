"""CPU tier: the C-ABI library builds for gfx950, loads without a GPU and exports every declared symbol."""
import os
import re

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from fastx_toolkit_amd import load_library
    from fastx_toolkit_amd.engine import EXPORTS
    hdr = open(os.path.join(ROOT, "include", "fxg.h")).read()
    declared = sorted(set(re.findall(r"\b(fxg_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = load_library()
    for name in declared:
        assert hasattr(lib, name), "libfxg.so does not export %s" % name
    assert sorted(EXPORTS) == declared
    assert lib.fxg_abi_version() == int(re.search(r"#define FXG_ABI_VERSION (\d+)", hdr).group(1))


def test_no_cpu_fallback():
    import torch
    from fastx_toolkit_amd import Engine, FxgError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(FxgError):
        Engine(0)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "fastx_toolkit_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                src = open(os.path.join(d, f), errors="replace").read()
                assert "fxoracle" not in src and "oracle/" not in src and "fxg_emu" not in src, os.path.join(d, f)

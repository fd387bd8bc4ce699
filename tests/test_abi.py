"""C-ABI surface: the shared library builds, loads and exports exactly what include/mmada_mi355x.h declares."""
import os
import re

from helpers import ROOT
from mmada_parallel_amd import abi, build


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mmada_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmada_[a-z_0-9]+)\s*\(", src)))


def test_library_builds_and_loads():
    path = build.build()
    assert os.path.exists(path)
    assert abi.lib().mmada_abi_version() == 1


def test_every_declared_symbol_is_exported_and_bound():
    syms = header_symbols()
    assert len(syms) >= 20
    lib = abi.lib()
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    assert sorted(abi.SIGNATURES) == syms, "abi.py and the header disagree on the symbol list"


def test_argument_errors_are_reported_without_a_gpu():
    lib = abi.lib()
    assert lib.mmada_create(None, None, None) != 0
    assert b"null" in lib.mmada_last_error()
    assert lib.mmada_workspace_bytes(None, 1, 16) == 0


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mmada_parallel_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt, f

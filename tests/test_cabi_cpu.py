"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/sumfact.h declares, with no compute call (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    lib = os.path.join(ROOT, "gpu-benchmarking_amd", "lib", "libsumfact.so")
    if not os.path.exists(lib):
        ge.build()
    return ge.load_package()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sumfact.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(pkg):
    lib = pkg.capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sumfact.h but not exported"
    # and the Python binding covers exactly the header
    assert sorted(pkg.capi.SYMBOLS) == declared


def test_version_and_strings(pkg):
    lib = pkg.capi.lib()
    assert lib.sf_version() == 100
    assert lib.sf_error_string(0) == b"success"
    assert b"invalid" in lib.sf_error_string(-1)
    names = [lib.sf_variant_name(i).decode() for i in range(9)]
    assert names == list(pkg.VARIANTS)


def test_argument_validation_without_gpu(pkg):
    """Validation happens before any HIP call, so it is testable on a CPU-only box."""
    lib = pkg.capi.lib()
    assert lib.sf_bwdtrans_hex_f64(1, 8, 8, 10, None, None, None, None, None, None) == -1
    assert lib.sf_bwdtrans_hex_f64(8, 8, 8, 0, None, None, None, None, None, None) == 0
    assert lib.sf_bwdtrans_hex_f64(8, 8, 8, 4, None, None, None, None, None, None) == -1
    assert lib.sf_bwdtrans_quad_f64(8, 1, 10, None, None, None, None, None) == -1
    assert lib.sf_bwdtrans_quad_f64(8, 8, 0, None, None, None, None, None) == 0
    assert lib.sf_bwdtrans_hex_f64_variant(99, 8, 8, 8, 4, None, None, None, None, None, None,
                                           None) == -1
    # misaligned (odd) addresses are rejected before launch
    assert lib.sf_bwdtrans_hex_f64(8, 8, 8, 4, 0x1000, 0x1000, 0x1000, 0x1001, 0x2000, None) == -2


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg_dir = os.path.join(ROOT, "gpu-benchmarking_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        if any(part in dirpath for part in ("_build", "__pycache__")):
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cc", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "liboracle" not in text and "oracle/" not in text.replace(
                    "oracle_fill_random", ""), f


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg.capi, "_lib", None)
    monkeypatch.setattr(pkg.capi, "LIB_PATH", "/nonexistent/libsumfact.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.capi.lib()

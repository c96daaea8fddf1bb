"""The C-ABI library loads and exports every symbol include/cvhip.h declares (no GPU needed)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "cvhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cvhip_[a-z0-9_]+)\s*\(", text)) - {"cvhip_progress_fn", "cvhip_matches_fn"})


def test_header_declares_symbols():
    syms = _declared_symbols()
    assert "cvhip_correlate_images" in syms and "cvhip_orb_extract" in syms and "cvhip_ransac_score" in syms
    assert len(syms) >= 20


def test_library_exports_every_declared_symbol():
    from cybervision_amd import _lib, build

    build.build()
    L = ctypes.CDLL(str(_lib.LIB_PATH))
    for s in _declared_symbols():
        assert hasattr(L, s), f"libcvhip.so does not export {s}"


def test_binding_table_matches_header():
    from cybervision_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    assert _lib.lib().cvhip_abi_version() == 2


def test_integration_guide_names_only_declared_symbols():
    """Every cvhip_* entry point INTEGRATION.md binds or mentions is declared in include/cvhip.h (a renamed or removed
    function must not survive in the guide's Rust snippets)."""
    text = (ROOT / "INTEGRATION.md").read_text()
    named = set(re.findall(r"\b(cvhip_[a-z0-9_]+)\s*\(", text)) | set(re.findall(r"fn\s+(cvhip_[a-z0-9_]+)", text))
    named |= set(re.findall(r"`(cvhip_[a-z0-9_]+)`", text))
    declared = set(_declared_symbols())
    prefixes = {n for n in named if n.endswith("_")}   # ("cvhip_rccl_*"-style family names)
    unknown = sorted(n for n in named - declared - prefixes if not any(d.startswith(n) for d in declared if n.endswith("_")))
    assert not unknown, unknown


def test_no_product_dependency_on_oracle():
    """The product must never import, include, link or load the CPU oracle."""
    pkg = ROOT / "cybervision_amd"
    pat = re.compile(r"cvref|from\s+oracle|import\s+oracle|oracle[/.]|libcvref")
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.hpp")) + list(pkg.rglob("*.cpp")):
        m = pat.search(p.read_text())
        assert m is None, f"{p} references the oracle: {m.group(0)!r}"


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from cybervision_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "libcvhip.so")
    with pytest.raises(ImportError):
        _lib.lib()

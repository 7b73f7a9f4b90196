"""CPU: the C-ABI library loads and exports every symbol include/mcd_hip.h declares; the Python mirror
keeps the reference's names, argument order and defaults; the product path refuses to run without a GPU
(no CPU fallback).  No compute call is made here."""
import inspect
import sys
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mcd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(mcd):
    L = mcd._lib.load()
    syms = header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(L, s), "libmcd_hip.so lacks %s" % s
    assert sorted(mcd._lib.SIGNATURES) == syms  # the ctypes table and the header agree
    assert L.mcd_abi_version() == 9


def test_blaslt_companion_exports_its_header(mcd):
    """include/mcd_blaslt.h <-> libmcd_blaslt.so <-> the ctypes table (no compute call)."""
    text = open(os.path.join(ROOT, "include", "mcd_blaslt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = sorted(set(re.findall(r"\b(mcd_[a-z0-9_]+)\s*\(", text)))
    L = mcd._lib.load_blaslt()
    assert L is not None, "libmcd_blaslt.so did not load"
    assert syms == sorted(mcd._lib.BLASLT_SIGNATURES)
    for s in syms:
        assert hasattr(L, s)
    assert L.mcd_linear_residual_workspace() == 32 << 20


def test_library_is_in_tree(mcd):
    assert mcd._lib.LIB_PATH.startswith(ROOT)


def test_similarity_signatures_match_reference(mcd):
    """Names, order and defaults of reference concept_vit/similarity.py:7,33,49,75,99."""
    from mammo_clip_dissect_amd.concept_vit import similarity as s
    want = {
        "soft_wpmi": [("clip_feats", None), ("target_feats", None), ("top_k", 100), ("a", 10), ("lam", 1),
                      ("device", "cuda"), ("min_prob", 1e-7), ("p_start", 0.998), ("p_end", 0.97)],
        "wpmi": [("clip_feats", None), ("target_feats", None), ("top_k", 28), ("a", 2), ("lam", 0.6),
                 ("device", "cuda"), ("min_prob", 1e-7)],
        "rank_reorder": [("clip_feats", None), ("target_feats", None), ("device", "cuda"), ("p", 3),
                         ("top_fraction", 0.05), ("scale_p", 0.5)],
        "cos_similarity": [("clip_feats", None), ("target_feats", None), ("device", "cuda")],
        "cos_similarity_cubed": [("clip_feats", None), ("target_feats", None), ("device", "cuda"),
                                 ("batch_size", 10000), ("min_norm", 1e-3)],
    }
    for name, params in want.items():
        sig = inspect.signature(getattr(s, name))
        got = [(k, (None if v.default is inspect._empty else v.default)) for k, v in sig.parameters.items()]
        assert got == params, name


def test_no_cpu_fallback(mcd):
    import torch
    from mammo_clip_dissect_amd import core
    from mammo_clip_dissect_amd.concept_vit import similarity as s
    with pytest.raises(RuntimeError, match="GPU only"):
        s.soft_wpmi(torch.randn(128, 5), torch.randn(128, 7), device="cpu")
    with pytest.raises(TypeError, match="GPU only"):
        core.row_softmax(torch.randn(4, 5), 10.0)
    with pytest.raises(TypeError, match="GPU only"):
        core.col_topk(torch.randn(8, 3), 2)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "mammo-clip-dissect_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                t = open(os.path.join(dp, f)).read()
                assert "import oracle" not in t and "libmcd_oracle" not in t and "oracle/" not in t.replace(
                    "oracle/ and is test infrastructure", ""), os.path.join(dp, f)


def test_graft_entry_build():
    """__graft_entry__.build() (what the driver runs as the "does it build" check) compiles incrementally and its own
    checks -- ABI version, exported symbols -- agree with the library."""
    import importlib
    sys.path.insert(0, ROOT)
    g = importlib.import_module("__graft_entry__")
    g.build()

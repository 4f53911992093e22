"""The C-ABI library loads without a GPU and exports every symbol include/gomoku_hip.h declares;
compute entries refuse to run without a device instead of falling back to the CPU."""
import os
import re

import pytest

from gomokuai_amd import lib as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gomoku_hip.h")).read()
    return sorted(set(re.findall(r"\b(gmk_[a-z0-9_]+)\s*\(", text)))


def test_exports_match_header():
    L = G.load()
    declared = _declared()
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(G.EXPORTS) == declared


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = G.load()
    assert L.gmk_init(0) == -1                      # GMK_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.gmk_last_error()
    assert L.gmk_eval_batch(None, 4, None, None, None, None, None) == -4      # GMK_ERR_STATE
    with pytest.raises(G.GmkError):
        G.eval_batch_host([[[0] * 16] * 2])


def test_product_does_not_reference_oracle():
    """Nothing under gomokuai_amd/ may import, include or link the oracle."""
    pkg = os.path.join(ROOT, "gomokuai_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "gomoku_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f

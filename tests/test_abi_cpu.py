"""CPU-side checks of the drop-in boundary: the C-ABI library loads here (no GPU) and exports every
symbol include/mi355q.h declares; geometry helpers agree with the oracle; compute entry points fail
loudly without a device.  No compute calls are made."""
import re

import pytest

import oracle
from conftest import ROOT


@pytest.fixture(scope="module")
def G():
    import ggml_mi355 as g
    if not g.LIB_PATH.exists():
        import __graft_entry__ as ge
        ge.build()
    return g


def test_header_symbols_all_exported(G):
    hdr = (ROOT / "include" / "mi355q.h").read_text()
    declared = set(re.findall(r"\b(mi355q_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = G.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(G.ABI_SYMBOLS), declared ^ set(G.ABI_SYMBOLS)
    assert L.mi355q_api_version() == 3


def test_geometry_matches_oracle(G, orc):
    for t in G.WEIGHT_TYPES:
        assert G.lib().mi355q_type_supported(t) == 1
        assert G.act_type(t) == orc.vec_dot_type(t)
        for k in (256, 4096, 14336):
            assert G.row_size(t, k) == orc.row_size(t, k)
    for other in (0, 1, 30, 34, 35):                  # f32, f16, bf16, tq1_0, tq2_0: not on this path
        assert G.lib().mi355q_type_supported(other) == 0
    assert G.row_size(oracle.Q4_K, 100) == 0          # not a multiple of the block


def test_planar_rule(G):
    # planar iff a fast kernel exists and every row / plane is 16-byte aligned
    assert G.is_planar(oracle.Q4_K, 256) and G.is_planar(oracle.Q5_K, 256)
    assert G.is_planar(oracle.Q6_K, 2048) and not G.is_planar(oracle.Q6_K, 256)      # 210-byte blocks
    assert G.is_planar(oracle.Q8_0, 256) and not G.is_planar(oracle.Q8_0, 32)        # 34-byte blocks
    assert not G.is_planar(oracle.Q3_K, 4096)


def test_no_silent_fallback_without_gpu(G):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    with pytest.raises(G.Mi355qError):
        G.QWeight.from_host(oracle.Q4_K, np.zeros((1, 144), np.uint8), 1, 256)

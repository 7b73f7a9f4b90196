"""Randomised differential tests (GPU): the fuzzers under scripts/ as regression tests, a few hundred cases each."""
import os
import runpy
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_topk_fuzz(dev, seed):
    """K3 / K6 against a stable reference order over shapes around every size-class boundary, tie patterns, NaN / inf."""
    argv = sys.argv
    sys.argv = ["fuzz_topk.py", "250", str(seed)]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(ROOT, "scripts", "fuzz_topk.py"), run_name="__main__")
        assert e.value.code == 0
    finally:
        sys.argv = argv


def test_scoring_chain_fuzz(dev):
    """K2 -> K3 -> K4 -> K5 -> K6 against the oracle on random shapes (concept counts around ATen's 32-column groups, every K,
    ragged neuron counts, soft / hard WPMI, padded and unpadded S)."""
    argv = sys.argv
    sys.argv = ["fuzz_core.py", "150", "21"]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(ROOT, "scripts", "fuzz_core.py"), run_name="__main__")
        assert e.value.code == 0
    finally:
        sys.argv = argv


def test_front_fuzz(dev):
    """K1a, K1 (bit-exact in MKL's K-block order) on random (N, C, D) and leading dimensions; K0 on random hook outputs."""
    argv = sys.argv
    sys.argv = ["fuzz_front.py", "120", "31"]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(ROOT, "scripts", "fuzz_front.py"), run_name="__main__")
        assert e.value.code == 0
    finally:
        sys.argv = argv


def test_stress_chain_fuzz(dev):
    """K1s and K4s against float64 on random shapes (tile / slice / batch boundaries of both kernels)."""
    argv = sys.argv
    sys.argv = ["fuzz_stress.py", "150", "41"]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(ROOT, "scripts", "fuzz_stress.py"), run_name="__main__")
        assert e.value.code == 0
    finally:
        sys.argv = argv


def test_similarity_functions_fuzz(dev):
    """soft_wpmi, wpmi, cos_similarity, cos_similarity_cubed, rank_reorder (seeded) of the drop-in module against the oracle."""
    argv = sys.argv
    sys.argv = ["fuzz_sim.py", "25", "51"]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(ROOT, "scripts", "fuzz_sim.py"), run_name="__main__")
        assert e.value.code == 0
    finally:
        sys.argv = argv

"""A slice of tools/fuzz_parity.py in the suite: random corpora x partitionings x per-query configurations x variants x source
tweets, bit-exact against the oracle (the full fuzzer ran 2,239 cases clean on the round-2 kernels)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_shapes_bit_exact(pkg):
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for seed in range(700_000, 700_080):
        bad = fz.one_case(pkg, seed)
        assert bad is None, bad

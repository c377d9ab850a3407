"""-m gpu: a slice of tools/fuzz_parity.py in the suite -- random shapes of the path's kernels (lsh with up to 1200 planes and
600-d rows, slsh with 0-40 planes and rows of thousands of floats, the gathers, the fused top-k with ties, SipHash, the
mapper, K queued batches per launch, the evaluation kernels, the exchange's bucketing) against the oracle, bit for bit.  The tool runs thousands of cases; here 100 with a fixed seed."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_shapes_against_the_oracle(oracle, dev):  # (oracle: built; dev: skips without a GPU)
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    makers = [fz.case_lsh, fz.case_lsh, fz.case_slsh, fz.case_gather, fz.case_topk, fz.case_hash, fz.case_multi, fz.case_eval, fz.case_misc, fz.case_plugin]
    for c in range(100):
        rng = np.random.default_rng([1234, c])
        with torch.no_grad():
            desc, ok = makers[c % len(makers)](rng)
        assert all(ok.values()), (c, desc, [k for k, v in ok.items() if not v])

"""CPU: the oracle against the REFERENCE's own outputs on 55 off-golden shapes (tests/golden/sim_fuzz.npz, made by
`make_golden.py --fuzz` from /root/reference/concept_vit/similarity.py).

The six hand-picked golden shapes are reproduced bit for bit; elsewhere the oracle (and the kernels) differ from the reference in
the last bit of a few sums, because torch's CPU exp / log are MKL's (closed source).  This file bounds that directly -- it is
the statement README / DESIGN section 5 make about off-golden shapes -- and counts the bit-identical fraction."""
import numpy as np
import pytest

import util

CASES = list(range(len(util.fuzz_cases())))


def test_fuzz_fixture_covers_what_it_says():
    cs = util.fuzz_cases()
    assert len(cs) >= 40
    assert {(1651, 763, 45), (1107, 763, 38), (410, 255, 48)} <= {(c["N"], c["C"], c["U"]) for c in cs}   # the round-4 campaign's three
    assert {255, 763, 1000, 1030} <= {c["C"] for c in cs} and {"scaled", "unit"} == {c["kind"] for c in cs}


@pytest.mark.parametrize("i", CASES)
def test_oracle_against_the_reference_off_the_golden_shapes(oracle, i):
    r, P, A, ref = util.fuzz_case(i)
    stats = []
    util.fuzz_compare(oracle.soft_wpmi(P, A, top_k=r["K"]), ref["soft"], "soft_wpmi case %d" % i, stats)
    util.fuzz_compare(oracle.wpmi(P, A, top_k=28), ref["wpmi"], "wpmi case %d" % i, stats)
    # the stored top-10 lists are the reference's torch.topk of those outputs (data consistency of the fixture)
    k = ref["soft_ids10"].shape[1]
    assert np.array_equal(np.take_along_axis(ref["soft"], ref["soft_ids10"].astype(np.int64), axis=1), ref["soft_vals10"][:, :k])

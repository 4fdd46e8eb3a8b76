"""CPU: the position-parallel formulation the kernels implement (tests/closed_form.py) against the
sequential oracle, including the cases that must be routed to the serial kernel."""
import numpy as np
import pytest

import closed_form as CF
import seqgen

SPECS = [(80, 56, 4, 64), (48, 56, 4, 12), (24, 24, 12, 24), (17, 9, 2, 0), (128, 56, 12, 64), (80, 56, 1, 64)]


@pytest.mark.parametrize("spec", SPECS)
def test_closed_form_matches_oracle(oracle, spec):
    w, k, r, ms = spec
    rng = np.random.default_rng(1234 + w)
    n_fb = 0
    for it in range(64):
        mode = it % seqgen.N_MODES
        L = int(rng.choice([0, 1, k - 1, k, k + 1, k + w - 2, k + w - 1, k + w, k + w + 1, 2 * w, 2 * w + k, 700, 2500]))
        s = seqgen.adversarial(rng, mode, L)
        for pad in (False, True):
            ref = oracle.sequence_to_shmmrs(5, s, oracle.spec(w, k, r, ms), pad)
            got, fb = CF.sequence_to_shmmrs(5, s, w, k, r, ms, False, pad)
            if fb:
                n_fb += 1
                continue
            assert np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"]), (spec, mode, L, pad)
        ref = oracle.sequence_to_shmmrs(5, s, oracle.spec(w, k, r, ms, True))
        got, _ = CF.sequence_to_shmmrs(5, s, w, k, r, ms, True)
        assert np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"])
    assert n_fb < 2 * 64  # the closed form covers most inputs


def test_synth_generator_definition(oracle):
    """BASELINE.md section 4 generator: uniform ACGT, deterministic in (seed, contig, pos)"""
    a = oracle.synth_contig(2, 7, 100000)
    b = oracle.synth_contig(2, 7, 100000)
    c = oracle.synth_contig(2, 8, 100000)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert set(np.unique(a)) == set(b"ACGT")
    counts = np.bincount(a, minlength=256)[[65, 67, 71, 84]]
    assert (abs(counts - 25000) < 1000).all()
    # prefix property: a shorter contig is a prefix of a longer one
    assert np.array_equal(oracle.synth_contig(2, 7, 1000), a[:1000])

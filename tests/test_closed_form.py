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


def test_run_length_form_of_the_reduction_equals_the_window_form(oracle):
    """The list stage inside a wavefront (csrc/small.hip: reduce_keep, csrc/query_fused.hip: qf_reduce_keep -- the level-1 form of the
    per-query kernel, round 6) decides reduce_shmmr (shmmrutils.rs:359-415, no padding) element by element: an element survives iff at
    least r consecutive list elements including itself are >= it (the run of such neighbours to its left + the run to its right + 1).
    That is the window form of closed_form.reduce_closed_form -- "a minimum, ties included, of some full r-window" -- restated; both
    against the oracle's reduce_shmmr on lists with many ties, for every r the spec allows, lists shorter than r included."""
    import closed_form as CF
    rng = np.random.default_rng(7)

    def run_length_form(x, r):
        n = len(x)
        keep = np.zeros(n, dtype=bool)
        for k in range(n):
            run, left, right = 1, True, True
            for d in range(1, r):
                left = left and k - d >= 0 and x[k - d] >= x[k]
                right = right and k + d < n and x[k + d] >= x[k]
                run += int(left) + int(right)
            keep[k] = run >= r
        return keep

    for r in range(2, 13):
        for n in (0, 1, r - 1, r, r + 1, 40, 333):
            for alphabet in (3, 50, 1 << 40):
                a = np.zeros(n, dtype=CF.MM128)
                a["x"] = (rng.integers(0, alphabet, n).astype(np.uint64) << np.uint64(8)) | np.uint64(56)
                a["y"] = (np.arange(n, dtype=np.uint64) * np.uint64(37)) << np.uint64(1)
                want = CF.reduce_closed_form(a, r, False)
                got = a[run_length_form(a["x"], r)]
                assert np.array_equal(want["x"], got["x"]) and np.array_equal(want["y"], got["y"]), (r, n, alphabet)
                ref = oracle.reduce_shmmr(a, r, False)
                assert np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"]), (r, n, alphabet)

"""seeded input generators shared by the CPU and GPU tests"""
import numpy as np


def rnd(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alpha, dtype=np.uint8), int(n)))


def adversarial(rng, mode, L):
    """the edge cases the reference's semantics care about (SURVEY.md section 7 'hard parts')"""
    if mode == 0:
        return rnd(rng, L)
    if mode == 1:
        return rnd(rng, L, b"AC")  # low complexity: ties
    if mode == 2:
        return rnd(rng, L, b"ACGTacgtN")  # lower case + scattered N
    if mode == 3:
        return b"N" * int(rng.integers(0, 200)) + rnd(rng, L)  # leading N run
    if mode == 4:
        return rnd(rng, L // 2) + b"N" * int(rng.integers(1, 300)) + rnd(rng, L // 2)  # internal N run
    if mode == 5:
        return rnd(rng, L // 3) + b"AT" * int(rng.integers(1, 80)) + rnd(rng, L // 3)  # palindromic k-mers
    if mode == 6:
        unit = rnd(rng, int(rng.integers(1, 9)))
        return (unit * (L // len(unit) + 1))[:L]  # tandem repeat
    if mode == 7:
        return rnd(rng, L, b"ACGT\x00\x01\x02\x03")  # bytes 0..3 are bases too (shmmrutils.rs:427)
    raise ValueError(mode)


N_MODES = 8

"""seeded input generators shared by the CPU and GPU tests"""
import numpy as np


def rnd(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alpha, dtype=np.uint8), int(n)))


_COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def rc(seq):
    return bytes(seq).translate(_COMP)[::-1]


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


def amy1a_like(seed=4, n_hap=96, L=200_000, unit=10_000, snp_rate=0.001):
    """BASELINE.json configs[3]: one random ancestor + per-haplotype
    0.1 % SNPs and 3..10 tandem copies of a 10 kbp unit (AMY1A-like copy-number variation)."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    anc = rng.choice(acgt, L)
    u0 = L // 2
    left, rep, right = anc[:u0], anc[u0:u0 + unit], anc[u0 + unit:]
    haps = []
    for _ in range(n_hap):
        copies = int(rng.integers(3, 11))
        h = np.concatenate([left] + [rep] * copies + [right]).copy()
        n_snp = int(len(h) * snp_rate)
        pos = rng.choice(len(h), n_snp, replace=False)
        h[pos] = acgt[(np.searchsorted(acgt, h[pos]) + rng.integers(1, 4, n_snp)) % 4]
        haps.append(h.tobytes())
    return haps

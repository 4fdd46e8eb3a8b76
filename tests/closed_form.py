"""Position-parallel ("closed form") restatement of the SHIMMER selection rules.

This is the formulation the HIP kernels implement: every rule below is
a data-parallel predicate over positions / list elements plus a short serial tail per
contig.  It is validated against the sequential oracle in tests/test_closed_form.py, so a
kernel bug and a formulation bug can be told apart without a GPU.

Not part of the product; numpy only.
"""
import numpy as np

U64MAX = np.uint64(0xFFFFFFFFFFFFFFFF)
MM128 = np.dtype([("x", "<u8"), ("y", "<u8")])

_TAB = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _TAB[ord(_c)] = _i
    _TAB[ord(_c.lower())] = _i
    _TAB[_i] = _i


def u64hash(v):
    v = v.astype(np.uint64)
    with np.errstate(over="ignore"):
        v = (~v) + (v << np.uint64(21))
        v = v ^ (v >> np.uint64(24))
        v = (v + (v << np.uint64(3))) + (v << np.uint64(8))
        v = v ^ (v >> np.uint64(14))
        v = (v + (v << np.uint64(2))) + (v << np.uint64(4))
        v = v ^ (v >> np.uint64(28))
        v = v + (v << np.uint64(31))
    return v


def kmer_stream(seq, k, rid=0):
    """per-position k-mer state -> (pushed mask, x, y).  Vectorised over positions; valid for
    any input (N handled through the compacted valid-base stream)."""
    seq = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
    L = len(seq)
    c = _TAB[seq]
    valid = c < 4
    vb = c[valid].astype(np.uint64)  # compacted valid bases
    V = len(vb)
    # k-mer planes on the valid stream: F0[v] = sum_{m<k} b0[v-m] << m  (zeros before the start)
    F0 = np.zeros(V, dtype=np.uint64)
    F1 = np.zeros(V, dtype=np.uint64)
    R0 = np.zeros(V, dtype=np.uint64)
    R1 = np.zeros(V, dtype=np.uint64)
    b0 = vb & np.uint64(1)
    b1 = vb >> np.uint64(1)
    c0 = np.uint64(1) - b0  # complement planes (rc = 3 ^ c)
    c1 = np.uint64(1) - b1
    for m in range(k):
        if m >= V:
            break
        sl = slice(m, V)
        F0[sl] |= b0[: V - m] << np.uint64(m)
        F1[sl] |= b1[: V - m] << np.uint64(m)
        R0[sl] |= c0[: V - m] << np.uint64(k - 1 - m)
        R1[sl] |= c1[: V - m] << np.uint64(k - 1 - m)
    vidx = np.cumsum(valid) - 1  # index into the valid stream for every position (-1 = none yet)
    has = vidx >= 0
    f0 = np.zeros(L, dtype=np.uint64)
    f1 = np.zeros(L, dtype=np.uint64)
    r0 = np.zeros(L, dtype=np.uint64)
    r1 = np.zeros(L, dtype=np.uint64)
    f0[has] = F0[vidx[has]]
    f1[has] = F1[vidx[has]]
    r0[has] = R0[vidx[has]]
    r1[has] = R1[vidx[has]]
    pos = np.arange(L, dtype=np.uint64)
    skip = (f0 == r0) & (f1 == r1)
    pushed = (~skip) & (pos >= np.uint64(k))
    fwd = ~(r0 < f0)
    m0 = np.where(fwd, f0, r0)
    m1 = np.where(fwd, f1, r1)
    h = u64hash(m0) ^ u64hash(m1 ^ np.uint64(0xAD12CF59))
    x = (h << np.uint64(8)) | np.uint64(k)
    y = (np.uint64(rid) << np.uint64(32)) | (pos << np.uint64(1)) | (~fwd).astype(np.uint64)
    return pushed, x, y, h


def sliding_min(x, w):
    """M[j] = min(x[j-w+1..j]) with +inf (U64MAX) to the left of index 0"""
    n = len(x)
    pad = np.concatenate([np.full(w - 1, U64MAX, dtype=np.uint64), x])
    win = np.lib.stride_tricks.sliding_window_view(pad, w)
    return win.min(axis=1)[:n]


def level1_closed_form(pushed, x, y, w, k):
    """returns (list of emitted (x,y) in order, needs_fallback)."""
    L = len(x)
    idx = np.flatnonzero(pushed)
    if len(idx) == 0:
        return np.zeros(0, dtype=MM128), False
    p0 = int(idx[0])
    N = L - p0
    if len(idx) != N:  # an internal skip (palindromic k-mer): not position-parallel
        return None, True
    xs = x[p0:].copy()
    ys = y[p0:]
    Lb = (L - w + k) % (1 << 64)  # B enabled for w+k <= p < Lb  (shmmrutils.rs:516-519)
    b_lo = max(w + k, p0)  # first position where B can fire
    b_enabled_ever = b_lo < min(Lb, L)
    if p0 >= k + 2 and b_enabled_ever and (b_lo - p0) < w - 1:
        nB = b_lo - p0
        jstart = nB
        jend = min(N - 1, Lb - 1 - p0)
    else:
        nB = 0
        jstart = w - 1
        if N < w:
            return np.zeros(0, dtype=MM128), False
        jend = max(w - 1, min(N - 1, Lb - 1 - p0))
    xs[:nB] = U64MAX
    M = sliding_min(xs, w)  # window mins (partial windows on the left see +inf)
    Mv = M.copy()
    Mv[:jstart] = 0
    Mv[jend + 1:] = 0
    # E[i] = max(Mv[i..i+w-1])
    padm = np.concatenate([Mv, np.zeros(w - 1, dtype=np.uint64)])
    E = np.lib.stride_tricks.sliding_window_view(padm, w).max(axis=1)[:N]
    emit = (xs == E)
    emit[:nB] = False
    out_idx = list(np.flatnonzero(emit))
    # serial tail: (R)/(I) only
    win_lo = max(nB, jend - w + 1)
    seg = xs[win_lo:jend + 1]
    mn = seg.min()
    min_idx = win_lo + int(np.flatnonzero(seg == mn)[-1])
    mdist = jend - min_idx
    for j in range(jend + 1, N):
        if mdist == w - 1:
            lo = j - w + 1
            seg = xs[lo:j + 1]
            mn = seg.min()
            hits = np.flatnonzero(seg == mn)
            out_idx.extend(int(lo + t) for t in hits)
            min_idx = lo + int(hits[-1])
            mdist = j - min_idx
        else:
            mdist += 1
    out = np.zeros(len(out_idx), dtype=MM128)
    out["x"] = xs[out_idx]
    out["y"] = ys[out_idx]
    return out, False


def reduce_closed_form(mers, r, padding):
    """emit i iff x_i is a minimum (ties included) of some full r-window of the (padded) list"""
    a = mers
    if padding:
        s = np.zeros(r - 1, dtype=MM128)
        s["x"] = U64MAX
        s["y"] = U64MAX
        a = np.concatenate([s, mers, s])
    n = len(a)
    if n < r:
        return np.zeros(0, dtype=MM128)
    x = a["x"]
    M = sliding_min(x, r)
    M[: r - 1] = 0
    padm = np.concatenate([M, np.zeros(r - 1, dtype=np.uint64)])
    E = np.lib.stride_tricks.sliding_window_view(padm, r).max(axis=1)[:n]
    return a[x == E]


def span_filter(a, min_span):
    n = len(a)
    if n <= 2:
        return a.copy()
    pos = ((a["y"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.uint32)
    x = a["x"]
    keep = np.ones(n, dtype=bool)
    d_prev = (pos[1:-1] - pos[:-2]).astype(np.uint32)  # wrapping u32, as in release Rust
    d_next = (pos[2:] - pos[1:-1]).astype(np.uint32)
    keep[1:-1] = (d_prev > min_span) & (d_next > min_span) & (x[:-2] != x[1:-1]) & (x[1:-1] != x[2:])
    return a[keep]


def sequence_to_shmmrs(rid, seq, w, k, r, min_span, sketch=False, padding=False):
    """returns (shmmrs, needs_fallback)"""
    pushed, x, y, h = kmer_stream(seq, k, rid)
    if sketch:
        sel = pushed & (h < (U64MAX >> np.uint64(4) >> np.uint64(r)))
        a = np.zeros(int(sel.sum()), dtype=MM128)
        a["x"] = x[sel]
        a["y"] = y[sel]
        return span_filter(a, min_span), False
    l1, fb = level1_closed_form(pushed, x, y, w, k)
    if fb:
        return None, True
    if r > 1:
        l1 = reduce_closed_form(reduce_closed_form(l1, r, padding), r, padding)
    return span_filter(l1, min_span), False

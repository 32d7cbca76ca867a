#!/usr/bin/env python3
"""Check the LDS image of the n = 8 matrix-core passes (wx_mfma.h: mf_idx) for bank conflicts (development tool).

Model: 64 banks x 4 B; a 64-bit access is served 32 lanes at a time (lanes 0-31, then 32-63); within such a group lanes
that touch the SAME address are one access (broadcast), different addresses on the same bank pair serialise.  Cost of
one wave-instruction = sum over its two groups of (max accesses on one bank pair); 2 = conflict-free.
Patterns of one directional pass d of mf4_dir_pass (lane l: k = l >> 4, g = (l >> 2) & 3, I = g & 1, u = 4 (g >> 1) +
(l & 3), v = wave): operand reads r0, r1, result write wo; and the point threads' plane write / read (lane -> jl = l >> 3,
il = l & 7 of plane kl)."""
import numpy as np

L = np.arange(64)
K, G, X = L >> 4, (L >> 2) & 3, L & 3
I, U = G & 1, 4 * (G >> 1) + X


def cost(addr):
    """addr: (..., 64) doubles index per lane -> cycles (2 = conflict-free)."""
    tot = 0
    for half in (slice(0, 32), slice(32, 64)):
        a = addr[..., half]
        bank = a % 32
        worst = 0
        for b in range(32):
            sel = bank == b
            # distinct addresses on this bank pair
            vals = np.where(sel, a, -1)
            s = np.sort(vals, axis=-1)
            distinct = ((s[..., 1:] != s[..., :-1]) & (s[..., 1:] >= 0)).sum(axis=-1) + (s[..., 0] >= 0)
            worst = np.maximum(worst, distinct)
        tot = tot + worst
    return tot


def patterns(idx):
    """idx(kl, jl, il) vectorised -> dict of pattern name -> (waves.., 64) address arrays (worst wave taken later)."""
    out = {}
    v = np.arange(8)[:, None]
    out["point"] = idx(v, (L >> 3)[None, :], (L & 7)[None, :])
    for d in range(3):
        def at(node):
            if d == 0:
                return idx(v, U[None, :], node[None, :])
            if d == 1:
                return idx(v, node[None, :], U[None, :])
            return idx(node[None, :], v, U[None, :])
        out[f"d{d}.r0"] = at(K)
        out[f"d{d}.r1"] = at(4 + K)
        out[f"d{d}.wo"] = at(4 * I + K)
    return out


def score(idx, verbose=False):
    tot = 0
    for name, a in patterns(idx).items():
        c = int(np.max(cost(a)))
        tot += c - 2
        if verbose:
            print(f"   {name:8s} {c} cycles")
    return tot


def report(name, idx, point_rows=None):
    pats = patterns(idx)
    if point_rows is not None:
        v = np.arange(8)[:, None]
        pats["point"] = idx(v, point_rows[(L >> 3)][None, :], (L & 7)[None, :])
    tot = 0
    print(name)
    for pname, a in pats.items():
        c = int(np.max(cost(a)))
        tot += c - 2
        print(f"   {pname:8s} {c} cycles")
    kl, jl, il = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
    assert len(np.unique(idx(kl, jl, il))) == 512
    print("   excess cycles per pass and field:", tot)
    return tot


if __name__ == "__main__":
    # round 2's image (found for the 16 x 16 x 4 lane map): every result write of the 4 x 4 x 4 passes takes two turns
    report("round 2: kl*72 + jl*8 + (il ^ jl)", lambda kl, jl, il: kl * 72 + jl * 8 + (il ^ jl))
    # round 3: planes and rows in the order P(x) = x with bits 1 and 2 swapped, the point thread t of a plane's wave owns
    # row P(t >> 3): conflict-free
    P = np.array([(x & 1) | ((x & 4) >> 1) | ((x & 2) << 1) for x in range(8)])
    report("round 3: P(kl)*72 + P(jl)*8 + (il ^ jl), point rows P(t >> 3)", lambda kl, jl, il: P[kl] * 72 + P[jl] * 8 + (il ^ jl), P)

"""Round-3 groundwork (CPU only): the tree elimination with every row kept in ITS OWN lane.

Today (csrc/rp_kernels.hpp: tree_solve) the first lane of every chain gathers its chain's rows into
registers (Ac[5][9], 90 VGPRs per lane, allocated in all 64 lanes) and eliminates them serially.
Here lane i keeps only its own row  R[i][e] = H[i][anc_e(i)]  (9 values) and the elimination of chain
position j = 4 .. 0 runs for all chains at once:

  * a lane at position j' < j of the same chain pulls the row of the lane (j - j') below it with a
    lane shift (DPP row_shr on the GPU; np.roll here) -- with the fixed 4-link trunk the register
    index of every value it needs is a compile-time constant;
  * the trunk rows receive  sum_v H[v][t] H[v][t'] / d_v  through adds into a small per-tree table
    (ds_add_f64 on the GPU).

This script checks the scheme against a dense solve on random SPD systems with the benchmark scene's
topology (2 trees x (4-link trunk + chains of 4, 4, 4, 5, 5 links), lanes in preorder) and prints the
operation counts per solve.  Nothing here is used by the product or the tests.
"""
import numpy as np

TL = 4
CHAINS = (4, 4, 4, 5, 5)
NTREE = 2
W = 64


def topology():
    parent, depth, tree, pos, chain = [], [], [], [], []
    for t in range(NTREE):
        base = len(parent)
        for k in range(TL):
            parent.append(base + k - 1 if k else -1); depth.append(k); tree.append(t); pos.append(-1); chain.append(-1)
        for c, n in enumerate(CHAINS):
            first = len(parent)
            for k in range(n):
                parent.append(first + k - 1 if k else base + TL - 1)
                depth.append(TL + k); tree.append(t); pos.append(k); chain.append(c)
    return map(np.array, (parent, depth, tree, pos, chain))


parent, depth, tree, pos, chain = topology()
NL = len(parent)
tbase = np.array([np.flatnonzero(tree == t)[0] for t in range(NTREE)])[tree]


def anc(i, e):
    """lane of the ancestor of i at depth e (e <= depth[i]): arithmetic, as in the kernel."""
    return tbase[i] + e if e < TL else i - (depth[i] - e)


def random_system(rng):
    """SPD, tree sparse: H = sum over a few random 'contacts' on single root-to-leaf paths + diag."""
    H = np.zeros((NL, NL))
    for i in range(NL):
        H[i, i] = rng.uniform(0.5, 2.0)
    for _ in range(40):
        leaf = rng.integers(NL)
        path = [anc(leaf, e) for e in range(depth[leaf] + 1)]
        j = np.zeros(NL); j[path] = rng.normal(size=len(path))
        H += rng.uniform(0.1, 5.0) * np.outer(j, j)
    return H, rng.normal(size=NL)


def lane_solve(H, b):
    """x = H^-1 b with one row per lane.  R[i, e] = H[i][anc_e(i)], diag at e = depth[i]."""
    MD = TL + max(CHAINS)
    R = np.zeros((W, MD)); rhs = np.zeros(W)
    for i in range(NL):
        for e in range(depth[i] + 1):
            R[i, e] = H[i, anc(i, e)]
        rhs[i] = b[i]
    lpos = np.full(W, -2); lpos[:NL] = pos
    ltree = np.zeros(W, int); ltree[:NL] = tree
    ldepth = np.zeros(W, int); ldepth[:NL] = depth
    dinv = np.zeros(W)
    ops = dict(shift=0, fma=0, lds_add=0, div=0)
    T = np.zeros((NTREE, TL, TL)); Tr = np.zeros((NTREE, TL))      # trunk deltas (LDS tables)
    # ---- chains: position j = 4 .. 0, all chains at once
    for j in range(max(CHAINS) - 1, -1, -1):
        src = lpos == j
        dinv = np.where(src, 1.0 / np.where(src, R[np.arange(W), np.minimum(ldepth, MD - 1)], 1.0), dinv); ops["div"] += 1
        # (a) in-chain ancestors at position j' pull the row of the lane (j - j') below
        for jp in range(j):
            k = j - jp
            dst = (lpos == jp) & np.roll(src, -k)            # my lane + k is at position j: same chain (preorder)
            Rv = np.roll(R, -k, axis=0); bv = np.roll(rhs, -k); dv = np.roll(dinv, -k)
            ops["shift"] += (TL + jp + 1) + 3                 # the columns <= mine, the multiplier column, d, rhs
            l = Rv[:, TL + jp] * dv                            # H[v][me] / d_v: a compile-time column index
            for e in range(TL + jp + 1):
                R[:, e] = np.where(dst, R[:, e] - l * Rv[:, e], R[:, e]); ops["fma"] += 1
            rhs = np.where(dst, rhs - l * bv, rhs); ops["fma"] += 1
        # (b) trunk: adds into the per-tree table
        for i in np.flatnonzero(src):
            for t in range(TL):
                l = R[i, t] * dinv[i]
                for t2 in range(t + 1):
                    T[ltree[i], t, t2] -= l * R[i, t2]
                Tr[ltree[i], t] -= l * rhs[i]
        ops["lds_add"] += TL * (TL + 1) // 2 + TL; ops["fma"] += TL * (TL + 1) // 2 + TL
    # ---- trunk rows collect, then eliminate position TL-1 .. 0 the same way (shifts along the trunk)
    for i in range(NL):
        if lpos[i] == -1:
            t = ldepth[i]
            for t2 in range(t + 1):
                R[i, t2] += T[ltree[i], t, t2]
            rhs[i] += Tr[ltree[i], t]
    for j in range(TL - 1, -1, -1):
        src = (lpos == -1) & (ldepth == j) & (np.arange(W) < NL)
        dinv = np.where(src, 1.0 / np.where(src, R[:, j], 1.0), dinv); ops["div"] += 1
        for jp in range(j):
            k = j - jp
            dst = (lpos == -1) & (ldepth == jp) & np.roll(src, -k)
            Rv = np.roll(R, -k, axis=0); bv = np.roll(rhs, -k); dv = np.roll(dinv, -k)
            ops["shift"] += jp + 1 + 3
            l = Rv[:, jp] * dv
            for e in range(jp + 1):
                R[:, e] = np.where(dst, R[:, e] - l * Rv[:, e], R[:, e]); ops["fma"] += 1
            rhs = np.where(dst, rhs - l * bv, rhs); ops["fma"] += 1
    # ---- back-substitution root -> leaves: x_v = (b_v - sum_a H[v][a] x_a) / d_v, by depth
    x = np.zeros(W)
    for d in range(TL + max(CHAINS)):
        for i in range(NL):
            if ldepth[i] == d:
                s = rhs[i]
                for e in range(d):
                    s -= R[i, e] * x[anc(i, e)]
                x[i] = s * dinv[i]
        ops["fma"] += d + 1
    return x[:NL], ops


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(200):
        H, b = random_system(rng)
        x, ops = lane_solve(H, b)
        ref = np.linalg.solve(H, b)
        worst = max(worst, np.abs(x - ref).max() / np.abs(ref).max())
    print("lanes per tree:", NL // NTREE, " max rel. error vs dense solve over 200 systems: %.2e" % worst)
    print("per solve (wave instructions, fp64): %(shift)d lane-shifted values, %(fma)d FMA-class, %(lds_add)d LDS adds, "
          "%(div)d reciprocals" % ops)

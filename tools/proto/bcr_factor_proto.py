"""Lane-level numpy prototype of the MFMA block factorisation used by k_bcr_factor (ssba_bcr.hip).

Checks the index algebra of the kernel before it goes to the GPU: the f64 16x16x4 MFMA is emulated with its
documented operand layout (A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
D[row = (lane >> 4) + 4 reg][col = lane & 15]) and the factorisation is run tile by tile exactly as the kernel does:
LDL^T in sub-steps of four pivots, the 4x4 pivot block factored "uniformly", the row panel and the trailing update
as two MFMAs per sub-step, every other tile of the block row following with the same two operands (P, Q).
Not part of the product; run by hand:  python tools/proto/bcr_factor_proto.py
"""
import numpy as np

BD = 72
lane = np.arange(64)
LG, LJ = lane >> 4, lane & 15


def mfma(a, b, c):
    """a, b: (64,) per-lane operands; c: (4, 64) accumulator registers."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[LJ, LG] = a
    B[LG, LJ] = b
    Dm = A @ B
    out = c.copy()
    for r in range(4):
        out[r] += Dm[LG + 4 * r, LJ]
    return out


def tile_load(M, tr, tc):
    """C layout: reg r of lane (g, j) = M[16 tr + 4 r + g][16 tc + j]"""
    t = np.zeros((4, 64))
    for r in range(4):
        t[r] = M[16 * tr + 4 * r + LG, 16 * tc + LJ]
    return t


def tile_store(M, tr, tc, t):
    for r in range(4):
        M[16 * tr + 4 * r + LG, 16 * tc + LJ] = t[r]


def factor_diag_substep(T, r):
    """sub-step r of a diagonal tile T (4 regs, symmetric full tile).  Returns P, Qa, rc (4,), and updates T in place:
    rows 4r..4r+3 become the unnormalised pivot rows c_g, rows below get the rank-4 update."""
    # gather the 4x4 pivot block S[a][b] = T[4r + a][4r + b]  (lane (a, 4r + b), reg r)
    S = np.zeros((4, 4))
    for a in range(4):
        for b in range(4):
            S[a, b] = T[r][a * 16 + 4 * r + b]
    # uniform LDL^T of S (unit lower Lu, pivots d), M = Lu^-1
    d = np.zeros(4); Lu = np.eye(4)
    W = S.copy()
    for p in range(4):
        d[p] = W[p, p]
        for i in range(p + 1, 4):
            Lu[i, p] = W[i, p] / d[p]
        for i in range(p + 1, 4):
            for j in range(p + 1, 4):
                W[i, j] -= Lu[i, p] * W[p, j]
    Mi = np.linalg.inv(Lu)
    rc = 1.0 / d
    # P: lane (g, i): M[i][g] for i < 4 else 0
    P = np.where(LJ < 4, Mi[np.minimum(LJ, 3), LG], 0.0)
    y = mfma(P, T[r], np.zeros((4, 64)))[0]      # rows 0..3 of the product live in reg 0: lane (g', j) = c_{g'}[j]
    T[r] = y
    # Qa: lane (g, i): -c_g[i] * rc_g for i > 4r + 3 else 0
    Qa = np.where(LJ > 4 * r + 3, -y * rc[LG], 0.0)
    if r < 3:
        Tn = mfma(Qa, y, T)
        T[:] = Tn
    return P, Qa, rc


def panel_tile(X, Ps, Qs, nsub):
    for r in range(nsub):
        y = mfma(Ps[r], X[r], np.zeros((4, 64)))[0]
        X[r] = y
        if r < 3:
            X[:] = mfma(Qs[r], y, X)


def factor_block(D, L, U, rv):
    """returns G (lower, reciprocal diagonal), YL, YU, yr like k_bcr_factor"""
    NR = 80
    # matrix [D | r | pad] (5 column tiles) and [L | U] (9 column tiles), rows padded to 80
    MD = np.zeros((NR, 80)); MD[:BD, :BD] = D; MD[:BD, BD] = rv
    MR = np.zeros((NR, 144)); MR[:BD, :BD] = L; MR[:BD, BD:] = U
    Dt = {(k, c): tile_load(MD, k, c) for k in range(5) for c in range(k, 5)}
    Rt = {(k, c): tile_load(MR, k, c) for k in range(5) for c in range(9)}
    rc_all = np.ones(NR)
    for k in range(5):
        nsub = 4 if k < 4 else 2
        Ps, Qs = [], []
        for r in range(nsub):
            P, Qa, rc = factor_diag_substep(Dt[(k, k)], r)
            Ps.append(P); Qs.append(Qa)
            rc_all[16 * k + 4 * r: 16 * k + 4 * r + 4] = rc
        # panel ops
        for c in range(k + 1, 5):
            panel_tile(Dt[(k, c)], Ps, Qs, nsub)
        for c in range(9):
            panel_tile(Rt[(k, c)], Ps, Qs, nsub)
        # A-use form of the D panel tiles: -u * rc(row)
        Apub = {}
        for c in range(k + 1, 5):
            Apub[c] = np.stack([-Dt[(k, c)][s] * rc_all[16 * k + 4 * s + LG] for s in range(4)])
        # trailing updates
        for i in range(k + 1, 5):
            for c in range(i, 5):
                for s in range(4):
                    Dt[(i, c)] = mfma(Apub[i][s], Dt[(k, c)][s], Dt[(i, c)])
            for c in range(9):
                for s in range(4):
                    Rt[(i, c)] = mfma(Apub[i][s], Rt[(k, c)][s], Rt[(i, c)])
    for (k, c), t in Dt.items():
        tile_store(MD, k, c, t)
    for (k, c), t in Rt.items():
        tile_store(MR, k, c, t)
    rs = np.sqrt(rc_all)
    Ut = MD[:BD, :BD] * rs[:BD, None]          # true upper factor rows (valid for col >= row)
    G = np.tril(Ut.T)
    np.fill_diagonal(G, rs[:BD])               # reciprocal diagonal
    yr = MD[:BD, BD] * rs[:BD]
    Y = MR[:BD] * rs[:BD, None]
    return G, Y[:, :BD], Y[:, BD:], yr


def main():
    rng = np.random.default_rng(7)
    A = rng.standard_normal((BD, 2 * BD))
    D = A @ A.T + BD * np.eye(BD)
    L = rng.standard_normal((BD, BD)); U = rng.standard_normal((BD, BD)); rv = rng.standard_normal(BD)
    G, YL, YU, yr = factor_block(D, L, U, rv)
    Gref = np.linalg.cholesky(D)
    Gt = G.copy(); np.fill_diagonal(Gt, 1.0 / np.diag(G))
    print("G   ", np.abs(Gt - Gref).max())
    print("YL  ", np.abs(YL - np.linalg.solve(Gref, L)).max())
    print("YU  ", np.abs(YU - np.linalg.solve(Gref, U)).max())
    print("yr  ", np.abs(yr - np.linalg.solve(Gref, rv)).max())


if __name__ == "__main__":
    main()

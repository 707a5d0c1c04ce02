"""Dev tool (CPU): lane-level emulation of the register-resident 64x64 Cholesky of one wavefront
(csrc/chol_kernels.hip, factor64_wave) -- the ten lower 16x16 tiles of the block live in MFMA
accumulator registers, TRANSPOSED (reg r of lane (c, q) = S[16 ti + c][16 tj + q + 4 r]); pivots are
rank-1 v_mfma_f64_16x16x4 updates of the diagonal tile, of the tiles below it (so the sub-panel needs
no triangular solve and no inverse) and of the inverse of the diagonal tile; trailing tiles are updated
by MFMAs whose operands are the finished panel tiles, straight out of their registers.  Checks the
index logic before it goes to the GPU."""
import numpy as np


def mfma(a, b, acc):
    A = a.reshape(4, 16).T          # [i][k], lane l supplies A[l & 15][l >> 4]
    B = b.reshape(4, 16)            # [k][n], lane l supplies B[l >> 4][l & 15]
    P = A @ B
    out = acc.copy()
    for l in range(64):
        c, q = l & 15, l >> 4
        for r in range(4):
            out[r, l] = acc[r, l] + P[q + 4 * r, c]
    return out


lane = np.arange(64)
c = lane & 15
q = lane >> 4


def load_T(S, ti, tj):
    """reg r of lane (c, q) = S[16 ti + c][16 tj + q + 4 r] (lower triangle of S only)"""
    acc = np.zeros((4, 64))
    for r in range(4):
        acc[r] = S[16 * ti + c, 16 * tj + q + 4 * r]
    return acc


def factor64(S):
    T = {(ti, tj): load_T(S, ti, tj) for ti in range(4) for tj in range(ti + 1)}
    L = np.zeros((64, 64))
    Linv16 = []
    for b in range(4):
        D = T[(b, b)]
        X = np.zeros((4, 64))
        for r in range(4):
            X[r] = (q + 4 * r == c).astype(float)
        U = np.zeros((4, 64))
        Xo = np.zeros((4, 64))
        P = {ti: np.zeros((4, 64)) for ti in range(b + 1, 4)}
        for j in range(16):
            kq, rj = j & 3, j >> 2
            d = D[rj, j + 16 * kq]
            rinv = 1.0 / np.sqrt(d)
            grp = (q == kq)
            lcol = D[rj] * rinv
            bD = np.where(grp, lcol, 0.0)
            D = mfma(-bD, bD, D)
            for ti in range(b + 1, 4):
                lt = T[(ti, b)][rj] * rinv
                bT = np.where(grp, lt, 0.0)
                T[(ti, b)] = mfma(-bD, bT, T[(ti, b)])
                P[ti][rj] = np.where(grp, lt, P[ti][rj])
            xs = X[rj] * rinv
            bX = np.where(grp, xs, 0.0)
            aX = np.where(grp & (c > j), -lcol, 0.0)
            X = mfma(aX, bX, X)                      # row j of X is final but unscaled: kept in Xo
            Xo[rj] = np.where(grp, xs, Xo[rj])
            U[rj] = np.where(grp, np.where(c == j, np.sqrt(d), np.where(c > j, lcol, 0.0)), U[rj])
        # outputs of round b: U[n'][c] = L[16b + c][16b + n'],  P[ti][n'][c] = L[16ti + c][16b + n']
        for r in range(4):
            L[16 * b + c, 16 * b + q + 4 * r] = U[r]
            for ti in range(b + 1, 4):
                L[16 * ti + c, 16 * b + q + 4 * r] = P[ti][r]
        Xi = np.zeros((16, 16))
        for r in range(4):
            Xi[q + 4 * r, c] = Xo[r]
        Linv16.append(Xi)
        # trailing updates straight from the P registers
        for s in range(4):
            for ti in range(b + 1, 4):
                for tj in range(b + 1, ti + 1):
                    T[(ti, tj)] = mfma(-P[tj][s], P[ti][s], T[(ti, tj)])
    return L, Linv16


rs = np.random.RandomState(1)
A = rs.randn(64, 150)
S = A @ A.T + 0.05 * np.eye(64)
L, Linv16 = factor64(np.tril(S))
Lref = np.linalg.cholesky(S)
print("L - chol", np.abs(L - Lref).max())
for b in range(4):
    print("Linv16[%d]" % b, np.abs(Linv16[b] - np.linalg.inv(Lref[16 * b:16 * b + 16, 16 * b:16 * b + 16])).max())

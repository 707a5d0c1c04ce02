"""Dev tool (CPU): lane-level emulation of the in-wave 16x16 Cholesky + inverse built from
v_mfma_f64_16x16x4 rank-1 updates (csrc/chol_kernels.hip, factor16_mfma) -- checks the register
layout logic before it goes to the GPU."""
import numpy as np

def mfma(a, b, acc):
    """a, b: (64,) per-lane operands; acc: (4, 64) regs.  D = A B + C with
    A[i][k] = a[i + 16 k], B[k][n] = b[n + 16 k], C reg r of lane l = [row (l>>4)+4r][col l&15]."""
    A = a.reshape(4, 16).T          # [i][k]
    B = b.reshape(4, 16)            # [k][n]
    P = A @ B                       # [i][n]
    out = acc.copy()
    for l in range(64):
        c, q = l & 15, l >> 4
        for r in range(4):
            out[r, l] = acc[r, l] + P[q + 4 * r, c]
    return out

def to_acc(M):
    acc = np.zeros((4, 64))
    for l in range(64):
        c, q = l & 15, l >> 4
        for r in range(4):
            acc[r, l] = M[q + 4 * r, c]
    return acc

def from_acc(acc):
    M = np.zeros((16, 16))
    for l in range(64):
        c, q = l & 15, l >> 4
        for r in range(4):
            M[q + 4 * r, c] = acc[r, l]
    return M

def factor16(S):
    lane = np.arange(64); c = lane & 15; q = lane >> 4
    accC = to_acc(S); accX = to_acc(np.eye(16)); accU = np.zeros((4, 64))
    for j in range(16):
        kq, rj = j & 3, j >> 2
        d = accC[rj, j + 16 * kq]                 # readlane
        rinv = 1.0 / np.sqrt(d)
        grp = (q == kq)
        lcol = accC[rj] * rinv                    # lanes of group kq: l[c][j]
        b = np.where(grp, lcol, 0.0)
        a = -b
        xs = accX[rj] * rinv
        bX = np.where(grp, xs, 0.0)
        aX = np.where(grp & (c > j), -lcol, 0.0)
        accC = mfma(a, b, accC)
        accX = mfma(aX, bX, accX)
        accX[rj] = np.where(grp, xs, accX[rj])
        accU[rj] = np.where(grp, np.where(c >= j, lcol, 0.0), accU[rj])
    return from_acc(accU), from_acc(accX)         # U = L^T (upper), X = L^-1 (lower)

rs = np.random.RandomState(0)
A = rs.randn(16, 40); S = A @ A.T + 0.1 * np.eye(16)
U, X = factor16(S)
L = np.linalg.cholesky(S)
print("U - L^T", np.abs(U - L.T).max(), " X - L^-1", np.abs(X - np.linalg.inv(L)).max())
# transposes through the identity: Y = X^T = sum_s mfma(X.reg[s], I.reg[s])
accX = to_acc(X); accI = to_acc(np.eye(16)); Y = np.zeros((4, 64))
for s in range(4):
    Y = mfma(accX[s], accI[s], Y)
print("transpose", np.abs(from_acc(Y) - X.T).max())
# T1^T T2 products straight from accumulator registers
T1 = rs.randn(16, 16); T2 = rs.randn(16, 16); a1 = to_acc(T1); a2 = to_acc(T2); D = np.zeros((4, 64))
for s in range(4):
    D = mfma(a1[s], a2[s], D)
print("T1^T T2", np.abs(from_acc(D) - T1.T @ T2).max())

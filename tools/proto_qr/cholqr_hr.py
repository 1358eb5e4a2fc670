"""CPU prototype of the panel factorisation the blocked QR uses on the GPU: Cholesky-QR twice, then the Householder
representation of the orthonormal factor by the sign-shifted LU (Ballard, Demmel, Grigori, Jacquelin, Nguyen, Solomonik:
"Reconstructing Householder vectors from tall-skinny QR", 2014). Checked against LAPACK's column-by-column reflectors."""
import numpy as np, scipy.linalg as sl

def panel(P):
    b = P.shape[1]
    R1 = np.linalg.cholesky(P.T @ P).T
    Q1 = P @ np.linalg.inv(R1)
    R2 = np.linalg.cholesky(Q1.T @ Q1).T
    A = Q1 @ np.linalg.inv(R2)
    R = R2 @ R1
    D = np.zeros(b)
    for i in range(b):
        D[i] = -1.0 if A[i, i] >= 0 else 1.0
        A[i, i] -= D[i]
        A[i + 1:, i] /= A[i, i]
        A[i + 1:, i + 1:] -= np.outer(A[i + 1:, i], A[i, i + 1:])
    U = np.triu(A[:b])
    V = np.tril(A, -1); V[np.arange(b), np.arange(b)] = 1.0
    T = -U @ np.diag(D) @ np.linalg.inv(V[:b].T)
    return V, T, D[:, None] * R

rng = np.random.default_rng(1)
for m, b, cond in ((10000, 64, 1.0), (2330, 64, 1e3), (300, 64, 1e5), (64, 64, 10.0), (100, 37, 1.0)):
    P = rng.normal(size=(m, b)) @ (np.eye(b) + (cond - 1) * np.outer(rng.normal(size=b), rng.normal(size=b)) / b)
    V, T, R = panel(P)
    (qr_raw, tau), _ = sl.qr(P, mode="raw")
    Vl = np.tril(qr_raw, -1); Vl[np.arange(b), np.arange(b)] = 1.0
    Rl = np.triu(qr_raw[:b])
    H = np.eye(m) - V @ T @ V.T
    print("m %5d b %2d cond %.0e: |V - V_lapack| %.1e  |tau - tau_lapack| %.1e  |R - R_lapack|/|R| %.1e  |H'H - I| %.1e  |H'P - [R;0]|/|P| %.1e" % (
        m, b, np.linalg.cond(P), np.abs(V - Vl).max(), np.abs(np.diag(T) - tau).max(), np.abs(R - Rl).max() / np.abs(Rl).max(),
        np.abs(H.T @ H - np.eye(m)).max() if m <= 2500 else -1, (np.abs((P - V @ (T.T @ (V.T @ P)))[b:]).max() if m > b else 0.0) / np.abs(P).max()))

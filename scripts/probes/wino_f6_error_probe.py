"""Numerical probe (CPU, numpy; r06, review item 3b): what would F(6x6,3x3) cost the fp32 engine in accuracy?

Winograd / Cook-Toom F(m x m, 3 x 3) with the usual interpolation points (F(2): 0, +-1, inf; F(4): 0, +-1, +-2, inf; F(6): 0, +-1, +-2,
+-1/2, inf), everything the kernel would do in fp32 emulated in fp32 -- U = G g G^T computed in float64 and rounded once (as the host
packers do), V = B^T d B in fp32, the channel sum of U .* V accumulated in fp32, Y = A^T M A in fp32 -- against the float64 direct
convolution, on the op tests' distribution (N(0,1) inputs, He-scaled weights), per conv layer; multiplies per output: 4 / 2.25 / 1.78.

    python scripts/probes/wino_f6_error_probe.py
"""
import numpy as np


def cook_toom(m, pts):
    """A^T (m x a), G (a x 3), B^T (a x a) of F(m, 3) for the finite points pts + infinity, a = m + 2."""
    a = m + 2
    assert len(pts) == a - 1
    pts = np.asarray(pts, np.float64)
    # polynomial f_i(x) = prod_{j != i} (x - p_j), N_i = f_i(p_i)
    AT = np.zeros((m, a)); G = np.zeros((a, 3)); BT = np.zeros((a, a))
    for i, p in enumerate(pts):
        others = np.delete(pts, i)
        N = np.prod(p - others)
        AT[:, i] = p ** np.arange(m)
        G[i] = p ** np.arange(3) / N
        # row i of B^T: coefficients of f_i(x) = prod_{j != i} (x - p_j) ... times the infinity column handled below
        c = np.poly(others)[::-1]                       # ascending powers, degree a - 2
        BT[i, :a - 1] = c
    AT[m - 1, a - 1] = 1.0
    G[a - 1, 2] = 1.0
    # infinity row of B^T: coefficients of prod_j (x - p_j), degree a - 1
    BT[a - 1] = np.poly(pts)[::-1]
    # the rows above miss the (x - inf) factor's bookkeeping: fix B^T so that the identity A^T [(G g) .* (B^T d)] = conv holds (solve numerically)
    return AT, G, BT


def solve_BT(m, AT, G):
    """B^T from A^T and G by the defining identity (least squares over all monomial pairs): exact in float64 up to rounding."""
    a = m + 2
    # y_k = sum_j g_j d_{k+j};  y = A^T [(G g) .* (B^T d)]  for all g, d  <=>  for each (k, j, s): sum_i AT[k,i] G[i,j] BT[i,s] = [s == k + j]
    rows, rhs = [], []
    for k in range(m):
        for j in range(3):
            for s in range(a):
                r = np.zeros((a, a))
                r[:, s] = AT[k] * G[:, j]
                rows.append(r.ravel()); rhs.append(1.0 if s == k + j else 0.0)
    sol, *_ = np.linalg.lstsq(np.asarray(rows), np.asarray(rhs), rcond=None)
    return sol.reshape(a, a)


def winograd_conv_fp32(x, w, m, pts):
    """x [H, W, Ci] (H, W multiples of m), w [3, 3, Ci, Co] -> y [H, W, Co] (SAME, zero padding), fp32 arithmetic as the kernel's."""
    AT, G, _ = cook_toom(m, pts)
    BT = solve_BT(m, AT, G)
    a = m + 2
    H, W, Ci = x.shape
    Co = w.shape[3]
    U = np.einsum("ip,pqco,jq->ijco", G, w.astype(np.float64), G).astype(np.float32)          # double, rounded once
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    xp = np.zeros((H + 2, W + 2, Ci), np.float32)
    xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, Co), np.float32)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = xp[ty:ty + a, tx:tx + a]                                                    # [a, a, Ci]
            t = np.einsum("ip,pqc->iqc", BT32, d, dtype=np.float32)
            V = np.einsum("iqc,jq->ijc", t, BT32, dtype=np.float32)
            M = np.zeros((a, a, Co), np.float32)
            for c0 in range(0, Ci, 4):                                                      # fp32 accumulation over 4-channel chunks
                M += np.einsum("ijc,ijco->ijo", V[:, :, c0:c0 + 4], U[:, :, c0:c0 + 4], dtype=np.float32)
            t2 = np.einsum("ki,ijo->kjo", AT32, M, dtype=np.float32)
            y[ty:ty + m, tx:tx + m] = np.einsum("kjo,lj->klo", t2, AT32, dtype=np.float32)
    return y


def direct64(x, w):
    H, W, Ci = x.shape
    xp = np.zeros((H + 2, W + 2, Ci)); xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, w.shape[3]))
    for a in range(3):
        for b in range(3):
            y += xp[a:a + H, b:b + W] @ w[a, b].astype(np.float64)
    return y


def main():
    rng = np.random.default_rng(6)
    print(f"{'layer':14s} {'F(2x2) max / rms':>24s} {'F(4x4) max / rms':>24s} {'F(6x6) max / rms':>24s}")
    for ci, co in ((64, 64), (128, 128), (256, 256)):
        x = rng.standard_normal((48, 48, ci)).astype(np.float32)
        w = (rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
        ref = direct64(x, w)
        cells = []
        for m, pts in ((2, [0, 1, -1]), (4, [0, 1, -1, 2, -2]), (6, [0, 1, -1, 2, -2, 0.5, -0.5])):
            e = np.abs(winograd_conv_fp32(x, w, m, pts).astype(np.float64) - ref)
            cells.append(f"{e.max():.2e} / {np.sqrt((e ** 2).mean()):.2e}")
        print(f"{ci:4d} -> {co:4d}   " + "   ".join(f"{c:>22s}" for c in cells), flush=True)
    print("multiplies per output: F(2x2) 4.00   F(4x4) 2.25   F(6x6) 1.78   (direct: 9)")


if __name__ == "__main__":
    main()

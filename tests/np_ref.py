"""NumPy/LAPACK restatement of the P4Pf template solve (reference P4pf.cpp:930-1227), used to
cross-check the C oracle with an independent linear-algebra stack.  Test infrastructure."""
import os
import re
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_template():
    txt = open(os.path.join(_ROOT, "oracle", "p4pf_template.h")).read()
    out = {}
    for name in ("ROW", "COL", "COEF"):
        m = re.search(r"P4PF_TPL_%s\[\d+\] = \{([^}]*)\}" % name, txt)
        out[name] = np.array([int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()])
    return out["ROW"], out["COL"], out["COEF"]


def roots_from_coefs(k):
    """k[42] -> accepted (f, zb, zc, zd) rows sorted by zd, via LAPACK solve + eig."""
    row, col, coef = load_template()
    M = np.zeros((78, 88))
    M[row, col] = k[coef]
    Mr = np.linalg.solve(M[:, :78], M[:, 78:])
    A = np.zeros((10, 10))
    A[0, 1] = A[1, 5] = A[2, 6] = A[3, 7] = A[4, 8] = 1
    for i in range(5, 10):
        for j in range(10):
            A[i, j] = -Mr[79 - i, 9 - j]
    w, V = np.linalg.eig(A)
    sol = V[1:5] / V[0]
    out = []
    for j in range(10):
        if abs(sol[3, j].imag) < 1e-5 and sol[3, j].real > 1e-5:
            out.append((np.sqrt(sol[3, j].real), sol[2, j].real, sol[1, j].real, sol[0, j].real))
    out.sort(key=lambda r: r[3])
    return np.array(out).reshape(-1, 4), A


# ---------------------------------------------------------------------------------------------
# The whole of P4Pf_m once more, independent of the C oracle: the 42 template coefficients are
# derived here from the geometric constraints behind the template (law of cosines on the four
# camera-frame points Xa = (a, f), Xb = zb (b, f), Xc = zc (c, f), Xd = zd (d, f), each divided
# by |Xa - Xd|^2 / glad, unknowns zb, zc, zd and w = f^2) instead of being transcribed from
# P4pf.cpp:959-1042; linear algebra is LAPACK's (solve, eig, svd).
# ---------------------------------------------------------------------------------------------
def coefs_from_geometry(m, gl):
    """m: (4, 2) normalised image points a, b, c, d; gl: (glab, glac, glad, glbc, glbd, glcd)."""
    a, b, c, d = m
    glab, glac, glad, glbc, glbd, glcd = gl
    aa, dd, cc = a @ a, d @ d, c @ c
    ab, ac, ad, bc, bd, cd = a @ b, a @ c, a @ d, b @ c, b @ d, c @ d
    k1 = 0.5 * (glab + glac - glbc) / glad
    k2 = glac / glad
    k3 = 0.5 * (glab + glad - glbd) / glad
    k4 = 0.5 * (glac + glad - glcd) / glad
    # (Xb - Xa).(Xc - Xa) - k1 |Xa - Xd|^2 in  w zb zc, w zd^2, w zb, w zc, zb zc, w zd, zd^2, w, zb, zc, zd, 1
    e1 = [1.0, -k1, -1.0, -1.0, bc, 2 * k1, -k1 * dd, 1 - k1, -ab, -ac, 2 * k1 * ad, aa * (1 - k1)]
    # |Xa - Xc|^2 - k2 |Xa - Xd|^2 in  w zc^2, w zd^2, w zc, zc^2, w zd, zd^2, w, zc, zd, 1
    e2 = [1.0, -k2, -2.0, cc, 2 * k2, -k2 * dd, 1 - k2, -2 * ac, 2 * k2 * ad, aa * (1 - k2)]
    # (Xb - Xa).(Xd - Xa) - k3 |Xa - Xd|^2 in  w zb zd, w zd^2, w zb, w zd, zb zd, zd^2, w, zb, zd, 1
    e3 = [1.0, -k3, -1.0, 2 * k3 - 1, bd, -k3 * dd, 1 - k3, -ab, (2 * k3 - 1) * ad, aa * (1 - k3)]
    # (Xc - Xa).(Xd - Xa) - k4 |Xa - Xd|^2 in  w zc zd, w zd^2, w zc, w zd, zc zd, zd^2, w, zc, zd, 1
    e4 = [1.0, -k4, -1.0, 2 * k4 - 1, cd, -k4 * dd, 1 - k4, -ac, (2 * k4 - 1) * ad, aa * (1 - k4)]
    return np.array(e1 + e2 + e3 + e4)


def rigid_transform(p1, p2):
    """P4pf.cpp:101-235 (Arun): R, t with p2 ~ R p1 + t from the unit-normalised centred points."""
    m1, m2 = p1.mean(axis=1), p2.mean(axis=1)
    q1, q2 = p1 - m1[:, None], p2 - m2[:, None]
    u1 = q1 / np.linalg.norm(q1, axis=0)
    u2 = q2 / np.linalg.norm(q2, axis=0)
    U, s, Vt = np.linalg.svd(u2 @ u1.T)
    S = np.diag([np.sign(s[0]), np.sign(s[1]), np.sign(np.linalg.det(U @ Vt))])
    R = U @ S @ Vt
    return R, -R @ m1 + m2


def p4pf_numpy(m2d, M3D):
    """P4pf.cpp:286-507 end to end.  m2d (4, 2), M3D (4, 3) -> (n, f[n], R[n,3,3], t[n,3]) sorted by zd,
    or (-1, ...) for the degenerate case of :343-346."""
    m = np.asarray(m2d, dtype=np.float64).T                     # 2 x 4, as the reference holds it
    M = np.asarray(M3D, dtype=np.float64).T                     # 3 x 4
    mean3d = M.sum(axis=1) / 4
    M = M - mean3d[:, None]
    var = np.sqrt((M * M).sum(axis=0)).sum() / 4
    M = (1 / var) * M
    var2d = np.sqrt((m * m).sum(axis=0)).sum() / 4
    m = (1 / var2d) * m
    d2 = lambda i, j: ((M[:, i] - M[:, j]) ** 2).sum()   # noqa: E731
    gl = (d2(0, 1), d2(0, 2), d2(0, 3), d2(1, 2), d2(1, 3), d2(2, 3))
    if gl[3] * gl[4] * gl[5] * gl[0] * gl[1] * gl[2] < 1e-15:
        return -1, np.zeros(0), np.zeros((0, 3, 3)), np.zeros((0, 3))
    roots, _ = roots_from_coefs(coefs_from_geometry(m.T, gl))
    fs, Rs, ts = [], [], []
    for f, zb, zc, zd in roots:
        p3dc = np.stack([np.append(m[:, 0], f), zb * np.append(m[:, 1], f), zc * np.append(m[:, 2], f), zd * np.append(m[:, 3], f)], 1)
        pairs = ((0, 1, 0), (0, 2, 1), (0, 3, 2), (1, 2, 3), (1, 3, 4), (2, 3, 5))
        gta = sum(np.sqrt(gl[g] / ((p3dc[:, i] - p3dc[:, j]) ** 2).sum()) for i, j, g in pairs) / 6
        R, tt = rigid_transform(M, gta * p3dc)
        fs.append(var2d * f); Rs.append(R); ts.append(var * tt - R @ mean3d)
    n = len(fs)
    return n, np.array(fs), np.array(Rs).reshape(n, 3, 3), np.array(ts).reshape(n, 3)

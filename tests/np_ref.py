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

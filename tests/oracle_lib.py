"""ctypes access to the CPU oracle (oracle/build/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "build", "liboracle.so")

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint32)
_bp = C.POINTER(C.c_uint8)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


class OrcQuery(C.Structure):
    _fields_ = [("n", C.c_int), ("kx", _fp), ("ky", _fp), ("X", _fp), ("Y", _fp), ("Z", _fp),
                ("width", C.c_int), ("height", C.c_int), ("focal", C.c_double)]


class OrcParams(C.Structure):
    _fields_ = [("select_coverage", C.c_int), ("eps", C.c_double), ("max_steps", C.c_uint),
                ("inlier_divisor", C.c_uint), ("inlier_absolute", C.c_uint), ("early_exit", C.c_int)]


class OrcResult(C.Structure):
    _fields_ = [("found", C.c_int), ("focal", C.c_double), ("C", C.c_double * 3), ("Rc", C.c_double * 9),
                ("hyposet", C.c_uint), ("area_fit", C.c_int64), ("area_all", C.c_int64), ("steps", C.c_uint),
                ("best_step", C.c_uint), ("best_sol", C.c_uint), ("poses_scored", C.c_uint64)]


def load():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
    lib = C.CDLL(_SO)
    lib.orc_p3p_solve_quartic.argtypes = [_dp, _dp]
    lib.orc_p3p_solve_quartic_ld.argtypes = [_dp, _dp]
    lib.orc_p3p_compute_poses.argtypes = [_dp, _dp, _dp]
    lib.orc_p3p_compute_poses_ld.argtypes = [_dp, _dp, _dp]
    lib.orc_project.argtypes = [_dp, C.c_int, _fp, _fp, _fp, _dp, _dp]
    lib.orc_project.restype = None
    lib.orc_ransac_backmatch.argtypes = [C.POINTER(OrcQuery), C.POINTER(OrcQuery), C.POINTER(OrcParams), _up, C.c_uint, _up, C.c_uint,
                                         C.POINTER(OrcResult), _bp, C.POINTER(C.c_int)]
    lib.orc_p4pf.argtypes = [_dp] * 5
    lib.orc_p4pf_ld.argtypes = [_dp] * 5
    lib.orc_p4pf_stages.argtypes = [_dp] * 5
    lib.orc_compute_matrices.argtypes = [C.c_double, _dp, _dp, _dp]
    lib.orc_compute_matrices.restype = None
    lib.orc_score.argtypes = [_dp, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_double, _bp]
    lib.orc_coverage_area.argtypes = [C.c_int, _fp, _fp, _bp, C.c_int, C.c_int]
    lib.orc_coverage_area.restype = C.c_int64
    lib.orc_philox4x32.argtypes = [_up, _up, _up]
    lib.orc_philox4x32.restype = None
    lib.orc_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.orc_draw.restype = C.c_uint32
    lib.orc_sample_basic.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, _up, _up]
    lib.orc_sample_advanced.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, _up, _up, _up, _up,
                                        C.POINTER(C.c_int)]
    lib.orc_ransac.argtypes = [C.POINTER(OrcQuery), C.POINTER(OrcParams), _up, C.c_uint, C.POINTER(OrcResult), _bp]
    lib.orc_ransac_reference_faithful.argtypes = [C.POINTER(OrcQuery), C.POINTER(OrcParams), _up, C.c_uint, C.c_int, C.POINTER(OrcResult)]
    lib.orc_ransac_fixed_budget.argtypes = [C.POINTER(OrcQuery), C.c_double, _up, C.c_uint, C.c_int, C.POINTER(C.c_uint64)]
    lib.orc_ransac_fixed_budget.restype = C.c_uint64
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def solve_quartic(factors, long_double=False):
    f = np.ascontiguousarray(factors, dtype=np.float64)
    r = np.zeros(4)
    (lib().orc_p3p_solve_quartic_ld if long_double else lib().orc_p3p_solve_quartic)(_p(f, _dp), _p(r, _dp))
    return r


def p3p(fcols, pcols, long_double=False):
    """fcols/pcols: (3,3) arrays whose ROWS are the three vectors. Returns (status, C[4,3], Rc[4,3,3])."""
    f = np.ascontiguousarray(fcols, dtype=np.float64)
    p = np.ascontiguousarray(pcols, dtype=np.float64)
    sol = np.full(48, np.nan)
    fn = lib().orc_p3p_compute_poses_ld if long_double else lib().orc_p3p_compute_poses
    st = fn(_p(f, _dp), _p(p, _dp), _p(sol, _dp))
    sol = sol.reshape(4, 12)
    return st, sol[:, :3].copy(), sol[:, 3:].reshape(4, 3, 3).copy()


def p4pf(m2d, M3D, long_double=False):
    """m2d: (4,2), M3D: (4,3). Returns (n, f[n], R[n,3,3], t[n,3])."""
    m = np.ascontiguousarray(m2d, dtype=np.float64)
    M = np.ascontiguousarray(M3D, dtype=np.float64)
    f = np.zeros(10); R = np.zeros(90); t = np.zeros(30)
    fn = lib().orc_p4pf_ld if long_double else lib().orc_p4pf
    n = fn(_p(m, _dp), _p(M, _dp), _p(f, _dp), _p(R, _dp), _p(t, _dp))
    k = max(n, 0)
    return n, f[:k].copy(), R.reshape(10, 3, 3)[:k].copy(), t.reshape(10, 3)[:k].copy()


def p4pf_stages(m2d, M3D):
    m = np.ascontiguousarray(m2d, dtype=np.float64)
    M = np.ascontiguousarray(M3D, dtype=np.float64)
    k = np.zeros(42); A = np.zeros(100); roots = np.zeros(40)
    n = lib().orc_p4pf_stages(_p(m, _dp), _p(M, _dp), _p(k, _dp), _p(A, _dp), _p(roots, _dp))
    return n, k, A.reshape(10, 10), roots.reshape(10, 4)[:max(n, 0)].copy()


def compute_matrices(focal, Cc, Rc):
    Cc = np.ascontiguousarray(Cc, dtype=np.float64); Rc = np.ascontiguousarray(Rc, dtype=np.float64)
    P = np.zeros(12)
    lib().orc_compute_matrices(float(focal), _p(Cc, _dp), _p(Rc, _dp), _p(P, _dp))
    return P.reshape(3, 4)


def score(P, kx, ky, X, Y, Z, width, height, eps=0.5):
    P = np.ascontiguousarray(P, dtype=np.float64)
    n = len(kx)
    mask = np.zeros(n, dtype=np.uint8)
    cnt = lib().orc_score(_p(P, _dp), n, _p(kx, _fp), _p(ky, _fp), _p(X, _fp), _p(Y, _fp), _p(Z, _fp), width, height,
                          eps, _p(mask, _bp))
    return cnt, mask


def project(P, X, Y, Z):
    """QueryProcessor::Project: (n,2) projected points and (n,3) world points."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    n = len(X)
    o2 = np.zeros((n, 2)); o3 = np.zeros((n, 3))
    lib().orc_project(_p(P, _dp), n, _p(X, _fp), _p(Y, _fp), _p(Z, _fp), _p(o2, _dp), _p(o3, _dp))
    return o2, o3


def coverage_area(kx, ky, mask, width, height):
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    return lib().orc_coverage_area(len(kx), _p(kx, _fp), _p(ky, _fp), _p(m, _bp), width, height)


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32); k = np.asarray(key, dtype=np.uint32); o = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32(_p(c, _up), _p(k, _up), _p(o, _up))
    return o


def sample_basic(seed, query, hyp, m, n, lindex=None):
    out = np.zeros(4, dtype=np.uint32)
    li = None if lindex is None else np.ascontiguousarray(lindex, dtype=np.uint32)
    k = lib().orc_sample_basic(seed, query, hyp, m, n, _p(li, _up), _p(out, _up))
    return out, k


def sample_advanced(seed, query, hyp, m, n, cam_off, cam_ids, lindex=None):
    out = np.zeros(4, dtype=np.uint32)
    li = None if lindex is None else np.ascontiguousarray(lindex, dtype=np.uint32)
    co = np.ascontiguousarray(cam_off, dtype=np.uint32); ci = np.ascontiguousarray(cam_ids, dtype=np.uint32)
    nprob = C.c_int(0)
    k = lib().orc_sample_advanced(seed, query, hyp, m, n, _p(li, _up), _p(co, _up), _p(ci, _up), _p(out, _up),
                                  C.byref(nprob))
    return out, k, nprob.value


def ransac(kx, ky, X, Y, Z, width, height, focal, samples, select_coverage=True, eps=0.5, max_steps=10000,
           early_exit=True, want_mask=False):
    q = OrcQuery(len(kx), _p(kx, _fp), _p(ky, _fp), _p(X, _fp), _p(Y, _fp), _p(Z, _fp), width, height, float(focal))
    p = OrcParams(int(select_coverage), eps, max_steps, 10, 12, int(early_exit))
    s = np.ascontiguousarray(samples, dtype=np.uint32)
    res = OrcResult()
    mask = np.zeros(len(kx), dtype=np.uint8) if want_mask else None
    lib().orc_ransac(C.byref(q), C.byref(p), _p(s, _up), len(s), C.byref(res), _p(mask, _bp))
    out = dict(found=res.found, focal=res.focal, C=np.array(res.C[:]), Rc=np.array(res.Rc[:]).reshape(3, 3),
               hyposet=res.hyposet, area_fit=res.area_fit, area_all=res.area_all, steps=res.steps,
               best_step=res.best_step, best_sol=res.best_sol, poses_scored=res.poses_scored)
    if want_mask:
        out["mask"] = mask
    return out


def _result_dict(res):
    return dict(found=res.found, focal=res.focal, C=np.array(res.C[:]), Rc=np.array(res.Rc[:]).reshape(3, 3),
                hyposet=res.hyposet, area_fit=res.area_fit, area_all=res.area_all, steps=res.steps,
                best_step=res.best_step, best_sol=res.best_sol, poses_scored=res.poses_scored)


def ransac_reference_faithful(kx, ky, X, Y, Z, width, height, focal, samples, eps=0.5, max_steps=10000, early_exit=True, threads=4):
    """The sequential loop with TestHypothesis executed as the reference executes it (two h x w
    float maps per pose, OpenMP inside projection and painting only): the CPU timing mode that
    mirrors the shipped reference.  Same result fields as ransac()."""
    q = OrcQuery(len(kx), _p(kx, _fp), _p(ky, _fp), _p(X, _fp), _p(Y, _fp), _p(Z, _fp), width, height, float(focal))
    p = OrcParams(1, eps, max_steps, 10, 12, int(early_exit))
    s = np.ascontiguousarray(samples, dtype=np.uint32)
    res = OrcResult()
    lib().orc_ransac_reference_faithful(C.byref(q), C.byref(p), _p(s, _up), len(s), int(threads), C.byref(res))
    return _result_dict(res)


def ransac_backmatch(q1, q2, samples1, samples2, select_coverage=True, eps=0.5, max_steps=100, early_exit=False):
    """The advanced processor's loop with its backmatching leg.  q1 / q2: dicts with kx, ky, X, Y, Z,
    width, height, focal (q2 = the matches after backmatching, or None).  Returns the result dict
    plus second_leg (did it run), best_leg (which leg kept the pose) and mask (of that leg's matches)."""
    def mk(q):
        arrs = [np.ascontiguousarray(q[k], dtype=np.float32) for k in ("kx", "ky", "X", "Y", "Z")]
        return OrcQuery(len(arrs[0]), *[_p(a, _fp) for a in arrs], q["width"], q["height"], float(q["focal"])), arrs
    o1, keep1 = mk(q1)
    o2, keep2 = mk(q2) if q2 is not None else (None, None)
    p = OrcParams(int(select_coverage), eps, max_steps, 10, 12, int(early_exit))
    s1 = np.ascontiguousarray(samples1, dtype=np.uint32)
    s2 = np.ascontiguousarray(samples2, dtype=np.uint32) if samples2 is not None else np.zeros((0, 4), np.uint32)
    res = OrcResult()
    nmax = max(o1.n, o2.n if o2 is not None else 0)
    mask = np.zeros(nmax, dtype=np.uint8)
    leg = C.c_int(0)
    second = lib().orc_ransac_backmatch(C.byref(o1), C.byref(o2) if o2 is not None else None, C.byref(p), _p(s1, _up), len(s1),
                                        _p(s2, _up), len(s2), C.byref(res), _p(mask, _bp), C.byref(leg))
    out = _result_dict(res)
    out.update(second_leg=bool(second), best_leg=leg.value, mask=mask[:(o2.n if leg.value == 1 else o1.n)])
    return out


def ransac_fixed_budget(kx, ky, X, Y, Z, width, height, focal, samples, eps=0.5, threads=1):
    """Returns (best_count, best_slot, poses_scored) of the fixed-budget inlier-count run."""
    q = OrcQuery(len(kx), _p(kx, _fp), _p(ky, _fp), _p(X, _fp), _p(Y, _fp), _p(Z, _fp), width, height, float(focal))
    s = np.ascontiguousarray(samples, dtype=np.uint32)
    scored = C.c_uint64(0)
    key = lib().orc_ransac_fixed_budget(C.byref(q), eps, _p(s, _up), len(s), int(threads), C.byref(scored))
    return int(key >> 32), int(0xFFFFFFFF - (key & 0xFFFFFFFF)), scored.value

"""Host-side mirror of the reference's interface for the RANSAC pose-hypothesis path.

`Context` is a thin numpy wrapper over the C ABI (include/pose_amd.h).  `P3p`, `P4pf`,
`Query`, `QueryProcessorBasic` and `QueryProcessorAdvanced` keep the names, argument
meaning and error behaviour of the reference classes (P3p.hpp:54-61, P4pf.hpp:28-59,
query_loader.hpp:62-91, query_processor.hpp:33-73) so that tests read like the reference's
own drivers.  All numeric work happens on the GPU through the ABI.
"""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import Params, PoseAmdError, Result

SELECT_INLIERS, SELECT_COVERAGE = 0, 1
SAMPLE_BASIC, SAMPLE_ADVANCED = 0, 1
P3P_MAXSOL, P4PF_MAXSOL = 4, 10


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


class PreparedBatch:
    """Query descriptors of a batch in the library's layout (see Context.prepare_batch)."""
    def __init__(self, nq, descs, keep):
        self.nq, self.descs, self._keep = nq, descs, keep


class ResidentBatch:
    """A query list uploaded once (pose_amd_batch_create); run() may be called any number of times.
    Close it (or let it be collected) before its context is closed."""
    def __init__(self, ctx, queries, with_camsets=False):
        prep = queries if isinstance(queries, PreparedBatch) else ctx.prepare_batch(queries)
        self._ctx, self.nq = ctx, prep.nq
        self._b = C.c_void_p()
        ctx._chk(ctx._lib.pose_amd_batch_create(ctx._h, prep.nq, prep.descs, int(bool(with_camsets)), C.byref(self._b)), "batch_create")

    def run(self, sample_mode=SAMPLE_BASIC, seed=0, budget=10000, select_mode=SELECT_COVERAGE, eps=0.5, early_exit=True, inlier_divisor=10,
            inlier_absolute=12, raw=False):
        """What Context.ransac_batch returns for the same list."""
        prm = Params(int(select_mode), float(eps), int(bool(early_exit)), int(inlier_divisor), int(inlier_absolute))
        res = (Result * max(self.nq, 1))()
        self._ctx._chk(self._ctx._lib.pose_amd_batch_run(self._b, int(sample_mode), int(seed), int(budget), C.byref(prm), res), "batch_run")
        return res if raw else [result_dict(r) for r in res[:self.nq]]

    def set_timing(self, on=True):
        self._ctx._chk(self._ctx._lib.pose_amd_batch_set_timing(self._b, int(bool(on))), "batch_set_timing")

    def last_timing(self):
        """HIP-event kernel times (ms, summed by kind) of the last run with timing on."""
        ms = (C.c_float * 12)()
        self._ctx._chk(self._ctx._lib.pose_amd_batch_last_timing(self._b, ms), "batch_last_timing")
        return dict(sample_ms=ms[0], p3p_ms=ms[1], elim_ms=ms[2], finish_ms=ms[3], score_ms=ms[4], coverage_ms=ms[5], select_ms=ms[6],
                    total_ms=ms[7], launches=int(ms[8]), p3p_steps_computed=int(ms[9]), p4pf_steps_computed=int(ms[10]), poses_painted=int(ms[11]))

    def close(self):
        if self._b:
            self._ctx._lib.pose_amd_batch_destroy(self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            if self._ctx._h:
                self.close()
        except Exception:
            pass


def result_dict(r):
    return dict(found=r.found, focal=r.focal, C=np.array(r.C[:]), Rc=np.array(r.Rc[:]).reshape(3, 3), inliers=r.inliers,
                area_fit=r.area_fit, area_all=r.area_all, steps=r.steps, best_hyp=r.best_hyp, best_sol=r.best_sol,
                poses_scored=r.poses_scored)


class Context:
    """One GPU, one stream.  Not thread-safe (one host thread per GPU)."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.pose_amd_create(int(device), C.byref(h))
        if rc != 0:
            raise PoseAmdError(f"pose_amd_create({device}) failed ({rc}): "
                               f"{self._lib.pose_amd_last_error(None).decode()}")
        self._h = h
        self.n = 0
        self.p4pf = True

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pose_amd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc != 0:
            raise PoseAmdError(f"{what} failed ({rc}): {self._lib.pose_amd_last_error(self._h).decode()}")

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, khz = C.c_int(), C.c_int()
        self._chk(self._lib.pose_amd_device_info(self._h, name, 256, C.byref(cus), C.byref(khz)), "device_info")
        return dict(name=name.value.decode(), compute_units=cus.value, clock_khz=khz.value)

    # ---- minimal solvers ---------------------------------------------------------------
    def p3p_batch(self, fvecs, wpts):
        """fvecs, wpts: (B,3,3), [b,i,:] = i-th bearing / world point.  Returns (status[B], C[B,4,3], Rc[B,4,3,3])."""
        f = _f64(fvecs).reshape(-1, 9)
        p = _f64(wpts).reshape(-1, 9)
        b = f.shape[0]
        sol = np.zeros((b, 48))
        st = np.zeros(b, dtype=np.int32)
        self._chk(self._lib.pose_amd_p3p_batch(self._h, b, _ptr(f, _lib._dp), _ptr(p, _lib._dp), _ptr(sol, _lib._dp),
                                               _ptr(st, _lib._ip)), "p3p_batch")
        sol = sol.reshape(b, 4, 12)
        return st, sol[:, :, :3].copy(), sol[:, :, 3:].reshape(b, 4, 3, 3).copy()

    def quartic_batch(self, factors):
        f = _f64(factors).reshape(-1, 5)
        r = np.zeros((f.shape[0], 4))
        self._chk(self._lib.pose_amd_quartic_batch(self._h, f.shape[0], _ptr(f, _lib._dp), _ptr(r, _lib._dp)), "quartic_batch")
        return r

    def p4pf_batch(self, m2d, M3D):
        """m2d: (B,4,2), M3D: (B,4,3).  Returns (nsol[B], f[B,10], R[B,10,3,3], t[B,10,3])."""
        m = _f64(m2d).reshape(-1, 8)
        M = _f64(M3D).reshape(-1, 12)
        b = m.shape[0]
        nsol = np.zeros(b, dtype=np.int32)
        f = np.zeros((b, 10)); R = np.zeros((b, 90)); t = np.zeros((b, 30))
        self._chk(self._lib.pose_amd_p4pf_batch(self._h, b, _ptr(m, _lib._dp), _ptr(M, _lib._dp), _ptr(nsol, _lib._ip),
                                                _ptr(f, _lib._dp), _ptr(R, _lib._dp), _ptr(t, _lib._dp)), "p4pf_batch")
        return nsol, f, R.reshape(b, 10, 3, 3), t.reshape(b, 10, 3)

    # ---- query -------------------------------------------------------------------------
    def set_query(self, kx, ky, X, Y, Z, width, height, focal=0.0, lindex=None):
        kx, ky, X, Y, Z = map(_f32, (kx, ky, X, Y, Z))
        li = None if lindex is None else _u32(lindex)
        self._chk(self._lib.pose_amd_set_query(self._h, len(kx), _ptr(kx, _lib._fp), _ptr(ky, _lib._fp), _ptr(X, _lib._fp),
                                               _ptr(Y, _lib._fp), _ptr(Z, _lib._fp), _ptr(li, _lib._up), int(width),
                                               int(height), float(focal)), "set_query")
        self.n = len(kx)
        self.p4pf = float(focal) == 0.0

    def set_camsets(self, cam_off, cam_ids):
        co, ci = _u32(cam_off), _u32(cam_ids)
        if len(ci) == 0:
            ci = np.zeros(1, dtype=np.uint32)
        self._chk(self._lib.pose_amd_set_camsets(self._h, _ptr(co, _lib._up), _ptr(ci, _lib._up)), "set_camsets")

    def compute_matrices_batch(self, focal, Cc, Rc):
        f = _f64(focal).reshape(-1); c = _f64(Cc).reshape(-1, 3); r = _f64(Rc).reshape(-1, 9)
        P = np.zeros((len(f), 12))
        self._chk(self._lib.pose_amd_compute_matrices_batch(self._h, len(f), _ptr(f, _lib._dp), _ptr(c, _lib._dp),
                                                            _ptr(r, _lib._dp), _ptr(P, _lib._dp)), "compute_matrices_batch")
        return P.reshape(-1, 3, 4)

    def score_batch(self, P, eps=0.5, want_masks=False):
        P = _f64(P).reshape(-1, 12)
        counts = np.zeros(len(P), dtype=np.int32)
        words = None
        if want_masks:
            words = np.zeros((len(P), (self.n + 63) // 64), dtype=np.uint64)
        self._chk(self._lib.pose_amd_score_batch(self._h, len(P), _ptr(P, _lib._dp), float(eps), _ptr(counts, _lib._ip),
                                                 _ptr(words, _lib._u64)), "score_batch")
        if not want_masks:
            return counts
        bits = np.unpackbits(words.view(np.uint8).reshape(len(P), -1), axis=1, bitorder="little")[:, :self.n]
        return counts, bits

    def coverage_batch(self, P, eps=0.5):
        P = _f64(P).reshape(-1, 12)
        areas = np.zeros(len(P), dtype=np.int64)
        all_ = C.c_int64(0)
        self._chk(self._lib.pose_amd_coverage_batch(self._h, len(P), _ptr(P, _lib._dp), float(eps), _ptr(areas, _lib._i64),
                                                    C.byref(all_)), "coverage_batch")
        return areas, all_.value

    def project_batch(self, P, want_3d=False):
        """QueryProcessor::Project for each projection matrix: (nposes, n, 2) image points
        [and the (n, 3) world points]."""
        P = _f64(P).reshape(-1, 12)
        p2 = np.zeros((len(P), self.n, 2))
        p3 = np.zeros((self.n, 3)) if want_3d else None
        self._chk(self._lib.pose_amd_project_batch(self._h, len(P), _ptr(P, _lib._dp), _ptr(p2, _lib._dp), _ptr(p3, _lib._dp)),
                  "project_batch")
        return (p2, p3) if want_3d else p2

    def best_fit(self):
        """Mask of the kept pose's fitted matches and their fitpoints_2d / fitpoints_3d."""
        m = np.zeros(self.n, dtype=np.uint8)
        f2 = np.zeros((self.n, 2)); f3 = np.zeros((self.n, 3))
        k = C.c_int(0)
        self._chk(self._lib.pose_amd_best_fit(self._h, _ptr(m, _lib._u8), _ptr(f2, _lib._dp), _ptr(f3, _lib._dp), C.byref(k)), "best_fit")
        return m, f2[:k.value].copy(), f3[:k.value].copy()

    # ---- sampling ----------------------------------------------------------------------
    def sample(self, mode, seed, query_id, hyp_begin, hyp_count, export=True):
        out = np.zeros((hyp_count, 4), dtype=np.uint32) if export else None
        self._chk(self._lib.pose_amd_sample(self._h, int(mode), int(seed), int(query_id), int(hyp_begin), int(hyp_count),
                                            _ptr(out, _lib._up)), "sample")
        return out

    def set_samples(self, samples):
        s = _u32(samples).reshape(-1, 4)
        self._chk(self._lib.pose_amd_set_samples(self._h, _ptr(s, _lib._up), len(s)), "set_samples")

    # ---- RANSAC ------------------------------------------------------------------------
    def ransac_run(self, select_mode=SELECT_COVERAGE, eps=0.5, early_exit=True, inlier_divisor=10, inlier_absolute=12,
                   hyp_count=None, want_per_hyp=False):
        prm = Params(int(select_mode), float(eps), int(bool(early_exit)), int(inlier_divisor), int(inlier_absolute))
        res = Result()
        nsol = counts = None
        if want_per_hyp:
            if hyp_count is None:
                raise ValueError("hyp_count is needed to size the per-hypothesis outputs")
            ms = P4PF_MAXSOL if self.p4pf else P3P_MAXSOL
            nsol = np.zeros(hyp_count, dtype=np.int32)
            counts = np.zeros((hyp_count, ms), dtype=np.int32)
        self._chk(self._lib.pose_amd_ransac_run(self._h, C.byref(prm), C.byref(res), _ptr(nsol, _lib._ip),
                                                _ptr(counts, _lib._ip)), "ransac_run")
        out = dict(found=res.found, focal=res.focal, C=np.array(res.C[:]), Rc=np.array(res.Rc[:]).reshape(3, 3),
                   inliers=res.inliers, area_fit=res.area_fit, area_all=res.area_all, steps=res.steps,
                   best_hyp=res.best_hyp, best_sol=res.best_sol, poses_scored=res.poses_scored)
        if want_per_hyp:
            out["nsol"], out["counts"] = nsol, counts
        return out

    def prepare_batch(self, queries):
        """Descriptor array for ransac_batch, built once for query lists that are processed more
        than once (it keeps the arrays it points to alive)."""
        from ._lib import QueryDesc
        nq = len(queries)
        descs = (QueryDesc * max(nq, 1))()
        keep = []
        for i, q in enumerate(queries):
            arrs = [_f32(q[k]) for k in ("kx", "ky", "X", "Y", "Z")]
            li = _u32(q["lindex"]) if q.get("lindex") is not None else None
            co = _u32(q["cam_off"]) if q.get("cam_off") is not None else None
            ci = _u32(q["cam_ids"]) if q.get("cam_ids") is not None else None
            keep.append((arrs, li, co, ci))
            d = descs[i]
            d.n = len(arrs[0])
            d.kx, d.ky, d.X, d.Y, d.Z = (_ptr(a, _lib._fp) for a in arrs)
            d.lindex, d.cam_off, d.cam_ids = _ptr(li, _lib._up), _ptr(co, _lib._up), _ptr(ci, _lib._up)
            d.width, d.height, d.focal = int(q["width"]), int(q["height"]), float(q["focal"])
            d.query_id = int(q.get("query_id", i))
        return PreparedBatch(nq, descs, keep)

    def resident_batch(self, queries, with_camsets=False):
        """Uploads a query list once; see ResidentBatch."""
        return ResidentBatch(self, queries, with_camsets)

    def ransac_batch(self, queries, sample_mode=SAMPLE_BASIC, seed=0, budget=10000, select_mode=SELECT_COVERAGE, eps=0.5,
                     early_exit=True, inlier_divisor=10, inlier_absolute=12, raw=False):
        """Process for many independent queries in one call.  `queries`: dicts with kx, ky, X, Y, Z,
        width, height, focal (0 = unknown) and optionally lindex, cam_off, cam_ids, query_id
        (default: position in the list), or what prepare_batch made of such a list.  Returns one
        result dict per query, equal to what set_query + sample(sample_mode, seed, query_id, 0,
        budget) + ransac_run give; with raw=True the array of result records as the library
        filled it (fields as in the dicts), without the per-query conversion."""
        prep = queries if isinstance(queries, PreparedBatch) else self.prepare_batch(queries)
        nq = prep.nq
        prm = Params(int(select_mode), float(eps), int(bool(early_exit)), int(inlier_divisor), int(inlier_absolute))
        res = (Result * max(nq, 1))()
        self._chk(self._lib.pose_amd_ransac_batch(self._h, nq, prep.descs, int(sample_mode), int(seed), int(budget), C.byref(prm), res),
                  "ransac_batch")
        if raw:
            return res
        return [result_dict(r) for r in res[:nq]]

    def ransac_batch_multi(self, devices, queries, sample_mode=SAMPLE_BASIC, seed=0, budget=10000, select_mode=SELECT_COVERAGE, eps=0.5,
                           early_exit=True, inlier_divisor=10, inlier_absolute=12):
        """ransac_batch spread over the listed devices inside the library (one host thread and one
        context per listed device, queries partitioned longest first).  Same results, same order."""
        prep = queries if isinstance(queries, PreparedBatch) else self.prepare_batch(queries)
        prm = Params(int(select_mode), float(eps), int(bool(early_exit)), int(inlier_divisor), int(inlier_absolute))
        res = (Result * max(prep.nq, 1))()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        rc = self._lib.pose_amd_ransac_batch_multi(_ptr(dev, _lib._ip), len(dev), prep.nq, prep.descs, int(sample_mode), int(seed), int(budget),
                                                   C.byref(prm), res)
        if rc != 0:
            raise PoseAmdError(f"ransac_batch_multi failed ({rc}): {self._lib.pose_amd_multi_last_error().decode()}")
        return [result_dict(r) for r in res[:prep.nq]]

    def ransac_range(self, sample_mode, seed, query_id, hyp_begin, hyp_count, eps=0.5):
        """One share of a hypothesis-sharded query: (result dict with global best_hyp, packed key)."""
        res = Result()
        key = C.c_uint64(0)
        self._chk(self._lib.pose_amd_ransac_range(self._h, int(sample_mode), int(seed), int(query_id), int(hyp_begin), int(hyp_count), float(eps),
                                                  C.byref(res), C.byref(key)), "ransac_range")
        return result_dict(res), key.value

    def best_mask(self):
        m = np.zeros(self.n, dtype=np.uint8)
        self._chk(self._lib.pose_amd_best_mask(self._h, _ptr(m, _lib._u8)), "best_mask")
        return m

    # ---- descriptor matching ----------------------------------------------------------------
    def knn_set_model(self, features):
        """Model descriptors as float32 [npoints, 128] (already averaged over the views)."""
        f = np.ascontiguousarray(features, dtype=np.float32).reshape(-1, 128)
        self._chk(self._lib.pose_amd_knn_set_model(self._h, len(f), _ptr(f, _lib._fp)), "knn_set_model")

    def knn_set_model_views(self, desc_off, descriptors):
        """Model descriptors from the uint8 SIFT descriptors of every point's views (CSR offsets);
        returns the float32 [npoints, 128] means as the library formed them."""
        off = _u32(desc_off)
        d = np.ascontiguousarray(descriptors, dtype=np.uint8).reshape(-1, 128)
        n = len(off) - 1
        out = np.zeros((n, 128), dtype=np.float32)
        self._chk(self._lib.pose_amd_knn_set_model_views(self._h, n, _ptr(off, _lib._up), _ptr(d, _lib._u8), _ptr(out, _lib._fp)), "knn_set_model_views")
        return out

    def knn_match(self, descriptors, ratio=0.7):
        """Two nearest model descriptors of every query descriptor (uint8 [nq, 128]) and the ratio
        test: dict(indices [nq, 2], dists [nq, 2] float32 squared L2, good [nq] bool,
        exhaustive_rows)."""
        d = np.ascontiguousarray(descriptors, dtype=np.uint8).reshape(-1, 128)
        nq = len(d)
        idx = np.zeros((nq, 2), dtype=np.int32)
        dist = np.zeros((nq, 2), dtype=np.float32)
        good = np.zeros(nq, dtype=np.uint8)
        ex = C.c_int(0)
        self._chk(self._lib.pose_amd_knn_match(self._h, nq, _ptr(d, _lib._u8), float(ratio), _ptr(idx, _lib._ip), _ptr(dist, _lib._fp), _ptr(good, _lib._u8),
                                               C.byref(ex)), "knn_match")
        return dict(indices=idx, dists=dist, good=good.astype(bool), exhaustive_rows=ex.value)

    def knn_last_timing(self):
        ms = (C.c_float * 4)()
        self._chk(self._lib.pose_amd_knn_last_timing(self._h, ms), "knn_last_timing")
        return dict(scan_ms=ms[0], finish_ms=ms[1], exhaustive_ms=ms[2], total_ms=ms[3])

    def last_timing(self):
        ms = (C.c_float * 8)()
        self._chk(self._lib.pose_amd_last_timing(self._h, ms), "last_timing")
        return dict(solve_ms=ms[0], score_ms=ms[1], select_ms=ms[2], total_ms=ms[3], elim_ms=ms[4], finish_ms=ms[5],
                    launches=int(ms[6]))

    def synchronize(self):
        self._chk(self._lib.pose_amd_synchronize(self._h), "synchronize")


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ---------------------------------------------------------------------------------------------
# Reference-shaped classes
# ---------------------------------------------------------------------------------------------
class P3p:
    """P3p.hpp:54-61.  computePoses(featureVectors 3x3, worldPoints 3x3, solutions 3x16) -> 0 / -1.

    Columns are vectors, as in the reference.  `solutions` is filled in place
    ([C | R] blocks of 4 columns); on -1 it is left untouched."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def computePoses(self, featureVectors, worldPoints, solutions):
        f = _f64(featureVectors).reshape(3, 3).T
        p = _f64(worldPoints).reshape(3, 3).T
        st, Cc, Rc = self.ctx.p3p_batch(f[None], p[None])
        if st[0] != 0:
            return -1
        for i in range(4):
            solutions[:, 4 * i] = Cc[0, i]
            solutions[:, 4 * i + 1:4 * i + 4] = Rc[0, i]
        return 0

    def solveQuartic(self, factors, realRoots):
        realRoots[:] = self.ctx.quartic_batch(_f64(factors).reshape(1, 5))[0]
        return 0


class P4pf:
    """P4pf.hpp:28-59.  P4Pf_m(m2D 2x4, M3D 3x4, focal list, rotation list, translation list) -> 0 / -1.

    Results are APPENDED to the three lists, like the reference's push_back."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def P4Pf_m(self, m2D, M3D, focalLengths, rotationMatrices, translations):
        m = _f64(m2D).reshape(2, 4).T
        M = _f64(M3D).reshape(3, 4).T
        nsol, f, R, t = self.ctx.p4pf_batch(m[None], M[None])
        if nsol[0] < 0:
            return -1
        for i in range(nsol[0]):
            focalLengths.append(float(f[0, i]))
            rotationMatrices.append(R[0, i].copy())
            translations.append(t[0, i].copy())
        return 0

    P4Pf = P4Pf_m   # the Gauss-Jordan variant of the reference solves the same system (P4pf.cpp:558-854)


class Query:
    """The pose-related part of query_loader.hpp:62-91 plus the good_matches data the
    processors build before RANSAC (keypoints, world points, camera sets)."""

    def __init__(self, kx, ky, X, Y, Z, image_width, image_height, focal_length=0.0, lindex=None, cam_off=None,
                 cam_ids=None):
        self.kx, self.ky, self.X, self.Y, self.Z = map(_f32, (kx, ky, X, Y, Z))
        self.lindex = None if lindex is None else _u32(lindex)
        self.cam_off = None if cam_off is None else _u32(cam_off)
        self.cam_ids = None if cam_ids is None else _u32(cam_ids)
        self._w, self._h = int(image_width), int(image_height)
        self._focal = float(focal_length)
        self._C = np.zeros(3)
        self._Rc = np.eye(3)
        self._P = None
        self.fitted_mask = None
        self.fitpoints_2d = None
        self.fitpoints_3d = None

    def image_width(self): return self._w
    def image_height(self): return self._h
    def focal_length(self): return self._focal
    def camera_position(self): return self._C
    def camera_rotation(self): return self._Rc
    def proj_matrix(self): return self._P
    def set_focal_length(self, f): self._focal = float(f)
    def set_camera_position(self, c): self._C = _f64(c).reshape(3)
    def set_camera_rotation(self, r): self._Rc = _f64(r).reshape(3, 3)

    def ComputeMatrices(self, ctx=None):
        ctx = ctx or default_context()
        self._P = ctx.compute_matrices_batch([self._focal], self._C, self._Rc)[0]
        return True


class QueryProcessor:
    """query_processor.hpp:33-73: Project / TestHypothesis on the current query."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def _upload(self, q):
        self.ctx.set_query(q.kx, q.ky, q.X, q.Y, q.Z, q.image_width(), q.image_height(), q.focal_length(), q.lindex)
        if q.cam_off is not None:
            self.ctx.set_camsets(q.cam_off, q.cam_ids)

    def Project(self, query, matches=None):
        """query_processor.cpp:46-86: (fitpoints_2d, fitpoints_3d) of query.proj_matrix() for the
        matches selected by the boolean mask / index list `matches` (default: all)."""
        self._upload(query)
        p2, p3 = self.ctx.project_batch(query.proj_matrix(), want_3d=True)
        if matches is None:
            return p2[0], p3
        return p2[0][matches], p3[matches]

    # BundleAdjust with local_optimisation set: lo_steps more RANSAC steps whose minimal samples are
    # drawn among the fitted matches, every pose scored against all matches
    local_optimisation = False
    lo_steps = 64

    def BundleAdjust(self, query, fitted_matches=None):
        """query_processor.cpp:36-44: an empty function that returns true in the reference (its
        refinement is a TODO there); called between ComputeMatrices and the final Project.  The
        same here unless `local_optimisation` is set: then the step the report's outlook names runs --
        RANSAC on samples of the fitted matches only, scored on all matches with the processor's
        criterion; pose, fitted_mask and fit points change only for a strictly better pose.  The
        reference has nothing to compare this with (parity unpinned)."""
        self.lo_samples, self.lo_improved = None, False
        res = getattr(self, "result", None)
        if not self.local_optimisation or fitted_matches is None or res is None or not res["found"] or self.lo_steps <= 0:
            return True
        pos = np.nonzero(np.asarray(fitted_matches, bool))[0]
        if len(pos) < 8:
            return True
        focal = getattr(self, "_solver_focal", query.focal_length())
        m = 4 if focal == 0.0 else 3
        li = None if query.lindex is None else query.lindex[pos]
        self.ctx.set_query(query.kx[pos], query.ky[pos], query.X[pos], query.Y[pos], query.Z[pos], query.image_width(), query.image_height(), focal, li)
        t = self.ctx.sample(SAMPLE_BASIC, self.seed, getattr(self, "_query_id", 0), 0x40000000, self.lo_steps)
        t = np.where(np.arange(4)[None, :] < m, pos[np.minimum(t, len(pos) - 1)], 0).astype(np.uint32)
        self.lo_samples = t
        self.ctx.set_query(query.kx, query.ky, query.X, query.Y, query.Z, query.image_width(), query.image_height(), focal, query.lindex)
        self.ctx.set_samples(t)
        r = self.ctx.ransac_run(self.select_mode, self.inlier_eps, False, self.inlier_divisor, self.inlier_absolute)
        if not r["found"]:
            return True
        if self.select_mode == SELECT_COVERAGE:
            better = np.float32(r["area_fit"]) / np.float32(r["area_all"]) > np.float32(res["area_fit"]) / np.float32(res["area_all"])
        else:
            better = r["inliers"] > res["inliers"]
        if better:
            r["steps"] = res["steps"]
            self.result, self.lo_improved = r, True
            if focal == 0.0:
                query.set_focal_length(r["focal"])
            query.set_camera_position(r["C"])
            query.set_camera_rotation(r["Rc"])
            mm, query.fitpoints_2d, query.fitpoints_3d = self.ctx.best_fit()
            query.fitted_mask = mm.astype(bool)
            query.ComputeMatrices(self.ctx)
        return True

    def TestHypothesis(self, query, match_eps=0.5):
        """Returns (inlier mask over the matches, fit_quality) for query.proj_matrix()."""
        self._upload(query)
        cnt, bits = self.ctx.score_batch(query.proj_matrix(), match_eps, want_masks=True)
        areas, area_all = self.ctx.coverage_batch(query.proj_matrix(), match_eps)
        q = np.float32(areas[0]) / np.float32(area_all) if area_all else np.float32("nan")
        return bits[0].astype(bool), float(q)


class QueryProcessorBasic(QueryProcessor):
    """query_processor_basic.cpp:57-310 from `good_matches` on: RANSAC with the parameters
    of :107-119.  The time-seeded mt19937_64 of pose_utils.cpp:58-70 is replaced by the
    counter-based generator; `seed` and `query_id` select the stream and the consumed
    tuples are kept in `self.samples`."""
    max_ransac_steps = 10000
    inlier_eps = 0.5
    inlier_divisor = 10
    inlier_absolute = 12
    early_exit = True
    sample_mode = SAMPLE_BASIC

    def __init__(self, ctx=None, seed=0, select_mode=SELECT_COVERAGE):
        super().__init__(ctx)
        self.seed = seed
        self.select_mode = select_mode
        self.samples = None
        self.steps = 0

    def Process(self, query, query_id=0):
        self._solver_focal, self._query_id = query.focal_length(), query_id
        self._upload(query)
        # all tuples of the step budget are drawn up front (they are independent of the
        # outcome of earlier steps); ransac_run consumes them in order and stops early
        self.samples = self.ctx.sample(self.sample_mode, self.seed, query_id, 0, self.max_ransac_steps)
        r = self.ctx.ransac_run(self.select_mode, self.inlier_eps, self.early_exit, self.inlier_divisor,
                                self.inlier_absolute)
        self.steps = r["steps"]
        self.result = r
        if r["found"]:
            if query.focal_length() == 0.0:
                query.set_focal_length(r["focal"])
            query.set_camera_position(r["C"])
            query.set_camera_rotation(r["Rc"])
            # the final projection of the fitted matches (basic.cpp:296-307)
            m, query.fitpoints_2d, query.fitpoints_3d = self.ctx.best_fit()
            query.fitted_mask = m.astype(bool)
        else:
            query.fitted_mask = np.zeros(query.kx.size, bool)
            query.fitpoints_2d, query.fitpoints_3d = np.zeros((0, 2)), np.zeros((0, 3))
        query.ComputeMatrices(self.ctx)
        self.BundleAdjust(query, query.fitted_mask)
        return True


class QueryProcessorAdvanced(QueryProcessorBasic):
    """query_processor_advanced.cpp:239-624 from `good_matches` on: co-occurrence sampling
    (:329-440), exactly 100 steps, no early exit (:296,327).  Backmatching (:75-237) is a
    FLANN search outside this path: `backmatch` may be a callable returning a new Query
    with the appended matches; it is invoked once under the condition of :587-606."""
    max_ransac_steps = 100
    early_exit = False
    sample_mode = SAMPLE_ADVANCED

    def __init__(self, ctx=None, seed=0, select_mode=SELECT_COVERAGE, backmatch=None):
        super().__init__(ctx, seed, select_mode)
        self.backmatch = backmatch

    def Process(self, query, query_id=0):
        self._solver_focal, self._query_id = query.focal_length(), query_id
        self._upload(query)
        s1 = self.ctx.sample(self.sample_mode, self.seed, query_id, 0, self.max_ransac_steps)
        r = self.ctx.ransac_run(self.select_mode, self.inlier_eps, False, self.inlier_divisor, self.inlier_absolute)
        best, fit, used = r, (self.ctx.best_fit() if r["found"] else None), [s1]
        # the size of good_matches before backmatching decides (advanced.cpp:587-589)
        thresh = min(query.kx.size // self.inlier_divisor, self.inlier_absolute)
        hyposet = r["inliers"] if r["found"] else 0
        self.backmatched = False
        if self.backmatch is not None and hyposet < thresh:
            q2 = self.backmatch(query)
            self.backmatched = True
            # Backmatching appends to good_matches in place (advanced.cpp:592): the caller's query
            # takes over the extended match arrays, whichever leg keeps the pose
            for k in ("kx", "ky", "X", "Y", "Z", "lindex", "cam_off", "cam_ids"):
                setattr(query, k, getattr(q2, k))
            self._upload(query)
            s2 = self.ctx.sample(self.sample_mode, self.seed, query_id, self.max_ransac_steps, self.max_ransac_steps)
            r2 = self.ctx.ransac_run(self.select_mode, self.inlier_eps, False, self.inlier_divisor, self.inlier_absolute)
            used.append(s2)
            # hypoquality is carried over while A_g changed (advanced.cpp:600-602): compare the
            # float ratios the reference compares
            better = False
            if r2["found"]:
                if self.select_mode == SELECT_COVERAGE:
                    q_old = np.float32(best["area_fit"]) / np.float32(best["area_all"]) if best["found"] else np.float32(0)
                    q_new = np.float32(r2["area_fit"]) / np.float32(r2["area_all"])
                    better = q_new > q_old
                else:
                    better = r2["inliers"] > hyposet
            if better:
                best, fit = r2, self.ctx.best_fit()
                self.best_leg = 1
            else:
                self.best_leg = 0
                if fit is not None:       # fitted_matches of the first leg, indexed in the extended list
                    m = np.zeros(query.kx.size, np.uint8)
                    m[:len(fit[0])] = fit[0]
                    fit = (m, fit[1], fit[2])
        self.samples = np.concatenate(used)
        self.steps = len(self.samples)
        self.result = best
        if best["found"]:
            if query.focal_length() == 0.0:
                query.set_focal_length(best["focal"])
            query.set_camera_position(best["C"])
            query.set_camera_rotation(best["Rc"])
            query.fitted_mask = fit[0].astype(bool)
            query.fitpoints_2d, query.fitpoints_3d = fit[1], fit[2]
        else:
            query.fitted_mask = np.zeros(query.kx.size, bool)
            query.fitpoints_2d, query.fitpoints_3d = np.zeros((0, 2)), np.zeros((0, 3))
        query.ComputeMatrices(self.ctx)
        self.BundleAdjust(query, query.fitted_mask)
        return True

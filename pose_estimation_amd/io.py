"""Data formats on either side of the RANSAC path (host only, no device work).

* `read_bundle` reads a Bundler `bundle.out` reconstruction the way
  parse_bundler::parse_data does (parse_bundler.cpp:80-171): cameras (f, k1, k2, R, t) and 3D
  points with colour and view lists (camera, key, x, y).  The per-point camera ids are what
  SiftMatch::camset holds for the advanced sampler.
* `golden_query` converts a Bundler camera into the pose convention of Query
  (LoadGoldenQuery, import_export.cpp:87-102): position = -R^T t, rotation = (diag(-1,1,-1) R)^T.
* `Benchmark` mirrors benchmark.cpp:19-69: Process every query, compare with the golden pose,
  print the summary block and write one CSV line per query
  (index, seconds, position error, rotation error (Frobenius norm), focal length error).
"""
import time

import numpy as np


def read_bundle(path):
    """-> dict(cameras=[dict(focal_length, kappa_1, kappa_2, rotation 3x3, translation 3)],
              points float32[n,3], colors uint8[n,3], views=[int array (k,4): camera, key, x, y])."""
    with open(path) as f:
        f.readline()                                  # "# Bundle file v0.3"
        tok = f.read().split()
    it = iter(tok)
    ncam, npts = int(next(it)), int(next(it))
    cams = []
    for _ in range(ncam):
        fl, k1, k2 = (float(next(it)) for _ in range(3))
        R = np.array([float(next(it)) for _ in range(9)]).reshape(3, 3)
        t = np.array([float(next(it)) for _ in range(3)])
        cams.append(dict(focal_length=fl, kappa_1=k1, kappa_2=k2, rotation=R, translation=t))
    pts = np.zeros((npts, 3), np.float32)
    col = np.zeros((npts, 3), np.uint8)
    views = []
    for i in range(npts):
        pts[i] = [np.float32(next(it)) for _ in range(3)]
        col[i] = [int(next(it)) for _ in range(3)]
        k = int(next(it))
        v = np.zeros((k, 4))
        for j in range(k):
            v[j] = [int(next(it)), int(next(it)), float(next(it)), float(next(it))]
        views.append(v)
    return dict(cameras=cams, points=pts, colors=col, views=views)


def camsets_from_views(views, point_ids):
    """CSR camera sets (cam_off, cam_ids) of the matches whose 3D points are `point_ids`
    (query_processor_advanced.cpp:272-285: the set of cameras that see the matched point)."""
    off, ids = [0], []
    for p in point_ids:
        s = sorted({int(c) for c in views[int(p)][:, 0]})
        ids.extend(s)
        off.append(len(ids))
    return np.array(off, np.uint32), np.array(ids, np.uint32)


def golden_query(camera):
    """Bundler camera -> (focal_length, camera_position, camera_rotation) as Query stores them."""
    R = np.asarray(camera["rotation"], dtype=np.float64)
    t = np.asarray(camera["translation"], dtype=np.float64)
    xzflip = np.diag([-1.0, 1.0, -1.0])
    return float(camera["focal_length"]), -R.T @ t, (xzflip @ R).T


def Benchmark(queries, goldens, processor, out_path="out/benchmark.txt", log=print):
    """queries: pose_estimation_amd.Query objects; goldens: (focal, position, rotation) per query."""
    rows = []
    with open(out_path, "w") as out:
        for i, (query, gold) in enumerate(zip(queries, goldens)):
            t0 = time.perf_counter()
            processor.Process(query, query_id=i)
            elapsed = time.perf_counter() - t0
            gf, gC, gR = gold
            position_error = float(np.linalg.norm(np.asarray(gC) - query.camera_position()))
            rotation_error = float(np.linalg.norm(np.asarray(gR) - query.camera_rotation()))
            focal_length_error = float(gf - query.focal_length())
            if log:
                log("=================================================")
                log("Index: %d" % (i + 1))
                log("Image size: %d x %d" % (query.image_width(), query.image_height()))
                log("Computation time: %g s" % elapsed)
                log("Position error: %g" % position_error)
                log("Rotation error: %g" % rotation_error)
                log("Focal length error: %g" % focal_length_error)
                log("=================================================")
            out.write("%d,%.12g,%.12g,%.12g,%.12g\n" % (i + 1, elapsed, position_error, rotation_error, focal_length_error))
            out.flush()
            rows.append((i + 1, elapsed, position_error, rotation_error, focal_length_error))
    return rows

"""Synthetic queries shaped like the reference's data (the Dubrovnik files are empty
placeholders in the reference: data/README.txt:1, data/Dubrovnik6K/README_POSE.txt:1).

All raw values are float32 like the reference types (SIFT_keypoint.hpp:38-41,
parse_bundler.hpp:54).  Keypoints are in Lowe's top-left pixel convention, so the
observation used by solvers and scoring is u = -(x - w/2), v = -(y - h/2).
"""
import numpy as np


def random_rotation(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    return q


def small_rotation(rng, angle):
    """Rotation by `angle` radians about a random axis (Rodrigues)."""
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def make_query(n, inlier_frac, seed, width=1600, height=1200, focal=1200.0, noise_px=0.3, known_focal=False,
               depth=(10.0, 200.0)):
    """One synthetic query.  Returns a dict with kx, ky, X, Y, Z (float32), width, height,
    focal (0 when unknown), and ground truth (f_true, C_true, Rc_true, inlier mask)."""
    rng = np.random.default_rng(seed)
    Rc = random_rotation(rng)                  # cam -> world
    Cc = rng.normal(size=3) * 50.0
    n_in = int(round(n * inlier_frac))
    # points in the camera frustum (camera looks down +Z, principal point 0)
    z = rng.uniform(depth[0], depth[1], n)
    u = rng.uniform(-width / 2 + 1, width / 2 - 1, n)
    v = rng.uniform(-height / 2 + 1, height / 2 - 1, n)
    Xc = np.stack([u * z / focal, v * z / focal, z], 1)
    Xw = (Xc @ Rc.T + Cc).astype(np.float32)   # X_w = Rc X_c + C
    # re-project the float32 world points so that inliers are exact up to the added noise
    Xc32 = (Xw.astype(np.float64) - Cc) @ Rc
    up = focal * Xc32[:, 0] / Xc32[:, 2]
    vp = focal * Xc32[:, 1] / Xc32[:, 2]
    is_in = np.zeros(n, dtype=bool)
    is_in[rng.permutation(n)[:n_in]] = True
    r = noise_px * np.sqrt(rng.uniform(0, 1, n))
    ang = rng.uniform(0, 2 * np.pi, n)
    uo = np.where(is_in, up + r * np.cos(ang), rng.uniform(-width / 2 + 1, width / 2 - 1, n))
    vo = np.where(is_in, vp + r * np.sin(ang), rng.uniform(-height / 2 + 1, height / 2 - 1, n))
    kx = (width / 2.0 - uo).astype(np.float32)
    ky = (height / 2.0 - vo).astype(np.float32)
    return dict(kx=kx, ky=ky, X=Xw[:, 0].copy(), Y=Xw[:, 1].copy(), Z=Xw[:, 2].copy(), width=width, height=height,
                focal=float(focal) if known_focal else 0.0, f_true=float(focal), C_true=Cc, Rc_true=Rc, inliers=is_in)


def add_camsets(q, seed, n_cams=6000, cluster=12):
    """Camera sets for the advanced sampler (report section 4.4: 2-4 cameras per point).
    Inliers share a cluster of cameras; at least 1% of the matches get >= 5 cameras."""
    rng = np.random.default_rng(seed)
    n = len(q["kx"])
    common = rng.choice(n_cams, cluster, replace=False)
    off = [0]
    ids = []
    for i in range(n):
        k = int(rng.integers(2, 5))
        if rng.uniform() < 0.03:
            k = int(rng.integers(5, 9))
        pool = common if q["inliers"][i] else np.arange(n_cams)
        k = min(k, len(pool))
        s = np.sort(rng.choice(pool, k, replace=False)).astype(np.uint32)
        ids.append(s)
        off.append(off[-1] + k)
    q["cam_off"] = np.array(off, dtype=np.uint32)
    q["cam_ids"] = np.concatenate(ids).astype(np.uint32)
    q["cam_common"], q["cam_total"] = common, n_cams
    return q


def add_camsets_vec(q, seed, n_cams=6000, cluster=12):
    """add_camsets with the same distribution drawn with array operations (a different random
    stream, so different sets): 1 ms instead of 120 ms per query of 5 000 matches, for callers that
    build hundreds of queries."""
    rng = np.random.default_rng(seed)
    n = len(q["kx"])
    common = np.sort(rng.choice(n_cams, cluster, replace=False))
    k = np.where(rng.uniform(size=n) < 0.03, rng.integers(5, 9, n), rng.integers(2, 5, n))
    ids_in = common[np.argsort(rng.random((n, cluster)), axis=1)[:, :8]]      # a random arrangement of the cluster per point
    ids_out = rng.integers(0, n_cams, (n, 8))
    while True:                                                                # 8 distinct cameras per point
        s = np.sort(ids_out, axis=1)
        dup = (s[:, 1:] == s[:, :-1]).any(1)
        if not dup.any():
            break
        ids_out[dup] = rng.integers(0, n_cams, (int(dup.sum()), 8))
    first = np.arange(8)[None, :] < k[:, None]
    ids = np.where(first, np.where(np.asarray(q["inliers"], bool)[:, None], ids_in, ids_out), n_cams + 1)
    ids.sort(axis=1)                                                           # the k kept ids ascending, the sentinel behind them
    q["cam_off"] = np.concatenate([[0], np.cumsum(k)]).astype(np.uint32)
    q["cam_ids"] = ids[first].astype(np.uint32)
    q["cam_common"], q["cam_total"] = common, n_cams
    return q


def append_matches(q, extra, inlier_frac, seed, noise_px=0.3, many_cams_frac=0.25):
    """What a backmatching pass does to good_matches (query_processor_advanced.cpp:592): a copy of
    the query with `extra` matches appended, a fraction of them consistent with the planted pose
    (camera sets included when the query has them).  The original matches keep their positions."""
    rng = np.random.default_rng(seed)
    w, h, f = q["width"], q["height"], q["f_true"]
    Rc, Cc = q["Rc_true"], q["C_true"]
    n_in = int(round(extra * inlier_frac))
    z = rng.uniform(10.0, 200.0, extra)
    u = rng.uniform(-w / 2 + 1, w / 2 - 1, extra)
    v = rng.uniform(-h / 2 + 1, h / 2 - 1, extra)
    Xw = (np.stack([u * z / f, v * z / f, z], 1) @ Rc.T + Cc).astype(np.float32)
    Xc32 = (Xw.astype(np.float64) - Cc) @ Rc
    up, vp = f * Xc32[:, 0] / Xc32[:, 2], f * Xc32[:, 1] / Xc32[:, 2]
    is_in = np.zeros(extra, dtype=bool)
    is_in[rng.permutation(extra)[:n_in]] = True
    r = noise_px * np.sqrt(rng.uniform(0, 1, extra))
    ang = rng.uniform(0, 2 * np.pi, extra)
    uo = np.where(is_in, up + r * np.cos(ang), rng.uniform(-w / 2 + 1, w / 2 - 1, extra))
    vo = np.where(is_in, vp + r * np.sin(ang), rng.uniform(-h / 2 + 1, h / 2 - 1, extra))
    out = dict(q)
    out["kx"] = np.concatenate([q["kx"], (w / 2.0 - uo).astype(np.float32)])
    out["ky"] = np.concatenate([q["ky"], (h / 2.0 - vo).astype(np.float32)])
    for k, col in (("X", 0), ("Y", 1), ("Z", 2)):
        out[k] = np.concatenate([q[k], Xw[:, col]])
    out["inliers"] = np.concatenate([q["inliers"], is_in])
    if "cam_off" in q:
        off, ids = list(q["cam_off"]), [q["cam_ids"]]
        for i in range(extra):
            pool = q["cam_common"] if is_in[i] else np.arange(q["cam_total"])
            k = int(rng.integers(2, 5))
            if rng.uniform() < many_cams_frac:         # seen by >= 5 cameras: can start a tuple (advanced.cpp:350-359)
                k = int(rng.integers(5, 9))
            k = min(k, len(pool))
            ids.append(np.sort(rng.choice(pool, k, replace=False)).astype(np.uint32))
            off.append(off[-1] + k)
        out["cam_off"] = np.array(off, dtype=np.uint32)
        out["cam_ids"] = np.concatenate(ids).astype(np.uint32)
    return out


def random_p4pf_problems(count, seed, contaminate=(0, 1, 2), noise=0.3):
    """Minimal P4Pf problems (m2d[count,4,2] centred px, M3D[count,4,3]) rounded to float32."""
    rng = np.random.default_rng(seed)
    m = np.zeros((count, 4, 2))
    M = np.zeros((count, 4, 3))
    for i in range(count):
        f = rng.uniform(600, 2000)
        Xc = np.stack([rng.uniform(-1, 1, 4) * rng.uniform(5, 50), rng.uniform(-1, 1, 4) * rng.uniform(5, 50),
                       rng.uniform(10, 200, 4)], 1)
        mi = f * Xc[:, :2] / Xc[:, 2:3] + rng.uniform(-0.3, 0.3, (4, 2)) * (noise / 0.3)
        R = random_rotation(rng)
        Wp = (Xc - rng.normal(size=3) * 30) @ R
        for _ in range(contaminate[i % len(contaminate)]):
            mi[rng.integers(4)] = rng.uniform(-800, 800, 2)
        m[i] = mi.astype(np.float32)
        M[i] = Wp.astype(np.float32)
    return m, M


def random_p3p_problems(count, seed):
    """Minimal P3P problems: unit bearings f[count,3,3] and world points P[count,3,3]."""
    rng = np.random.default_rng(seed)
    f = np.zeros((count, 3, 3))
    P = np.zeros((count, 3, 3))
    for i in range(count):
        foc = rng.uniform(600, 2000)
        Xc = np.stack([rng.uniform(-1, 1, 3) * rng.uniform(5, 50), rng.uniform(-1, 1, 3) * rng.uniform(5, 50),
                       rng.uniform(10, 200, 3)], 1)
        uv = (foc * Xc[:, :2] / Xc[:, 2:3]).astype(np.float32).astype(np.float64)
        b = np.concatenate([uv, np.full((3, 1), foc)], 1)
        f[i] = b / np.linalg.norm(b, axis=1, keepdims=True)
        R = random_rotation(rng)
        P[i] = ((Xc - rng.normal(size=3) * 30) @ R).astype(np.float32)
    return f, P

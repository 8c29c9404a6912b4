"""Mesh input helpers for the hot path's callers: Wavefront OBJ reader and a
deterministic vertex-clustering decimator.

The reference loads meshes with pyigl (`igl.readOBJ`, exp_bunny/test.py:71-77) and gets
its ~5k-face working meshes from MATLAB (`init/cnlos_bunny_threshold_64.obj`, not in
the repository).  These helpers produce the benchmark mesh from
the shipped ground-truth bunny instead.
"""
import numpy as np


def read_obj(path):
    """Return (v float32 [V,3], f int32 [F,3]) from a triangle OBJ (v / f records only)."""
    vs, fs = [], []
    with open(path, "r") as fh:
        for line in fh:
            if line.startswith("v "):
                p = line.split()
                vs.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                p = line.split()[1:]
                idx = [int(t.split("/")[0]) - 1 for t in p]
                for k in range(1, len(idx) - 1):      # fan-triangulate polygons
                    fs.append((idx[0], idx[k], idx[k + 1]))
    return (np.asarray(vs, dtype=np.float32).reshape(-1, 3),
            np.asarray(fs, dtype=np.int32).reshape(-1, 3))


def write_obj(path, v, f):
    with open(path, "w") as fh:
        for p in v:
            fh.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        for t in f:
            fh.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))


def drop_coincident_faces(v, f, prefer=(0.0, 0.0, -1.0)):
    """Faces as sets: of all faces over the same three vertices (any rotation, EITHER winding) keep one -- the one whose
    normal points most along `prefer` (default -z: towards the wall z = 0 the scenes are seen from), first occurrence
    on ties -- and drop faces of zero area (float32, the renderer's precision).  Vertex clustering flattens thin
    features (the bunny's ears) into two coincident sheets of opposite winding; which of two coincident triangles a
    ray hits is decided by the last bit of `t`, so such pairs make every visibility result a coin toss."""
    v32 = np.asarray(v, dtype=np.float32)
    f = np.asarray(f)
    if f.size == 0:
        return f
    p0, p1, p2 = (v32[f[:, k]].astype(np.float64) for k in range(3))
    n = np.cross(p1 - p0, p2 - p0)
    area = np.linalg.norm(n, axis=1)
    score = n @ np.asarray(prefer, dtype=np.float64)
    key = np.sort(f, axis=1)
    # stable lexicographic order over (vertex set, -score, original index): the first of each group wins
    order = np.lexsort((np.arange(f.shape[0]), -score, key[:, 2], key[:, 1], key[:, 0]))
    ks = key[order]
    first = np.ones(f.shape[0], dtype=bool)
    first[1:] = np.any(ks[1:] != ks[:-1], axis=1)
    keep = np.zeros(f.shape[0], dtype=bool)
    keep[order[first]] = True
    keep &= area > 0
    return f[keep]


def cluster_decimate(v, f, cell):
    """Vertex clustering on a uniform grid of edge `cell`: vertices of a cell merge into
    their mean, faces that collapse are dropped, and of faces over the same vertex set (either winding) one
    survives (drop_coincident_faces).  Winding is preserved.  Deterministic (pure numpy, stable sorts)."""
    v = np.asarray(v, dtype=np.float64)
    lo = v.min(axis=0)
    key3 = np.floor((v - lo) / cell).astype(np.int64)
    dims = key3.max(axis=0) + 1
    key = (key3[:, 0] * dims[1] + key3[:, 1]) * dims[2] + key3[:, 2]
    uniq, inv = np.unique(key, return_inverse=True)
    cnt = np.bincount(inv, minlength=uniq.size).astype(np.float64)
    nv = np.stack([np.bincount(inv, weights=v[:, c], minlength=uniq.size) / cnt for c in range(3)], axis=1)
    nf = inv[np.asarray(f, dtype=np.int64)]
    keep = (nf[:, 0] != nf[:, 1]) & (nf[:, 1] != nf[:, 2]) & (nf[:, 0] != nf[:, 2])
    nf = drop_coincident_faces(nv, nf[keep])
    used = np.unique(nf)
    remap = -np.ones(nv.shape[0], dtype=np.int64)
    remap[used] = np.arange(used.size)
    return nv[used].astype(np.float32), remap[nf].astype(np.int32)


def decimate_to(v, f, target_faces, tol=0.03, iters=40):
    """Bisect the cluster size until the face count is within `tol` of `target_faces`."""
    v = np.asarray(v)
    ext = float((v.max(axis=0) - v.min(axis=0)).max())
    lo_c, hi_c = ext / 2000.0, ext / 4.0
    best = None
    for _ in range(iters):
        c = np.sqrt(lo_c * hi_c)
        nv, nf = cluster_decimate(v, f, c)
        if best is None or abs(nf.shape[0] - target_faces) < abs(best[1].shape[0] - target_faces):
            best = (nv, nf)
        if abs(nf.shape[0] - target_faces) <= tol * target_faces:
            break
        if nf.shape[0] > target_faces:
            lo_c = c
        else:
            hi_c = c
    return best


def face_affinity(f):
    """int32 [F,3]: for face i and its edge (f[i,k], f[i,(k+1)%3]) the face on the other side, or -1
    on a boundary / non-manifold edge.  Stands in for the reference's cgal_api.face_affinity
    (cgal_api/c_cgal_api.cpp:156-171, a CGAL half-edge walk); the regulariser only sums over the
    up-to-three neighbours, so the slot order does not matter."""
    f = np.asarray(f)
    F = f.shape[0]
    a = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0).astype(np.int64)
    a.sort(axis=1)
    key = a[:, 0] * (int(f.max()) + 1 if F else 1) + a[:, 1]
    face = np.tile(np.arange(F), 3)
    slot = np.repeat(np.arange(3), F)
    order = np.argsort(key, kind="stable")
    ks, fs, ss = key[order], face[order], slot[order]
    out = -np.ones((F, 3), np.int32)
    same_next = np.zeros(len(ks), bool)
    same_next[:-1] = ks[1:] == ks[:-1]
    same_prev = np.zeros(len(ks), bool)
    same_prev[1:] = same_next[:-1]
    # manifold interior edges appear exactly twice
    first = same_next & ~same_prev
    idx = np.nonzero(first)[0]
    twice = np.ones(len(idx), bool)
    nxt2 = idx + 2
    ok = nxt2 < len(ks)
    twice[ok] = ks[nxt2[ok]] != ks[idx[ok]]
    idx = idx[twice]
    out[fs[idx], ss[idx]] = fs[idx + 1]
    out[fs[idx + 1], ss[idx + 1]] = fs[idx]
    return np.ascontiguousarray(out)


def read_transient_mat(path, fold=1):
    """Measured data of the reference's real-data experiments as the hot path consumes it.

    The shipped files (exp_mannequin/transient.mat, exp_s/transient.mat, exp_su/...) are MAT v5 with
    `transient` uint8 [L, T] + `lighting` float64 [L, 3] (mannequin) or `rect_data` [R, R, T]
    (exp_s/test.py:64-68); the scripts read them with scipy.io.loadmat and reshape ad hoc.
    Returns a dict: `transient` float64 [L, T // fold] (C order; `fold` adjacent bins summed -- BASELINE
    config 4 uses the 2048-bin mannequin data pairwise-summed to 1024), `lighting` float32 [L, 3] and
    `lighting_normal` float32 [L, 3] = (0, 0, 1) when the file has `lighting`.
    """
    import scipy.io
    m = scipy.io.loadmat(path)
    if "transient" in m:
        t = np.asarray(m["transient"], dtype=np.float64)
    elif "rect_data" in m:
        r = np.asarray(m["rect_data"], dtype=np.float64)
        t = r.reshape(-1, r.shape[-1])
    else:
        raise ValueError("%s: no 'transient' or 'rect_data' array" % path)
    if t.ndim != 2:
        raise ValueError("%s: transient must be [L, T]" % path)
    fold = int(fold)
    if fold > 1:
        if t.shape[1] % fold:
            raise ValueError("number of bins %d is not a multiple of fold=%d" % (t.shape[1], fold))
        t = t.reshape(t.shape[0], t.shape[1] // fold, fold).sum(axis=2)
    out = {"transient": np.ascontiguousarray(t)}
    if "lighting" in m:
        lighting = np.ascontiguousarray(m["lighting"], dtype=np.float32)
        if lighting.shape != (t.shape[0], 3):
            raise ValueError("%s: lighting must be [L, 3]" % path)
        out["lighting"] = lighting
        out["lighting_normal"] = np.ascontiguousarray(np.tile(np.array([0, 0, 1], np.float32), (t.shape[0], 1)))
    return out


def subdivide(v, f, times=1):
    """1 -> 4 midpoint subdivision with shared edge midpoints (keeps the surface, multiplies F by 4 per
    pass).  Used to build large synthetic meshes of a known shape for tests and side benchmarks."""
    v = np.asarray(v, np.float32)
    f = np.asarray(f, np.int64)
    for _ in range(int(times)):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        es = np.sort(e, axis=1)
        key = es[:, 0] * (v.shape[0] + 1) + es[:, 1]
        uk, inv = np.unique(key, return_inverse=True)
        a = (uk // (v.shape[0] + 1)).astype(np.int64)
        b = (uk % (v.shape[0] + 1)).astype(np.int64)
        mid = ((v[a].astype(np.float64) + v[b]) / 2).astype(np.float32)
        base = v.shape[0]
        F = f.shape[0]
        m01, m12, m20 = base + inv[:F], base + inv[F:2 * F], base + inv[2 * F:]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([m01, f[:, 1], m12], 1),
                            np.stack([m20, m12, f[:, 2]], 1), np.stack([m01, m12, m20], 1)], axis=0)
        v = np.concatenate([v, mid], axis=0)
    return np.ascontiguousarray(v, np.float32), np.ascontiguousarray(f, np.int32)

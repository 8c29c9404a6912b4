"""Counterpart of the reference's differentiable Python prototype (BASELINE configuration 1).

`transient_rendering_python/rendering_grad.py:16-126` renders the transient of ONE (lighting, sensor) pair by
angular sampling -- fixed ray directions from the lighting point, nearest hit, second segment visible from the
sensor, `cos(theta_2) / d_2^2` binned by `ceil((d_1 + d_2) / res) - 1` -- on torch Variables, so that
`backward()` gives the gradient with respect to the vertices.  It was written for torch <= 0.4 and does not run
on current torch (0-d `len()`, `ByteTensor ^ 1`); this module is the same estimator on current torch, batched
over directions and faces instead of the prototype's Python loop over samples.  Same names and argument order:

    angular_sampling(mesh, direction, lighting, sensor, lighting_normal, sensor_normal, opt) -> Tensor[max_distance_bin]

with `mesh.v` a float64 tensor [V,3] (requires_grad for gradients), `mesh.f` int64 [F,3], `direction` [N,3] (numpy
or tensor), `opt.{sample_num, max_distance_bin, distance_resolution, epsilon}`.  The wall normals are accepted and
ignored, as in the reference.  This is the reference's CPU "plumbing" path (it runs on whatever device `mesh.v`
lives on); the production renderer is `renderer` / `device`.

Pinned by tests/test_rendering_grad.py: forward rows against fixtures produced by the reference's numpy twin
(`rendering.py`, same formulas: rendering_grad.py:100-125 == rendering.py:72-93 with face normals recomputed from
the vertices), gradients against central finite differences of that numpy forward.
"""
import math

import torch


def _as_tensor(x, like):
    if isinstance(x, torch.Tensor):
        return x.to(dtype=like.dtype, device=like.device)
    return torch.as_tensor(x, dtype=like.dtype, device=like.device)


def intersect_ray_mesh_batch_directions(origin, direction, mesh, epsilon):
    """Moeller-Trumbore of every face against every ray `origin + t * direction[i]`
    (mesh_intersection_grad.py:3-56: edges from the THIRD vertex, u weights f[:,0], v weights f[:,1]).
    `origin` is [3] (one point for all rays) or [N,3].  Returns (hit [F,N] bool, t, u, v [F,N])."""
    v0, v1, v2 = mesh.v[mesh.f[:, 0]], mesh.v[mesh.f[:, 1]], mesh.v[mesh.f[:, 2]]
    e1, e2 = v0 - v2, v1 - v2                                     # [F,3]
    d = direction                                                 # [N,3]
    o = origin if origin.dim() == 2 else origin[None, :]          # [N or 1, 3]
    pvec = torch.cross(d[None, :, :], e2[:, None, :].expand(-1, d.shape[0], -1), dim=2)    # d x e2, [F,N,3]
    det = (e1[:, None, :] * pvec).sum(2)                          # [F,N]
    hit = det.abs() >= epsilon
    inv = 1.0 / torch.where(hit, det, torch.ones_like(det))
    tvec = o[None, :, :] - v2[:, None, :]                         # [F,N,3]
    u = (tvec * pvec).sum(2) * inv
    qvec = torch.cross(tvec, e1[:, None, :].expand_as(tvec), dim=2)
    v = (d[None, :, :] * qvec).sum(2) * inv
    t = (e2[:, None, :] * qvec).sum(2) * inv
    hit = hit & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1)
    return hit, t, u, v


def angular_sampling(mesh, direction, lighting, sensor, lighting_normal, sensor_normal, opt):
    vtx = mesh.v
    direction = _as_tensor(direction, vtx)
    lighting = _as_tensor(lighting, vtx)
    sensor = _as_tensor(sensor, vtx)
    n_bin = int(opt.max_distance_bin)
    out = torch.zeros(n_bin, dtype=vtx.dtype, device=vtx.device)
    # first segment: nearest hit by |t| (rendering_grad.py:47-63)
    hit, t, u, v = intersect_ray_mesh_batch_directions(lighting, direction, mesh, opt.epsilon)
    tabs = torch.where(hit, t.abs(), torch.full_like(t, float("inf")))
    d1, tri = tabs.min(dim=0)                                     # [N]
    any_hit = hit.any(dim=0)
    if not bool(any_hit.any()):
        return out
    cols = torch.arange(direction.shape[0], device=vtx.device)
    uu, vv = u[tri, cols], v[tri, cols]
    f = mesh.f[tri]                                               # [N,3]
    p = (1 - uu - vv)[:, None] * vtx[f[:, 2]] + uu[:, None] * vtx[f[:, 0]] + vv[:, None] * vtx[f[:, 1]]   # :74-75
    # second segment: towards the sensor; visible iff at most one face is met in (0, d2 + eps] (:78-88)
    v2 = sensor[None, :] - p
    d2 = v2.norm(dim=1)
    v2 = v2 / d2[:, None]
    hit2, t2, _, _ = intersect_ray_mesh_batch_directions(sensor, -v2, mesh, opt.epsilon)
    hit2 = hit2 & (t2 > 0) & (t2 <= d2[None, :] + opt.epsilon)
    visible = any_hit & (hit2.sum(dim=0) <= 1)
    # face normal of the hit face, recomputed from the vertices (:100-107)
    fn = torch.cross(vtx[f[:, 1]] - vtx[f[:, 0]], vtx[f[:, 2]] - vtx[f[:, 0]], dim=1)
    fn = fn / fn.norm(dim=1, keepdim=True)
    cos2 = (fn * v2).sum(dim=1).clamp(min=0.0)                    # :109-114
    dbin = torch.ceil((d1 + d2) / opt.distance_resolution).long() - 1      # :117
    keep = visible & (dbin < n_bin) & (dbin >= 0)
    val = torch.where(keep, cos2 / d2 ** 2, torch.zeros_like(cos2))
    out = out.index_add(0, dbin.clamp(0, n_bin - 1), val)         # :120-123
    return out * (2 * math.pi) / opt.sample_num                   # :125-126

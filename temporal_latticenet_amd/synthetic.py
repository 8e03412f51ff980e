"""Seeded synthetic "SemanticKITTI-shaped" sequences.

Stands in for dataloader/kitti_dataloader.py (reference kitti:100-201): one sequence is a
list of frames, every frame expressed in the coordinate system of frame 0 (kitti:122,
160-167), axis convention rotated -90 deg about x so that "up" is +y (kitti:166), one
reflectance channel in [0,1) (kitti:183-184), points shuffled (cfg `shuffle_points`),
range gated to [min_distance, cap_distance] (cfg:98-99).

The scene is an undulating ground 1.73 m below the sensor, seeded axis-aligned boxes
(cars, walls, buildings) and volumetric spheres (vegetation); a 64-beam spinning sensor
(elevation -24.8..+2 deg) is ray-cast against it.  The sensor advances 1.5 m and yaws
1 deg per frame.
"""
import numpy as np

__all__ = ["make_scene", "make_frame", "make_sequence"]

_GROUND = -1.73


def make_scene(seed=1234, nr_boxes=70, nr_trees=140, extent=55.0):
    rng = np.random.default_rng(seed)

    def place(n, clear):
        xy = rng.uniform(-extent, extent, (n, 2))
        r = np.linalg.norm(xy, axis=1)
        xy[r < clear] *= (clear / np.maximum(r[r < clear], 1e-3))[:, None]
        return xy

    c = np.empty((nr_boxes, 3)); h = np.empty((nr_boxes, 3))
    c[:, :2] = place(nr_boxes, 7.0)
    kind = rng.uniform(0, 1, nr_boxes)
    car = kind < 0.5
    h[car] = np.stack([rng.uniform(1.8, 2.4, car.sum()), rng.uniform(0.8, 1.0, car.sum()),
                       rng.uniform(0.7, 0.9, car.sum())], 1)
    nb = (~car).sum()
    h[~car] = np.stack([rng.uniform(2.0, 9.0, nb), rng.uniform(0.3, 6.0, nb), rng.uniform(1.5, 5.0, nb)], 1)
    swap = rng.uniform(0, 1, nr_boxes) < 0.5
    h[swap, 0], h[swap, 1] = h[swap, 1].copy(), h[swap, 0].copy()
    c[:, 2] = _GROUND + h[:, 2]
    tc = np.empty((nr_trees, 3))
    tc[:, :2] = place(nr_trees, 6.0)
    tr = rng.uniform(0.8, 3.0, nr_trees)
    tc[:, 2] = _GROUND + rng.uniform(1.0, 4.0, nr_trees)
    f32 = np.float32
    return {"lo": (c - h).astype(f32), "hi": (c + h).astype(f32), "tc": tc.astype(f32), "tr": tr.astype(f32)}


def _raycast(origin, dirs, scene, max_range, rng):
    """First-hit distance of the rays origin + t*dirs (float32, chunked over rays)."""
    f32 = np.float32
    origin = origin.astype(f32)
    n = dirs.shape[0]
    t_hit = np.full(n, np.inf, f32)
    lo, hi, tc, tr = scene["lo"], scene["hi"], scene["tc"], scene["tr"]
    for s in range(0, n, 32768):
        d = dirs[s:s + 32768].astype(f32)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            # ground (flat solve, relief applied to the hit point afterwards)
            tg = (f32(_GROUND) - origin[2]) / d[:, 2]
            best = np.where((d[:, 2] < 0) & (tg > 0), tg, f32(np.inf)).astype(f32)
            inv = f32(1.0) / d
            t0 = (lo[None, :, :] - origin) * inv[:, None, :]
            t1 = (hi[None, :, :] - origin) * inv[:, None, :]
            tmin = np.minimum(t0, t1).max(axis=2)
            tmax = np.maximum(t0, t1).min(axis=2)
            tb = np.where((tmax >= tmin) & (tmin > 0), tmin, f32(np.inf)).min(axis=1)
            best = np.minimum(best, tb)
            oc = origin[None, :] - tc                                   # [S,3]
            b = d @ oc.T                                                # [R,S]
            cc = (oc * oc).sum(-1) - tr * tr
            disc = b * b - cc[None, :]
            sq = np.sqrt(np.maximum(disc, 0))
            t_in, t_out = -b - sq, -b + sq
            depth = rng.uniform(0.0, 1.0, b.shape).astype(f32)
            ts = np.where((disc > 0) & (t_in > 0), t_in + depth * (t_out - t_in), f32(np.inf)).min(axis=1)
            best = np.minimum(best, ts)
        t_hit[s:s + 32768] = best
    t_hit[t_hit > max_range] = np.inf
    return t_hit


def make_frame(nr_points, frame_idx=0, seed=1234, scene=None, cap_distance=60.0,
               min_distance=3.0, step=1.5, yaw_deg=1.0):
    """Returns positions [N,3] f32 (frame-0 coordinates, +y up) and values [N,1] f32."""
    if scene is None:
        scene = make_scene(seed)
    rng = np.random.default_rng(seed + 1000 * (frame_idx + 1))
    yaw = np.deg2rad(yaw_deg * frame_idx)
    origin = np.array([step * frame_idx, 0.0, 0.0])
    pts = np.zeros((0, 3), np.float32)
    nr_rays = int(nr_points * 1.25) + 64
    while pts.shape[0] < nr_points:
        beam = rng.integers(0, 64, nr_rays)
        elev = np.deg2rad(-24.8 + (beam + 0.5) * (26.8 / 64.0))
        azim = rng.uniform(0.0, 2 * np.pi, nr_rays) + yaw
        dirs = np.stack([np.cos(elev) * np.cos(azim), np.cos(elev) * np.sin(azim), np.sin(elev)], 1)
        t = _raycast(origin, dirs, scene, 80.0, rng)
        ok = np.isfinite(t) & (t >= min_distance) & (t <= cap_distance)
        hit = origin[None, :] + dirs[ok] * t[ok, None].astype(np.float64)
        # gentle terrain relief + 1 cm range noise
        hit[:, 2] += 0.35 * np.sin(hit[:, 0] / 13.0) + 0.25 * np.cos(hit[:, 1] / 9.0)
        hit += rng.normal(0.0, 0.01, hit.shape)
        pts = np.concatenate([pts, hit.astype(np.float32)], 0)
    pts = pts[rng.permutation(pts.shape[0])[:nr_points]]
    # KITTI (x fwd, y left, z up) -> loader convention: rotate -90 deg about x => (x, z, -y)
    pos = np.ascontiguousarray(np.stack([pts[:, 0], pts[:, 2], -pts[:, 1]], 1), dtype=np.float32)
    val = rng.uniform(0.0, 1.0, (nr_points, 1)).astype(np.float32)
    return pos, val


def make_sequence(nr_points, nr_frames, seed=1234, **kw):
    scene = make_scene(seed)
    return [make_frame(nr_points, t, seed, scene, **kw) for t in range(nr_frames)]

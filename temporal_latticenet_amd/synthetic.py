"""Seeded synthetic "SemanticKITTI-shaped" sequences.

Stands in for dataloader/kitti_dataloader.py (reference kitti:100-201): one sequence is a
list of frames, every frame expressed in the coordinate system of frame 0 (kitti:122,
160-167), axis convention rotated -90 deg about x so that "up" is +y (kitti:166), one
reflectance channel in [0,1) (kitti:183-184), points shuffled (cfg `shuffle_points`),
range gated to [min_distance, cap_distance] (cfg:98-99).

The scene is calibrated against the one sizing statement the reference holds
(seq_config/lnn_train_semantic_kitti.cfg:71: SemanticKITTI scans "splat around 10k
[vertices] with sigma of 1"): `tests/test_scene_calibration.py` checks that a 120k-point
frame of the default scene hashes to 8-12k vertices at sigma = 1.0.  What it takes to get
there is what a street scene has: rolling ground 1.73 m below the sensor, a road corridor
kept clear along the drive, parked cars next to it, building fronts set back from it and
behind them (so the near-horizontal beams of the 64-beam sensor, -24.8..+2 deg, end on
structure 20-60 m away instead of leaving the scene) and volumetric vegetation (spheres
whose returns are spread over the chord of the ray).  Range statistics of a frame:
10 % / 50 % / 90 % of the points within 4.5 / 9 / 29 m.  The sensor advances 1.5 m and
yaws 1 deg per frame.
"""
import numpy as np

__all__ = ["make_scene", "make_frame", "make_sequence"]

_GROUND = -1.73
_SECTORS = 96


def make_scene(seed=1234, nr_buildings=215, nr_cars=30, nr_trees=700, extent=75.0):
    rng = np.random.default_rng(seed)

    def place(n, clear):
        """n positions outside the road corridor (|y| < clear along the drive) and the disc r < clear."""
        out = np.empty((0, 2))
        while out.shape[0] < n:
            xy = rng.uniform(-extent, extent, (2 * n, 2))
            ok = (np.abs(xy[:, 1]) > clear) | (np.abs(xy[:, 0]) > 45.0)
            ok &= np.linalg.norm(xy, axis=1) > clear
            out = np.concatenate([out, xy[ok]])
        return out[:n]

    # a few building fronts close to the road, the bulk further back
    near = nr_buildings // 30
    bc = np.concatenate([place(near, 13.0), place(nr_buildings - near, 30.0)])
    bh = np.stack([rng.uniform(3.0, 10.0, nr_buildings), rng.uniform(3.0, 10.0, nr_buildings),
                   rng.uniform(1.5, 10.0, nr_buildings)], 1)
    bh[:near, :2] = np.minimum(bh[:near, :2], 6.0)
    cc = place(nr_cars, 3.0)
    ch = np.stack([rng.uniform(1.8, 2.4, nr_cars), rng.uniform(0.8, 1.0, nr_cars), rng.uniform(0.7, 0.9, nr_cars)], 1)
    swap = rng.uniform(0, 1, nr_cars) < 0.3
    ch[swap, 0], ch[swap, 1] = ch[swap, 1].copy(), ch[swap, 0].copy()
    c2 = np.concatenate([bc, cc])
    h = np.concatenate([bh, ch])
    c = np.empty((c2.shape[0], 3))
    c[:, :2] = c2
    c[:, 2] = _GROUND + h[:, 2]
    tc = np.empty((nr_trees, 3))
    tc[:, :2] = place(nr_trees, 12.0)
    tr = rng.uniform(1.0, 3.5, nr_trees)
    tc[:, 2] = _GROUND + rng.uniform(0.3, 1.5, nr_trees) + tr
    f32 = np.float32
    return {"lo": (c - h).astype(f32), "hi": (c + h).astype(f32), "tc": tc.astype(f32), "tr": tr.astype(f32)}


def _sector_lists(origin, centres_xy, radii):
    """For each of the _SECTORS azimuth sectors seen from `origin`: the objects whose bounding circle reaches it."""
    rel = centres_xy - origin[None, :2]
    dist = np.linalg.norm(rel, axis=1)
    theta = np.arctan2(rel[:, 1], rel[:, 0])
    width = 2 * np.pi / _SECTORS
    everywhere = dist <= radii * 1.05 + 0.1
    half = np.arcsin(np.clip(radii / np.maximum(dist, 1e-6), 0.0, 1.0)) + 0.01
    lo = np.floor((theta - half) / width).astype(np.int64)
    hi = np.floor((theta + half) / width).astype(np.int64)
    lists = [[] for _ in range(_SECTORS)]
    for i in range(centres_xy.shape[0]):
        if everywhere[i] or hi[i] - lo[i] >= _SECTORS - 1:
            for s in range(_SECTORS):
                lists[s].append(i)
        else:
            for s in range(lo[i], hi[i] + 1):
                lists[s % _SECTORS].append(i)
    return [np.asarray(l, np.int64) for l in lists]


def _raycast(origin, dirs, azim, scene, max_range, rng):
    """First-hit distance of the rays origin + t*dirs (float32); rays are binned by azimuth sector and only
    tested against the objects that reach their sector."""
    f32 = np.float32
    origin = origin.astype(f32)
    n = dirs.shape[0]
    lo, hi, tc, tr = scene["lo"], scene["hi"], scene["tc"], scene["tr"]
    box_lists = _sector_lists(origin, 0.5 * (lo[:, :2] + hi[:, :2]),
                              0.5 * np.linalg.norm(hi[:, :2] - lo[:, :2], axis=1))
    sph_lists = _sector_lists(origin, tc[:, :2], tr)
    d_all = dirs.astype(f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        tg = (f32(_GROUND) - origin[2]) / d_all[:, 2]
        t_hit = np.where((d_all[:, 2] < 0) & (tg > 0), tg, f32(np.inf)).astype(f32)
    sector = np.floor(np.mod(azim, 2 * np.pi) / (2 * np.pi / _SECTORS)).astype(np.int64) % _SECTORS
    by_sector = np.argsort(sector, kind="stable")
    bounds = np.searchsorted(sector[by_sector], np.arange(_SECTORS + 1))
    for s in range(_SECTORS):
        rays = by_sector[bounds[s]:bounds[s + 1]]
        if rays.size == 0:
            continue
        d = d_all[rays]
        best = t_hit[rays]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            bl = box_lists[s]
            if bl.size:
                inv = f32(1.0) / d
                t0 = (lo[bl][None, :, :] - origin) * inv[:, None, :]
                t1 = (hi[bl][None, :, :] - origin) * inv[:, None, :]
                tmin = np.minimum(t0, t1).max(axis=2)
                tmax = np.maximum(t0, t1).min(axis=2)
                tb = np.where((tmax >= tmin) & (tmin > 0), tmin, f32(np.inf)).min(axis=1)
                best = np.minimum(best, tb)
            sl = sph_lists[s]
            if sl.size:
                oc = origin[None, :] - tc[sl]                               # [S,3]
                b = d @ oc.T                                                # [R,S]
                cc = (oc * oc).sum(-1) - tr[sl] * tr[sl]
                disc = b * b - cc[None, :]
                sq = np.sqrt(np.maximum(disc, 0))
                t_in, t_out = -b - sq, -b + sq
                depth = rng.uniform(0.0, 1.0, b.shape).astype(f32)
                ts = np.where((disc > 0) & (t_in > 0), t_in + depth * (t_out - t_in), f32(np.inf)).min(axis=1)
                best = np.minimum(best, ts)
        t_hit[rays] = best
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
        t = _raycast(origin, dirs, azim, scene, 80.0, rng)
        ok = np.isfinite(t) & (t >= min_distance) & (t <= cap_distance)
        hit = origin[None, :] + dirs[ok] * t[ok, None].astype(np.float64)
        # rolling terrain (a gentle grade along the drive, a cross fall, short-wave relief) + 1 cm range noise
        hit[:, 2] += (0.35 * np.sin(hit[:, 0] / 13.0) + 0.25 * np.cos(hit[:, 1] / 9.0)
                      + 0.015 * hit[:, 0] + 0.9 * np.sin(hit[:, 1] / 38.0))
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

"""SemanticKITTI on-disk formats used around the hot path: what
dataloader/kitti_dataloader.py reads (kitti:42-47, 100-201, 205-291) and what test_ln.py writes (test:219-231),
so that real scans, labels, poses and predictions can be fed to / taken from the MI355X path.

Formats
  velodyne/NNNNNN.bin   float32 [N,4]  x, y, z, reflectance                       (kitti:129-131)
  labels/NNNNNN.label   uint32 [N]     low 16 bits = class, high 16 = instance     (kitti:281-291)
  calib.txt             "key: 12 floats" rows of a 3x4 matrix, key "Tr" = velodyne->camera   (kitti:205-228)
  poses.txt             12 floats per scan (camera pose); velo pose = Tr^-1 * P * Tr (kitti:230-258)
  predictions/NNNNNN.label  one decimal class id per line (text), last cloud of the sequence only (test:219-231)
"""
import os

import numpy as np

__all__ = ["read_scan", "write_scan", "make_remap_lut", "read_labels", "write_labels", "parse_calibration",
           "parse_poses", "window_indices", "rotation_minus90_x", "to_first_frame", "range_gate", "load_sequence",
           "accumulate", "write_prediction_labels", "read_prediction_labels"]


def read_scan(path):
    a = np.fromfile(path, dtype=np.float32)
    if a.size % 4:
        raise ValueError("%s: size %d is not a multiple of 4 floats" % (path, a.size))
    a = a.reshape(-1, 4)
    return np.ascontiguousarray(a[:, :3]), np.ascontiguousarray(a[:, 3])


def write_scan(path, xyz, reflectance):
    np.concatenate([np.asarray(xyz, np.float32), np.asarray(reflectance, np.float32).reshape(-1, 1)], 1).tofile(path)


def make_remap_lut(learning_map):
    """dict raw label -> training class, as a lookup table with 100 spare entries (kitti:42-47)"""
    maxkey = max(learning_map.keys())
    lut = np.zeros(maxkey + 100, dtype=np.int32)
    lut[list(learning_map.keys())] = list(learning_map.values())
    return lut


def read_labels(path, remap_lut=None):
    raw = np.fromfile(path, dtype="<u4")            # little-endian u32 per point (kitti:281-291 reads it as u16 pairs)
    sem = (raw & 0xFFFF).astype(np.int64)           # low 16 bits: class, high 16: instance
    return remap_lut[sem].astype(np.int64) if remap_lut is not None else sem


def write_labels(path, sem, instance=None):
    sem = np.asarray(sem, np.uint32)
    inst = np.zeros_like(sem) if instance is None else np.asarray(instance, np.uint32)
    ((inst << 16) | (sem & 0xFFFF)).astype(np.uint32).tofile(path)


def _rows_to_4x4(values):
    m = np.zeros((4, 4))
    m[0, :], m[1, :], m[2, :] = values[0:4], values[4:8], values[8:12]
    m[3, 3] = 1.0
    return m


def parse_calibration(path):
    calib = {}
    with open(path) as f:
        for line in f:
            if ":" not in line:
                continue
            key, content = line.strip().split(":", 1)
            calib[key] = _rows_to_4x4([float(v) for v in content.split()])
    return calib


def parse_poses(path, calibration):
    """velodyne poses: Tr^-1 * P_cam * Tr per scan"""
    tr = calibration["Tr"]
    tr_inv = np.linalg.inv(tr)
    poses = []
    with open(path) as f:
        for line in f:
            vals = [float(v) for v in line.split()]
            if len(vals) == 12:
                poses.append(tr_inv @ _rows_to_4x4(vals) @ tr)
    return poses


def window_indices(index, frames_per_seq, cloud_scope):
    """scan ids of the sequence ending at `index`: stride cloud_scope, clamped at 0 (kitti:100, 116)"""
    return np.maximum((np.arange(frames_per_seq) - (frames_per_seq - 1)) * cloud_scope + index, 0)


def rotation_minus90_x():
    c, s = 0.0, -1.0
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float64)


def to_first_frame(xyz, pose, pose_first):
    """velodyne points of a scan -> coordinate system of the sequence's FIRST scan, then the -90 deg rotation about x
    that makes +y the up axis (kitti:160-167)"""
    h = np.ones((4, xyz.shape[0]))
    h[:3] = np.asarray(xyz, np.float64).T
    w = rotation_minus90_x() @ (np.linalg.inv(pose_first) @ (pose @ h))
    return np.ascontiguousarray((w[:3] / w[3]).T.astype(np.float32))


def range_gate(xyz, cap_distance=60.0, min_distance=3.0):
    """boolean mask of the points kept by cap_distance / min_distance (kitti:142-154; negative disables)"""
    r = np.linalg.norm(xyz, axis=1)
    keep = np.ones(xyz.shape[0], bool)
    if cap_distance >= 0:
        keep &= r < cap_distance
    if min_distance >= 0:
        keep &= r > min_distance
    return keep


def load_sequence(data_dir, seq, index, frames_per_seq=4, cloud_scope=3, remap_lut=None, cap_distance=60.0,
                  min_distance=3.0, rng=None, with_labels=True, split="train"):
    """The loader's output contract (kitti:100-197): lists of positions [N_i,3] f32, values [N_i,1] f32,
    labels [N_i] i64, paths, lengths — all frames in the first frame's coordinates.
    As in the reference, the range gate (kitti:142-154) and the point shuffle (kitti:172-177) apply to the "train" split
    only: a valid / test cloud keeps every point in file order, so that the per-point prediction file written by
    write_prediction_labels lines up with the .bin scan; the "test" split has no label files (kitti:135-136)."""
    is_training = split == "train"
    if split == "test":
        with_labels = False
    sdir = os.path.join(data_dir, "sequences", "%02d" % seq)
    poses = parse_poses(os.path.join(sdir, "poses.txt"), parse_calibration(os.path.join(sdir, "calib.txt")))
    ids = window_indices(index, frames_per_seq, cloud_scope)
    out = ([], [], [], [], [])
    for i in ids:
        path = os.path.join(sdir, "velodyne", "%06d.bin" % i)
        xyz, refl = read_scan(path)
        lab = read_labels(os.path.join(sdir, "labels", "%06d.label" % i), remap_lut) if with_labels \
            else np.zeros(xyz.shape[0], np.int64)
        if is_training:
            keep = range_gate(xyz, cap_distance, min_distance)
            xyz, refl, lab = xyz[keep], refl[keep], lab[keep]
        pos = to_first_frame(xyz, poses[i], poses[ids[0]])
        if rng is not None and is_training:              # shuffle_points (kitti:172-177)
            perm = rng.permutation(pos.shape[0])
            pos, refl, lab = pos[perm], refl[perm], lab[perm]
        out[0].append(pos)
        out[1].append(refl.reshape(-1, 1).astype(np.float32))
        out[2].append(lab)
        out[3].append(path)
        out[4].append(pos.shape[0])
    return out


def accumulate(positions, values, labels):
    """accumulate_clouds: the whole sequence as ONE cloud (kitti:198-201)"""
    return np.concatenate(positions, 0), np.concatenate(values, 0), np.concatenate(labels, 0)


def write_prediction_labels(path, predicted_classes, len_last_cloud=None):
    """one class id per line; with accumulated clouds only the points of the last cloud are written (test:219-231)"""
    p = np.asarray(predicted_classes).reshape(-1).astype(np.uint32)
    if len_last_cloud is not None:
        p = p[-int(len_last_cloud):]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write("".join("%d\n" % v for v in p))


def read_prediction_labels(path):
    return np.fromfile(path, dtype=np.uint32, sep="\n")

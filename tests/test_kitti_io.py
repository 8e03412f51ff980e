"""CPU: SemanticKITTI file formats (scan / label / calib / poses / prediction writer) round trips and the
first-frame coordinate transform."""
import os

import numpy as np

from temporal_latticenet_amd import kitti_io as K


def _make_dataset(tmp, nr_scans=8, n=500):
    rng = np.random.default_rng(0)
    sdir = os.path.join(tmp, "sequences", "03")
    os.makedirs(os.path.join(sdir, "velodyne"))
    os.makedirs(os.path.join(sdir, "labels"))
    tr = np.eye(4)
    tr[:3, :3] = [[0, -1, 0], [0, 0, -1], [1, 0, 0]]       # a velodyne->camera style axis permutation
    tr[:3, 3] = [0.1, -0.2, 0.3]
    with open(os.path.join(sdir, "calib.txt"), "w") as f:
        f.write("P0: " + " ".join(["1"] * 12) + "\n")
        f.write("Tr: " + " ".join("%.9f" % v for v in tr[:3].reshape(-1)) + "\n")
    scans, labels, cam_poses = [], [], []
    with open(os.path.join(sdir, "poses.txt"), "w") as f:
        for i in range(nr_scans):
            a = 0.05 * i
            p = np.eye(4)
            p[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
            p[:3, 3] = [0.0, 0.0, 1.5 * i]
            cam_poses.append(p)
            f.write(" ".join("%.9f" % v for v in p[:3].reshape(-1)) + "\n")
            xyz = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
            refl = rng.uniform(0, 1, n).astype(np.float32)
            sem = rng.choice([0, 10, 40, 252], n)
            inst = rng.integers(0, 1000, n)
            K.write_scan(os.path.join(sdir, "velodyne", "%06d.bin" % i), xyz, refl)
            K.write_labels(os.path.join(sdir, "labels", "%06d.label" % i), sem, inst)
            scans.append((xyz, refl))
            labels.append(sem)
    return sdir, tr, cam_poses, scans, labels


def test_scan_and_label_round_trip(tmp_path):
    sdir, tr, cam, scans, labels = _make_dataset(str(tmp_path))
    xyz, refl = K.read_scan(os.path.join(sdir, "velodyne", "000002.bin"))
    assert np.array_equal(xyz, scans[2][0]) and np.array_equal(refl, scans[2][1])
    lut = K.make_remap_lut({0: 0, 10: 1, 40: 9, 252: 20})
    assert lut.shape[0] == 352
    lab = K.read_labels(os.path.join(sdir, "labels", "000002.label"), lut)
    assert np.array_equal(lab, lut[labels[2]])              # instance ids in the high 16 bits are ignored
    assert set(np.unique(lab)) <= {0, 1, 9, 20}


def test_poses_and_first_frame_transform(tmp_path):
    sdir, tr, cam, scans, labels = _make_dataset(str(tmp_path))
    calib = K.parse_calibration(os.path.join(sdir, "calib.txt"))
    np.testing.assert_allclose(calib["Tr"], tr, atol=1e-8)
    poses = K.parse_poses(os.path.join(sdir, "poses.txt"), calib)
    assert len(poses) == 8
    np.testing.assert_allclose(poses[3], np.linalg.inv(tr) @ cam[3] @ tr, atol=1e-7)
    # a scan expressed in its own frame only gets the -90 deg rotation about x: (x, y, z) -> (x, z, -y)
    xyz = scans[5][0]
    same = K.to_first_frame(xyz, poses[5], poses[5])
    np.testing.assert_allclose(same, np.stack([xyz[:, 0], xyz[:, 2], -xyz[:, 1]], 1), atol=1e-4)
    # a world-fixed point seen from two scans lands on the same first-frame coordinates
    world = np.array([[3.0, -2.0, 7.0, 1.0]]).T
    p_a = (np.linalg.inv(poses[2]) @ world)[:3].T
    p_b = (np.linalg.inv(poses[6]) @ world)[:3].T
    np.testing.assert_allclose(K.to_first_frame(p_a, poses[2], poses[1]), K.to_first_frame(p_b, poses[6], poses[1]),
                               atol=1e-4)


def test_sequence_window_loader_contract_and_accumulate(tmp_path):
    sdir, tr, cam, scans, labels = _make_dataset(str(tmp_path))
    assert K.window_indices(7, 4, 3).tolist() == [0, 1, 4, 7]      # (arange(4)-3)*3+7, clamped at 0
    assert K.window_indices(2, 4, 3).tolist() == [0, 0, 0, 2]
    lut = K.make_remap_lut({0: 0, 10: 1, 40: 9, 252: 20})
    pos, val, lab, paths, lens = K.load_sequence(str(tmp_path), 3, 7, 4, 3, lut, rng=np.random.default_rng(1))
    assert len(pos) == 4 and [p.shape[0] for p in pos] == lens
    for p, v, l in zip(pos, val, lab):
        assert p.dtype == np.float32 and p.shape[1] == 3 and v.shape == (p.shape[0], 1) and l.shape == (p.shape[0],)
    assert paths[-1].endswith("000007.bin")
    r = np.linalg.norm(scans[7][0], axis=1)
    assert lens[-1] == int(((r < 60.0) & (r > 3.0)).sum())          # range gate
    P, V, L = K.accumulate(pos, val, lab)
    assert P.shape[0] == sum(lens)
    out = os.path.join(str(tmp_path), "pred", "sequences", "03", "predictions", "000007.label")
    pred = np.arange(P.shape[0]) % 26
    K.write_prediction_labels(out, pred, len_last_cloud=lens[-1])
    back = K.read_prediction_labels(out)
    assert np.array_equal(back, pred[-lens[-1]:].astype(np.uint32))
    assert open(out).readline().strip().isdigit()                   # text, one label per line (test_ln.py:228-231)

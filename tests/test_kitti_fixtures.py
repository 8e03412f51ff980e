"""CPU: the SemanticKITTI readers against byte-level samples built BY HAND from the format rules of the reference's
loader — not through this repo's own writers:
  velodyne/*.bin   little-endian float32 x, y, z, reflectance per point   (dataloader/kitti_dataloader.py:129-131)
  labels/*.label   little-endian uint32 per point, low 16 bits = class, high 16 = instance id; the loader reads the
                   file as uint16 and keeps every second value, then maps it through learning_map + a 100-entry
                   margin (kitti:42-47, 281-291)
  calib.txt        "KEY: 12 numbers" = rows of a 3x4 matrix, bottom row (0 0 0 1) implied (kitti:205-228)
  poses.txt        12 numbers per scan, camera pose; velodyne pose = Tr^-1 P Tr (kitti:230-258)
  predictions      one decimal class id per line, the LAST cloud's points only (test_ln.py:219-231)"""
import os
import struct

import numpy as np

from temporal_latticenet_amd import kitti_io as K

# three points of scan 0, two of scan 1 (x, y, z, reflectance)
SCAN0 = [(10.0, 0.0, -1.5, 0.25), (0.0, -20.0, 0.5, 0.5), (1.0, 1.0, 1.0, 0.0)]
SCAN1 = [(70.0, 0.0, 0.0, 1.0), (5.0, 5.0, -1.0, 0.75)]
# (class, instance) per point
LAB0 = [(40, 0), (252, 7), (10, 65535)]
LAB1 = [(0, 3), (48, 1)]
LEARNING_MAP = {0: 0, 10: 1, 40: 9, 48: 11, 252: 20}
# Tr: velodyne (x fwd, y left, z up) -> camera (x right, y down, z fwd), with an offset
TR_ROWS = "0 -1 0 0.5   0 0 -1 0.25   1 0 0 -0.125"
# camera poses: identity, then 2 m forward (camera z)
POSES_TXT = "1 0 0 0 0 1 0 0 0 0 1 0\n1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0 2.0\n"


def _write_fixture(tmp):
    sdir = os.path.join(tmp, "sequences", "07")
    os.makedirs(os.path.join(sdir, "velodyne"))
    os.makedirs(os.path.join(sdir, "labels"))
    for i, (scan, lab) in enumerate(((SCAN0, LAB0), (SCAN1, LAB1))):
        with open(os.path.join(sdir, "velodyne", "%06d.bin" % i), "wb") as f:
            for pt in scan:
                f.write(struct.pack("<4f", *pt))
        with open(os.path.join(sdir, "labels", "%06d.label" % i), "wb") as f:
            for cls, inst in lab:
                f.write(struct.pack("<HH", cls, inst))        # low half first: little-endian uint32 = inst << 16 | cls
    with open(os.path.join(sdir, "calib.txt"), "w") as f:
        f.write("P0: 7.0e+02 0 6.0e+02 0 0 7.0e+02 1.8e+02 0 0 0 1 0\n")
        f.write("Tr: " + TR_ROWS + "\n")
    with open(os.path.join(sdir, "poses.txt"), "w") as f:
        f.write(POSES_TXT)
    return sdir


def test_scan_bytes(tmp_path):
    sdir = _write_fixture(str(tmp_path))
    xyz, refl = K.read_scan(os.path.join(sdir, "velodyne", "000000.bin"))
    assert xyz.dtype == np.float32 and xyz.shape == (3, 3) and refl.shape == (3,)
    assert xyz.tolist() == [[10.0, 0.0, -1.5], [0.0, -20.0, 0.5], [1.0, 1.0, 1.0]]
    assert refl.tolist() == [0.25, 0.5, 0.0]


def test_label_low_16_bits_and_learning_map(tmp_path):
    sdir = _write_fixture(str(tmp_path))
    raw = K.read_labels(os.path.join(sdir, "labels", "000000.label"))
    assert raw.tolist() == [40, 252, 10]                      # the instance ids (0, 7, 65535) in the high half are dropped
    lut = K.make_remap_lut(LEARNING_MAP)
    assert lut.shape == (252 + 100,) and lut.dtype == np.int32 and lut[253] == 0      # kitti:42-47, the +100 margin
    assert K.read_labels(os.path.join(sdir, "labels", "000000.label"), lut).tolist() == [9, 20, 1]
    assert K.read_labels(os.path.join(sdir, "labels", "000001.label"), lut).tolist() == [0, 11]
    # the reference's own decoding of the same bytes: uint16 pairs, every second value (kitti:284-288)
    as_u16 = np.fromfile(os.path.join(sdir, "labels", "000000.label"), dtype=np.uint16)
    assert as_u16[0::2].tolist() == raw.tolist()


def test_calibration_and_poses(tmp_path):
    sdir = _write_fixture(str(tmp_path))
    calib = K.parse_calibration(os.path.join(sdir, "calib.txt"))
    assert set(calib) == {"P0", "Tr"}
    tr = np.array([[0, -1, 0, 0.5], [0, 0, -1, 0.25], [1, 0, 0, -0.125], [0, 0, 0, 1.0]])
    assert np.array_equal(calib["Tr"], tr)
    poses = K.parse_poses(os.path.join(sdir, "poses.txt"), calib)
    assert len(poses) == 2
    np.testing.assert_allclose(poses[0], np.eye(4), atol=1e-12)
    # 2 m along the camera's z axis is 2 m along the velodyne's x axis (Tr maps velodyne x to camera z)
    want = np.eye(4)
    want[0, 3] = 2.0
    np.testing.assert_allclose(poses[1], want, atol=1e-12)


def test_sequence_contract_per_split(tmp_path):
    sdir = _write_fixture(str(tmp_path))
    lut = K.make_remap_lut(LEARNING_MAP)
    root = str(tmp_path)
    # valid split: no range gate, no shuffle (kitti:142, 149, 172 are `and is_training`): every point, file order
    pos, val, lab, paths, lens = K.load_sequence(root, 7, 1, frames_per_seq=2, cloud_scope=1, remap_lut=lut, split="valid",
                                                 rng=np.random.default_rng(0))
    assert lens == [3, 2] and [os.path.basename(p) for p in paths] == ["000000.bin", "000001.bin"]
    # frame 0 in its own coordinates: only the -90 deg rotation about x, (x, y, z) -> (x, z, -y)  (kitti:166)
    np.testing.assert_allclose(pos[0], [[10.0, -1.5, 0.0], [0.0, 0.5, 20.0], [1.0, 1.0, -1.0]], atol=1e-6)
    # frame 1 sits 2 m further along x in frame-0 coordinates
    np.testing.assert_allclose(pos[1], [[72.0, 0.0, 0.0], [7.0, -1.0, -5.0]], atol=1e-6)
    assert val[0].shape == (3, 1) and val[0].dtype == np.float32 and val[1][:, 0].tolist() == [1.0, 0.75]
    assert lab[0].tolist() == [9, 20, 1] and lab[1].tolist() == [0, 11]
    # train split: points outside (min_distance, cap_distance) = (3, 60) m of THEIR OWN sensor are dropped (kitti:142-154)
    pos_t, val_t, lab_t, _, lens_t = K.load_sequence(root, 7, 1, frames_per_seq=2, cloud_scope=1, remap_lut=lut,
                                                     split="train")
    assert lens_t == [2, 1]                                   # (1,1,1) is closer than 3 m, (70,0,0) further than 60 m
    assert lab_t[0].tolist() == [9, 20] and lab_t[1].tolist() == [11]
    # test split: no label files are read (kitti:135-136)
    os.remove(os.path.join(sdir, "labels", "000000.label"))
    os.remove(os.path.join(sdir, "labels", "000001.label"))
    pos_x, _, lab_x, _, lens_x = K.load_sequence(root, 7, 1, frames_per_seq=2, cloud_scope=1, remap_lut=lut, split="test")
    assert lens_x == [3, 2] and lab_x[0].tolist() == [0, 0, 0]
    # the window clamps at scan 0 (kitti:116): index 0 with two frames reads scan 0 twice
    assert K.window_indices(0, 2, 3).tolist() == [0, 0] and K.window_indices(7, 4, 3).tolist() == [0, 1, 4, 7]


def test_prediction_file_is_text_of_the_last_cloud(tmp_path):
    """test_ln.py:219-231: argmax classes, only the last len_seq[-1] points (accumulate_clouds concatenates the whole
    sequence, kitti:198-201), one decimal per line"""
    pred = np.array([3, 3, 3, 25, 0, 17])                     # an accumulated cloud of 3 + 3 points
    path = os.path.join(str(tmp_path), "sequences", "11", "predictions", "000042.label")
    K.write_prediction_labels(path, pred, len_last_cloud=3)
    with open(path, "rb") as f:
        assert f.read() == b"25\n0\n17\n"
    assert K.read_prediction_labels(path).tolist() == [25, 0, 17]       # remap script: np.fromfile(dtype=uint32, sep="\n")
    K.write_prediction_labels(path, pred)
    with open(path, "rb") as f:
        assert f.read() == b"3\n3\n3\n25\n0\n17\n"

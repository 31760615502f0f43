"""Host-side loader of the replay harness (iv_slam_amd/kitti.py): file listing quirks of the
reference's driver, settings keys, PNG decoding (all five filter types), grey conversion."""
import os
import struct
import zlib

import numpy as np
import pytest

from iv_slam_amd import kitti

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _png_with_filters(path, img):
    """test-side encoder that cycles through PNG filter types 0..4 row by row"""
    a = np.ascontiguousarray(img, np.uint8)
    h, w = a.shape[:2]; bpp = 1 if a.ndim == 2 else 3
    rows = a.reshape(h, -1).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(rows.shape[1], np.int32)
    for y in range(h):
        cur = rows[y]; ft = y % 5
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul; pa = np.abs(p - left); pb = np.abs(p - prev); pc = np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            f = cur - pred
        raw.append(ft); raw += (f & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)

    comp = zlib.compress(bytes(raw))
    with open(path, "wb") as fo:
        fo.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0 if bpp == 1 else 2, 0, 0, 0)) +
                 chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b""))


def test_png_round_trips(tmp_path):
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, size=(23, 41)).astype(np.uint8)
    c = rng.integers(0, 256, size=(17, 29, 3)).astype(np.uint8)              # B,G,R as cv::imread returns it
    kitti.imwrite(str(tmp_path / "g.png"), g); kitti.imwrite(str(tmp_path / "c.png"), c)
    assert np.array_equal(kitti.imread(str(tmp_path / "g.png")), g)
    assert np.array_equal(kitti.imread(str(tmp_path / "c.png")), c)
    _png_with_filters(str(tmp_path / "gf.png"), g)
    assert np.array_equal(kitti.imread(str(tmp_path / "gf.png")), g)
    _png_with_filters(str(tmp_path / "cf.png"), c[..., ::-1])                # the file stores R,G,B
    assert np.array_equal(kitti.imread(str(tmp_path / "cf.png")), c)
    (tmp_path / "bad.png").write_bytes(b"not a png")
    with pytest.raises(ValueError):
        kitti.imread(str(tmp_path / "bad.png"))
    # 16-bit files are refused, not silently truncated
    hdr = struct.pack(">IIBBBBB", 2, 2, 16, 0, 0, 0, 0)
    body = b"\x89PNG\r\n\x1a\n" + struct.pack(">I", 13) + b"IHDR" + hdr + struct.pack(">I", zlib.crc32(b"IHDR" + hdr)) + \
        struct.pack(">I", 0) + b"IDAT" + struct.pack(">I", zlib.crc32(b"IDAT"))
    (tmp_path / "d16.png").write_bytes(body)
    with pytest.raises(ValueError):
        kitti.imread(str(tmp_path / "d16.png"))


def test_to_gray_fixed_point():
    rng = np.random.default_rng(4)
    bgr = rng.integers(0, 256, size=(9, 11, 3)).astype(np.uint8)
    b, g, r = [bgr[..., k].astype(np.int64) for k in range(3)]
    # OpenCV <= 3.x: 14-bit coefficients; OpenCV 4.x (the default, like every other frozen primitive): 15-bit
    assert np.array_equal(kitti.to_gray(bgr, rgb=False, cv3=True), ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8))
    assert np.array_equal(kitti.to_gray(bgr, rgb=True, cv3=True), ((b * 4899 + g * 9617 + r * 1868 + 8192) >> 14).astype(np.uint8))
    assert np.array_equal(kitti.to_gray(bgr, rgb=False), ((r * 9798 + g * 19235 + b * 3735 + 16384) >> 15).astype(np.uint8))
    assert np.array_equal(kitti.to_gray(bgr, rgb=True), ((b * 9798 + g * 19235 + r * 3735 + 16384) >> 15).astype(np.uint8))
    white = np.full((2, 2, 3), 255, np.uint8)
    assert (kitti.to_gray(white, True) == 255).all() and (kitti.to_gray(white, False, cv3=True) == 255).all()
    grey = rng.integers(0, 256, size=(5, 5)).astype(np.uint8)
    assert kitti.to_gray(grey, True) is grey


def _sequence(tmp_path, first, n_files, n_times):
    for d in ("image_0", "image_1"):
        os.makedirs(tmp_path / d)
        for i in range(first, first + n_files):
            (tmp_path / d / ("%06d.png" % i)).write_bytes(b"")
    with open(tmp_path / "times.txt", "w") as f:
        for i in range(n_times):
            f.write("%e\n\n" % (0.1 * i))                                   # empty lines are skipped


def test_listing_from_zero(tmp_path):
    _sequence(tmp_path, 0, 5, 5)
    left, right, ts = kitti.LoadImages(str(tmp_path))
    assert len(ts) == 5 and abs(ts[3] - 0.3) < 1e-12
    assert [os.path.basename(p) for p in left] == ["%06d.png" % i for i in range(5)]
    assert all(os.path.dirname(p).endswith("image_1") for p in right)
    assert kitti.GetSmallestImgIdx(str(tmp_path / "image_0")) == 0


def test_listing_offset_quirk(tmp_path):
    # files start at 000003: LoadImages fills entries 3.. with numbers i + 3 and leaves 0..2 empty (stereo_kitti.cc:645-651);
    # LoadImagesWithGT's variant fills every entry (:745-751)
    _sequence(tmp_path, 3, 6, 6)
    left, _, _ = kitti.LoadImages(str(tmp_path))
    assert left[:3] == ["", "", ""] and [os.path.basename(p) for p in left[3:]] == ["000006.png", "000007.png", "000008.png"]
    left, _, _ = kitti.LoadImages(str(tmp_path), first_entry_from_smallest=False)
    assert [os.path.basename(p) for p in left] == ["%06d.png" % (i + 3) for i in range(6)]


def test_qual_file_names_and_poses(tmp_path):
    os.makedirs(tmp_path / "q")
    for i in (0, 1, 4):
        (tmp_path / "q" / ("%06d.jpg" % i)).write_bytes(b"")
    names, found = kitti.GetImageQualFileNames(str(tmp_path / "q"), 6)
    assert found == 3 and [bool(x) for x in names] == [True, True, False, False, True, False]
    with pytest.raises(ValueError):
        kitti.GetImageQualFileNames(str(tmp_path / "q"), 4)                 # index 4 is not < 4
    with open(tmp_path / "poses.txt", "w") as f:
        f.write("1 0 0 0.5 0 1 0 -2 0 0 1 3.25\n\n0 -1 0 1 1 0 0 2 0 0 1 3\n")
    P = kitti.LoadPoses(str(tmp_path / "poses.txt"))
    assert P.shape == (2, 4, 4) and P.dtype == np.float32
    assert P[0, 0, 3] == 0.5 and P[0, 2, 3] == 3.25 and P[1, 0, 1] == -1 and (P[:, 3] == [0, 0, 0, 1]).all()


def test_settings_keys(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import replay_kitti
    p = tmp_path / "s.yaml"
    p.write_text(replay_kitti.SYNTH_SETTINGS)
    S = kitti.Settings.load(str(p))
    assert S.extractor_params() == (600, 1.2, 8, 20, 7, False)
    bf, b = S.stereo()
    assert abs(bf - 386.1448) < 1e-9 and abs(b - 386.1448 / 718.856) < 1e-12
    K, D, R, P, size = S.rectification("LEFT")
    assert K.shape == (3, 3) and D.shape == (5,) and R.shape == (3, 3) and P.shape == (3, 4) and size == (640, 240)
    assert K.dtype == np.float64 and K[0, 2] == 322.0 and P[0, 3] == 0.0 and S.rectification("RIGHT")[3][0, 3] == -386.1448


def test_reference_settings_files_parse():
    """The reference's own FileStorage files, where the reference tree is mounted (not on the GPU box)."""
    d = "/root/reference/introspective_ORB_SLAM/Examples/Stereo"
    if not os.path.isdir(d):
        pytest.skip("reference tree not mounted")
    S = kitti.Settings.load(os.path.join(d, "KITTI00-02.yaml"))
    assert S.extractor_params()[:5] == (2000, 1.2, 8, 20, 7) or S.extractor_params()[2:5] == (8, 20, 7)
    assert abs(S.stereo()[0] - 386.1448) < 1e-6
    J = kitti.Settings.load(os.path.join(d, "jackal_visual_odom_stereo_inference.yaml"))
    K, D, R, P, size = J.rectification("LEFT")
    assert size == (960, 600) and K.shape == (3, 3) and P.shape == (3, 4) and abs(R[0, 0] - 0.99994) < 1e-9 and len(D) == 5

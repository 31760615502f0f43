"""KITTI-layout sequence loading for the replay harness: the file listing, settings keys and image
handling of the reference's stereo driver, host side only (numpy + zlib; OpenCV is not in this image).

  LoadImages               introspective_ORB_SLAM/Examples/Stereo/stereo_kitti.cc:618-654
  GetSmallestImgIdx        :868-889
  GetImageQualFileNames    :826-866
  LoadImagesWithGT / LoadPoses   :712-778 (12-float pose rows -> 4x4 float matrices)
  Settings                 OpenCV FileStorage YAML; keys of src/Tracking.cc:100-171 and stereo_kitti.cc:250-272
  imread / to_gray         cv::imread(..., UNCHANGED) for 8-bit PNGs; cvtColor(RGB2GRAY / BGR2GRAY) of Tracking.cc:278-291
"""
import os
import re
import struct
import zlib

import numpy as np


def _atoi_prefix(name, n):
    """atoi() on the first n characters of a directory entry, as GetSmallestImgIdx / GetImageQualFileNames do."""
    m = re.match(r"\s*[+-]?\d+", name[:n])
    return int(m.group(0)) if m else 0


def GetSmallestImgIdx(directory, prefix_length=6):
    """Smallest numeric prefix among the entries of `directory` (INT_MAX when it is empty)."""
    smallest = 2 ** 31 - 1
    for name in os.listdir(directory):
        smallest = min(smallest, _atoi_prefix(name, prefix_length))
    return smallest


def LoadTimes(path_to_sequence):
    """First number of every non-empty line of times.txt."""
    ts = []
    with open(os.path.join(path_to_sequence, "times.txt")) as f:
        for line in f:
            if line.strip():
                ts.append(float(line.split()[0]))
    return ts


def LoadImages(path_to_sequence, first_entry_from_smallest=True):
    """(left files, right files, timestamps).  LoadImages / LoadImagesWithQual (:645-651, :683-689) fill entry i, for i
    from the smallest index on disk, with image number i + smallest and leave the entries before it empty;
    LoadImagesWithGT (:745-751, first_entry_from_smallest=False) fills every entry.  For sequences that start at
    000000 -- every KITTI odometry sequence -- both are the plain listing."""
    ts = LoadTimes(path_to_sequence)
    left_dir = os.path.join(path_to_sequence, "image_0"); right_dir = os.path.join(path_to_sequence, "image_1")
    n = len(ts)
    left = [""] * n; right = [""] * n
    s = GetSmallestImgIdx(left_dir, 6)
    for i in range(s if first_entry_from_smallest else 0, n):
        name = "%06d.png" % (i + s)
        left[i] = os.path.join(left_dir, name); right[i] = os.path.join(right_dir, name)
    return left, right, ts


def GetImageQualFileNames(directory, size):
    """Predicted cost images named by frame number (%06d.*): list of `size` paths, "" where a frame has none;
    second result = files found.  A number >= size is an error (CHECK_LT in the reference)."""
    out = [""] * size
    found = 0
    for name in os.listdir(directory):
        k = _atoi_prefix(name, 6)
        if not k < size:
            raise ValueError("cost image %s: index outside [0,%d)" % (name, size))
        out[k] = os.path.join(directory, name)
        found += 1
    return out, found


def LoadPoses(path):
    """Ground-truth poses (:752-768): every non-empty line holds 12 space-separated floats = the top 3 rows of a 4x4
    camera pose (stof, CV_32F; last row 0 0 0 1) -> [N][4][4] float32."""
    poses = []
    with open(path) as f:
        for line in f:
            v = line.split()
            if v:
                if len(v) < 12:
                    raise ValueError("pose row with %d values" % len(v))
                m = np.eye(4, dtype=np.float32)
                m[:3, :] = np.asarray([float(x) for x in v[:12]], np.float32).reshape(3, 4)
                poses.append(m)
    return np.stack(poses) if poses else np.zeros((0, 4, 4), np.float32)


# ---- settings ------------------------------------------------------------------------------------
class Settings(dict):
    """An OpenCV FileStorage YAML file as a dict: scalars -> float / int / str, !!opencv-matrix -> numpy array."""

    @classmethod
    def load(cls, path):
        import yaml

        class L(yaml.SafeLoader):
            pass

        def matrix(loader, node):
            m = loader.construct_mapping(node, deep=True)
            dt = {"d": np.float64, "f": np.float32, "i": np.int32, "u": np.uint8}[str(m["dt"])[-1]]
            return np.asarray(m["data"], dt).reshape(int(m["rows"]), int(m["cols"]))

        L.add_constructor("tag:yaml.org,2002:opencv-matrix", matrix)
        with open(path) as f:
            text = f.read()
        text = re.sub(r"^%YAML[:\s]*1\.0\s*\n", "", text, count=1)      # FileStorage's header is not a YAML 1.1 directive
        text = text.replace("\t", " ")
        text = re.sub(r"^(\s*[A-Za-z_][\w.]*):(?=\S)", r"\1: ", text, flags=re.M)   # FileStorage accepts "key:value"
        return cls(yaml.load(text, Loader=L) or {})

    def extractor_params(self):
        """ORBextractor(nFeatures, scaleFactor, nLevels, iniThFAST, minThFAST[, enableIntrospection]) (Tracking.cc:160-191)."""
        return (int(self["ORBextractor.nFeatures"]), float(self["ORBextractor.scaleFactor"]), int(self["ORBextractor.nLevels"]),
                int(self["ORBextractor.iniThFAST"]), int(self["ORBextractor.minThFAST"]),
                bool(int(self.get("ORBextractor.enableIntrospection", 0))))

    def stereo(self):
        """(bf, b) for ComputeStereoMatches: mbf and mb = mbf / fx (Tracking.cc:125, Frame.cc:208-225)."""
        bf = float(self["Camera.bf"]); fx = float(self["Camera.fx"])
        return bf, bf / fx

    def rectification(self, side):
        """(K, D, R, P, (width, height)) of LEFT / RIGHT (stereo_kitti.cc:250-272)."""
        return (self[side + ".K"], self[side + ".D"].ravel(), self[side + ".R"], self[side + ".P"],
                (int(self[side + ".width"]), int(self[side + ".height"])))


# ---- images --------------------------------------------------------------------------------------
def _unfilter(raw, h, stride, bpp):
    out = np.empty((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    pos = 0
    for y in range(h):
        ft = raw[pos]; line = np.frombuffer(raw, np.uint8, stride, pos + 1).astype(np.int32); pos += stride + 1
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.copy()
            for c in range(bpp):                                   # per byte lane: prefix sums modulo 256
                cur[c::bpp] = np.cumsum(line[c::bpp]) & 255
        else:
            cur = np.zeros(stride, np.int32)
            ln = line.tolist(); pv = prev.tolist(); cu = [0] * stride
            for i in range(stride):
                a = cu[i - bpp] if i >= bpp else 0
                b = pv[i]
                c = pv[i - bpp] if i >= bpp else 0
                if ft == 3:
                    pr = (a + b) >> 1
                elif ft == 4:
                    p = a + b - c; pa = abs(p - a); pb = abs(p - b); pc = abs(p - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError("PNG filter type %d" % ft)
                cu[i] = (ln[i] + pr) & 255
            cur = np.asarray(cu, np.int32)
        out[y] = cur
        prev = cur
    return out


def imread(path):
    """8-bit PNG -> numpy like cv::imread(path, CV_LOAD_IMAGE_UNCHANGED): grey [H][W], colour [H][W][3] in B,G,R
    order (palette images expand to BGR).  16-bit, alpha and interlaced files are refused."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s: not a PNG file" % path)
    pos = 8; idat = []; plte = None; hdr = None
    while pos + 8 <= len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + n
    if hdr is None or not idat:
        raise ValueError("%s: truncated PNG" % path)
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace or ctype not in (0, 2, 3):
        raise ValueError("%s: only 8-bit non-interlaced grey / RGB / palette PNGs (depth %d, colour type %d)" % (path, depth, ctype))
    ch = {0: 1, 2: 3, 3: 1}[ctype]
    px = _unfilter(zlib.decompress(b"".join(idat)), h, w * ch, ch)
    if ctype == 0:
        return px
    if ctype == 3:
        if plte is None:
            raise ValueError("%s: palette image without PLTE" % path)
        return np.ascontiguousarray(plte[px][..., ::-1])
    return np.ascontiguousarray(px.reshape(h, w, 3)[..., ::-1])


def imwrite(path, img):
    """8-bit grey or B,G,R image -> PNG (filter 0; enough for synthetic sequences and round trips)."""
    a = np.ascontiguousarray(img, np.uint8)
    if a.ndim == 3:
        a = np.ascontiguousarray(a[..., ::-1]); ctype = 2
    else:
        ctype = 0
    h, w = a.shape[:2]
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def to_gray(img, rgb, cv3=False):
    """cvtColor(img, CV_RGB2GRAY) when Camera.RGB = 1, CV_BGR2GRAY otherwise (Tracking.cc:278-291), OpenCV's 8-bit fixed point:
    (R*9798 + G*19235 + B*3735 + 16384) >> 15 as OpenCV 4.x computes it (the semantics every other primitive here is frozen to), or with
    cv3=True OpenCV <= 3.x's (R*4899 + G*9617 + B*1868 + 8192) >> 14.  Grey images pass through.  The device form: ivf_frontend_run_color."""
    if img.ndim == 2:
        return img
    a = img.astype(np.int32)
    c0, c1, c2 = a[..., 0], a[..., 1], a[..., 2]
    r, b = (c0, c2) if rgb else (c2, c0)
    if cv3:
        return ((r * 4899 + c1 * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
    return ((r * 9798 + c1 * 19235 + b * 3735 + 16384) >> 15).astype(np.uint8)

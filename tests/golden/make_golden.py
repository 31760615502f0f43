"""Regenerates tests/golden/*.npz.  Run in the build container only: `python tests/golden/make_golden.py`.

What these fixtures are: the reference ships NO tests, vectors or sample images for this path and its
arithmetic lives in un-vendored OpenCV (absent here), so there is nothing of the reference's to copy.
The fixtures freeze the ORACLE's outputs (oracle/ivf_oracle.c, itself pinned per primitive by the KATs in
tests/test_oracle_primitives.py) on seeded inputs, so that any later drift of the oracle or of the HIP
path is caught against committed data.  Inputs of the miniature case are stored in full; the full-size
case stores CRCs of the seeded generator's images plus all outputs.
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from iv_slam_amd import synth  # noqa: E402

BF, B = 386.1448, 386.1448 / 718.856


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def case(L, R, cost, n, introspection):
    eL = O.Extractor(n, 1.2, 8, 20, 7, introspection); eR = O.Extractor(n, 1.2, 8, 20, 7, False)
    kL, dL = eL(L, cost); kR, dR = eR(R, cost)
    ur, dp = O.stereo_match(eL, eR, kL, dL, kR, dR, BF, B)
    pyr = np.array([crc(eL.pyramid(l)) for l in range(8)], np.uint32)
    dims = np.array([eL.pyramid(l).shape for l in range(8)], np.int32)
    out = dict(kpsL=kL, descL=dL, kpsR=kR, descR=dR, uright=ur, depth=dp, pyr_crc=pyr, pyr_dims=dims,
               level_counts=np.array(eL.level_counts(), np.int32))
    if introspection:
        out["qpyr_crc"] = np.array([crc(eL.quality_pyramid(l)) for l in range(8)], np.uint32)
    return out


def main():
    # miniature: inputs stored in full
    L, R = synth.make_pair(320, 200, seed=7, idx=0)
    cost = synth.make_cost_map(320, 200, seed=7, idx=0)
    np.savez_compressed(os.path.join(HERE, "mini_320x200.npz"), left=L, right=R, cost=cost,
                        **{"plain_" + k: v for k, v in case(L, R, None, 300, False).items()},
                        **{"intro_" + k: v for k, v in case(L, R, cost, 300, True).items()})
    # full size (BASELINE configs[0]/[1]/[2] shape): generator seeds + CRCs of inputs, all outputs
    L, R = synth.make_pair(1242, 375, seed=7, idx=1)
    cost = synth.make_cost_map(1242, 375, seed=7, idx=1)
    np.savez_compressed(os.path.join(HERE, "kitti_1242x375.npz"), seed=np.array([7, 1]),
                        in_crc=np.array([crc(L), crc(R), crc(cost)], np.uint32),
                        **{"plain_" + k: v for k, v in case(L, R, None, 1000, False).items()},
                        **{"intro_" + k: v for k, v in case(L, R, cost, 1000, True).items()},
                        **{"n2000_" + k: v for k, v in case(L, R, None, 2000, False).items()})
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()

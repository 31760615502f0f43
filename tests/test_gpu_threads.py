"""The reference starts two fresh std::threads per Frame for the left / right extractors (ORB/src/Frame.cc:116-124): everything the
C-ABI keeps per host thread must be handed back when a thread ends, or a live robot leaks page-locked memory frame after frame."""
import threading

import numpy as np
import pytest

import oracle_lib as O
from iv_slam_amd import synth

pytestmark = pytest.mark.gpu


def test_short_lived_threads_hand_their_scratch_back():
    """operator() + mvImagePyramid copies + ComputeStereoMatches from 600 short-lived threads (two alive at a time, like the
    reference): the per-thread scratch slots (pinned host + device buffers) are recycled, so their number follows the peak number
    of CONCURRENT threads, not the number of threads ever started; results stay bit-exact on every thread."""
    import iv_slam_amd as iv
    lib = iv.load()
    assert lib.ivf_device_count() >= 1, "no HIP device: libivfront has no CPU fallback"
    L, R = synth.make_pair(640, 240, seed=41, idx=0)
    eL = iv.ORBextractor(500, 1.2, 8, 20, 7); eR = iv.ORBextractor(500, 1.2, 8, 20, 7)
    oL = O.Extractor(500, 1.2, 8, 20, 7)
    okL, odL = oL(L)
    exp_top = oL.pyramid(7)
    errors = []

    def side(ext, img, want_k, want_d, out):
        try:
            k, d = ext(img)
            pyr = ext.mvImagePyramid
            assert len(pyr) == 8 and pyr[0].shape == img.shape and np.array_equal(pyr[0], img)
            if want_k is not None:
                assert k.tobytes() == want_k.tobytes() and np.array_equal(d, want_d) and np.array_equal(pyr[7], exp_top)
            out.append((k, d))
        except Exception as e:              # noqa: BLE001 -- collected and re-raised on the main thread
            errors.append(e)

    def frame():
        a, b = [], []
        tl = threading.Thread(target=side, args=(eL, L, okL, odL, a)); tr = threading.Thread(target=side, args=(eR, R, None, None, b))
        tl.start(); tr.start(); tl.join(); tr.join()
        return a, b

    frame()
    warm = lib.ivf_debug_scratch_slots()
    assert 1 <= warm <= 3
    for _ in range(300):
        a, b = frame()
        assert not errors, errors[:1]
        # ComputeStereoMatches on yet another short-lived thread (its device scratch is thread-leased too)
        res = []
        t = threading.Thread(target=lambda: res.append(iv.ComputeStereoMatches(eL, eR, a[0][0], a[0][1], b[0][0], b[0][1], 386.1448, 386.1448 / 718.856)))
        t.start(); t.join()
        assert len(res) == 1 and (res[0][0] >= 0).sum() > 50
    assert lib.ivf_debug_scratch_slots() <= warm + 1, (warm, lib.ivf_debug_scratch_slots())

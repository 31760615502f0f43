"""GPU parity of the introspection FCN: HIP path (C-ABI) vs golden vectors of the reference's own Python model
(1e-3 bar from BASELINE.json:north_star) and vs the numpy oracle."""
import numpy as np
import pytest

import fcn_common as FC
from iv_slam_amd import fcn_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def iv():
    import iv_slam_amd
    assert iv_slam_amd.load().ivf_device_count() >= 1
    return iv_slam_amd


@pytest.mark.parametrize("tag", ["kitti", "jackal", "jackal_full", "kitti_smallw", "jackal_smallw", "kitti_bigw"])
def test_fcn_matches_reference_goldens(iv, tag):
    g, W, bgr, out_size = FC.load_case(tag)
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out_size)
    u8, cost = fcn(bgr, want_f32=True)
    err = FC.check_against_golden(g, cost, u8, tol=1e-3)
    assert err < 3e-4, "f32-MFMA path should sit well inside the 1e-3 bar (got %.3g)" % err
    assert np.array_equal(u8, (cost * np.float32(255.0)).astype(np.uint8))     # truncation, not rounding
    # idempotent
    u8b = fcn(bgr)
    assert np.array_equal(u8, u8b)


def test_fcn_matches_numpy_oracle_full_map(iv):
    import fcn_oracle
    g, W, bgr, out_size = FC.load_case("kitti")
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out_size)
    u8, cost = fcn(bgr, want_f32=True)
    oc, ou8 = fcn_oracle.forward(W, bgr, out_size)
    assert np.abs(cost - oc).max() < 3e-4
    assert (np.abs(u8.astype(int) - ou8.astype(int)) <= 1).all() and (u8 != ou8).mean() < 0.01


def test_fcn_batch_device_path_and_extractor_coupling(iv):
    """configs[2]: FCN cost map (device) gating keypoints in the extractor, all resident in HBM."""
    import torch
    import oracle_lib as O
    g, W, bgr, out_size = FC.load_case("kitti")
    dev = torch.device("cuda:0")
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out_size, max_batch=2)
    batch = torch.from_numpy(np.stack([bgr, bgr[:, ::-1].copy()])).to(dev)
    cost_u8 = torch.empty((2,) + tuple(out_size), dtype=torch.uint8, device=dev)
    fcn.forward_device(batch, cost_u8=cost_u8)
    torch.cuda.synchronize()
    host = cost_u8.cpu().numpy()
    # a batch of 2 and a single image cut the hidden groups of blocks 8-17 into different numbers of ranges (small-batch schedule):
    # same map up to the f32 summation order of the projections, i.e. at most one u8 step at a truncation boundary
    single = fcn(bgr)
    assert np.abs(host[0].astype(int) - single.astype(int)).max() <= 1 and (host[0] != single).mean() < 0.01
    assert not np.array_equal(host[0], host[1])
    # feed the device cost maps straight into the batched front end
    grey = np.stack([bgr[..., 1], bgr[:, ::-1, 1]]).copy()
    left = torch.from_numpy(grey).to(dev)
    fe = iv.StereoFrontend(out_size[1], out_size[0], 2, nfeatures=1000, enableIntrospection=True)
    fe.run(left, left.clone(), cost_u8)
    fe.sync()
    for p in range(2):
        ok, od = O.Extractor(1000, 1.2, 8, 20, 7, introspection=True)(grey[p], host[p])
        r = fe.fetch(p, 0)
        assert r["kps"].tobytes() == ok.tobytes() and np.array_equal(r["desc"], od)


def test_fcn_rejects_bad_blob(iv):
    W = fcn_weights.make_seeded_weights(0)
    blob = fcn_weights.pack_blob(W)
    with pytest.raises(iv.IvfError):
        iv.IntrospectionFCN(blob[:-5], (375, 1242))
    with pytest.raises(iv.IvfError):
        iv.IntrospectionFCN(np.concatenate([blob, blob[:3]]), (375, 1242))


_VARIANT_SCRIPT = r"""
import sys, os
sys.path.insert(0, os.environ["IVF_REPO"]); sys.path.insert(0, os.path.join(os.environ["IVF_REPO"], "tests"))
import numpy as np, torch
import fcn_common as FC, iv_slam_amd as iv
from iv_slam_amd import fcn_weights
g, W, bgr, out = FC.load_case("kitti")
NB = 20      # 640 workgroups in the 64x64 stages: every CU holds co-resident workgroups of every kernel
f = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out, max_batch=NB)
u8, cost = f(bgr, want_f32=True)
err = FC.check_against_golden(g, cost, u8, tol=1e-3)
dev = torch.device("cuda:0")
flipped = bgr[:, ::-1].copy()
batch = torch.from_numpy(np.stack([bgr if i % 2 == 0 else flipped for i in range(NB)])).to(dev)
outs = []
for rep in range(2):
    c = torch.empty((NB,) + tuple(out), dtype=torch.uint8, device=dev)
    f.forward_device(batch, cost_u8=c); torch.cuda.synchronize()
    outs.append(c.cpu().numpy())
assert np.array_equal(outs[0], outs[1]), "batched forward is not deterministic"
u8b = f(bgr)
assert np.array_equal(u8, u8b), "single-image forward is not deterministic"
strict = os.environ.get("IVF_FCN_SPLIT") == "0"      # without the small-batch schedule a single image runs the batched kernels: bit-identical
for i in range(0, NB, 2):
    assert np.array_equal(outs[0][i], outs[0][0]), "batch slot %d differs from slot 0 (same input)" % i
    assert np.array_equal(outs[0][i + 1], outs[0][1]), "batch slot %d differs from slot 1 (same input)" % (i + 1)
if strict:
    assert np.array_equal(outs[0][0], u8), "batch slot 0 differs from the single-image result"
else:
    d = np.abs(outs[0][0].astype(int) - u8.astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 0.01, "batch slot 0 vs single image: %d steps, %.4f of the map" % (d.max(), (d != 0).mean())
print("OK %.3g" % err)
"""


@pytest.mark.parametrize("env", [
    {},                                                                        # the measured defaults
    # layer-by-layer fallbacks only: k_fcn_gemm for every 1x1 / 3x3, k_fcn_dw for every depthwise layer
    {"IVF_FCN_NOFUSE": "1", "IVF_FCN_EXPAND": "0", "IVF_FCN_OLD3X3": "1"},
    # 64-pixel expansion tiles
    {"IVF_FCN_EXPAND": "2"},
    # fused depthwise+projection also on the 256-wide map of block 1 (off by default: slower there)
    {"IVF_FCN_WIDE256": "1"},
    # stride-2 blocks layer by layer (the default fuses their depthwise + projection)
    {"IVF_FCN_NOSTRIDE2": "1"},
    {"IVF_FCN_NOSTEM": "1"},
    # whole-block kernels also on blocks 5-11 (k_fcn_irb64; opt-in: slower than expand + dwpw there)
    {"IVF_FCN_IRBMASK": "0x3ff"},
    # ... and on none (blocks 2-4 through k_fcn_gemm + k_fcn_dwpw)
    {"IVF_FCN_IRBMASK": "0"},
    # the 8-wave depthwise + projection kernel on every shape it covers / on none (block 17 keeps it: IVF_FCN_NODWPW10 is separate)
    {"IVF_FCN_DWPW8": "0x2c"},
    {"IVF_FCN_DWPW8": "0", "IVF_FCN_NODWPW10": "1"},
    # k_fcn_dwpw with workgroups of four image rows (eight waves) on every 64 x 64 shape / on none (default: <= 2 output tiles)
    {"IVF_FCN_NW": "8", "IVF_FCN_DWPW8": "0", "IVF_FCN_NODWPW10": "1"},
    {"IVF_FCN_NW": "4"},
    # conv_last as its own kernel after the decoder 3x3 (default: folded into its epilogue)
    {"IVF_FCN_NOFUSELAST": "1"},
    # blocks 8-14 / 15-17 / 5-7 as expand + depthwise-projection launches instead of k_fcn_irbd2 / k_fcn_irbd4 (the defaults)
    {"IVF_FCN_FUSED2": "0"},
    {"IVF_FCN_FUSED4": "0"},
    {"IVF_FCN_FUSED1": "0"},
    # block 17 as two workgroups per 256-pixel tile (k_fcn_irbd4<false>, r03 / r04) instead of the one-pass half-tile kernel k_fcn_irbd4h (r05 default)
    {"IVF_FCN_HALF4": "0"},
    # r05 experiments kept as paths: the 512 / 256 / 128 stage in chunks of 8 images back to back; its three inner tensors row-interleaved ([y][channel][x])
    {"IVF_FCN_HEADCHUNK": "8"},
    {"IVF_FCN_HEAD_IL": "1"},
    {"IVF_FCN_HEAD_IL": "1", "IVF_FCN_HEADCHUNK": "8"},
    # no small-batch schedule: a single image runs the batched whole-block kernels (16 workgroups per launch) and equals its batch slot bit for bit
    {"IVF_FCN_SPLIT": "0"},
    # r06: the expansion's two correction products of blocks 15 / 16 (and block 17's two-workgroup form) on the block-scaled bf6 x fp6 matrix instruction
    # (k_fcn_irbd4<.., FP6>; not the default: DESIGN.md section 7.r06) -- with and without the small-batch schedule, so the batched instance runs too
    {"IVF_FCN_FP6": "1"},
    {"IVF_FCN_FP6": "1", "IVF_FCN_SPLIT": "0", "IVF_FCN_HALF4": "0"},
    # r06: the decoder's 3x3 as three f16 products (k_fcn_conv3x3_all, r02-r05) instead of hi * hi + the fp6 correction product (k_fcn_conv3x3_f6, the default)
    {"IVF_FCN_DEC6": "0"},
    # ... and the first fp6 form (four-pixel lanes: every (unit, row) loads its own input row) instead of the row-sharing one
    {"IVF_FCN_DEC6": "1"},
], ids=["default", "layerwise", "expand-pxt2", "dwpw-256", "no-stride2-fusion", "no-stem-fusion", "irb-blocks-2-11",
        "irb-none", "dwpw8-all", "dwpw8-none", "dwpw-4rows-all", "dwpw-4rows-none", "conv-last-unfused", "blocks-8-14-unfused", "blocks-15-17-unfused", "blocks-5-7-unfused", "block-17-two-workgroups", "head-chunked", "head-row-interleaved", "head-chunked-and-interleaved", "no-small-batch-split", "fp6-expansion", "fp6-expansion-batched-kernels", "decoder-three-f16-products", "decoder-fp6-without-row-sharing"])
def test_fcn_kernel_variants_match_goldens_and_are_deterministic(env):
    """The FCN picks between several kernels per layer (measured defaults, env overrides for tuning).  Every variant must
    meet the same bar, batch results must not depend on the batch slot, and repeated runs must be bit-identical (this is
    the test that caught two co-resident workgroups of the stride-2 block kernel corrupting each other)."""
    import os, subprocess, sys
    from iv_slam_amd import _lib
    e = dict(os.environ); e.update(env); e["IVF_REPO"] = FC.ROOT
    if env:
        # the selectors exist only in the experiment build (make EXPERIMENT=1; __graft_entry__.build() makes it): the product
        # library ignores them (tests/test_abi_cpu.py), so a variant that silently ran the defaults would prove nothing
        assert os.path.exists(_lib.EXPERIMENT_LIB_PATH), "libivfront_exp.so missing: make -C iv_slam_amd/csrc EXPERIMENT=1"
        e["IVFRONT_LIB"] = _lib.EXPERIMENT_LIB_PATH
    r = subprocess.run([sys.executable, "-c", _VARIANT_SCRIPT], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert float(r.stdout.split("OK")[1]) < 3e-4


_DEC_SCRIPT = r"""
import sys, os
sys.path.insert(0, os.environ["IVF_REPO"]); sys.path.insert(0, os.path.join(os.environ["IVF_REPO"], "tests"))
import numpy as np, torch
import fcn_common as FC, iv_slam_amd as iv
from iv_slam_amd import fcn_weights
g, W, bgr, out = FC.load_case("jackal")
dev = torch.device("cuda:0")
outs = []
for nb in (16, 3, 1):          # the batched kernel; ranges of whole K-step pairs (5 ways); ranges that are not (15 ways: the four-pixel form runs)
    f = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out, max_batch=nb)
    batch = torch.from_numpy(np.stack([bgr] * nb)).to(dev)
    cf = torch.empty((nb,) + tuple(out), dtype=torch.float32, device=dev)
    f.forward_device(batch, cost_f32=cf); f.status()
    outs.append(cf[nb - 1].cpu().numpy())
np.save(os.environ["IVF_OUT"], np.stack(outs))
print("OK")
"""


def test_decoder_row_sharing_form_is_bit_identical_to_the_four_pixel_form(tmp_path):
    """r06: k_fcn_conv3x3_f6r (two-pixel lanes, the two output rows of a wave share their input rows: four row loads / conversions per K-step pair instead of six) issues,
    per accumulator, the products of k_fcn_conv3x3_f6 in the same order with the same operands -- the cost maps must be EQUAL bit for bit at every batch size
    (batched kernel, ranges of whole pairs, ranges that fall back)."""
    import os, subprocess, sys
    from iv_slam_amd import _lib
    assert os.path.exists(_lib.EXPERIMENT_LIB_PATH), "libivfront_exp.so missing: make -C iv_slam_amd/csrc EXPERIMENT=1"
    res = {}
    for v in ("1", "2"):
        e = dict(os.environ); e.update({"IVF_FCN_DEC6": v, "IVF_REPO": FC.ROOT, "IVFRONT_LIB": _lib.EXPERIMENT_LIB_PATH, "IVF_OUT": str(tmp_path / ("dec%s.npy" % v))})
        r = subprocess.run([sys.executable, "-c", _DEC_SCRIPT], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        res[v] = np.load(e["IVF_OUT"])
    assert np.array_equal(res["1"], res["2"]), "max |d| %.3g" % float(np.abs(res["1"] - res["2"]).max())
    g, W, bgr, out = FC.load_case("jackal")
    sub = int(g["sub"][0]) if "sub" in g.files else 6
    assert float(np.abs(res["2"][0][::sub, ::sub] - g["cost_sub"]).max()) < 3e-4


def test_fcn_small_batch_schedule_agrees_with_the_batched_one(iv):
    """Batch 1 (the per-call drop-in path: hidden groups of blocks 8-17 cut into up to 15 ranges so that the launch fills the chip),
    batches 2 / 4 / 8 (fewer ranges) and batch 128 (none): the f32 cost maps agree far inside the 1e-3 bar, each is within 3e-4 of the
    reference golden, and each schedule is deterministic."""
    import torch
    g, W, bgr, out_size = FC.load_case("kitti")
    dev = torch.device("cuda:0")
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out_size, max_batch=128)
    u8_1, c1 = fcn(bgr, want_f32=True)
    assert FC.check_against_golden(g, c1, u8_1, tol=1e-3) < 3e-4
    u8_1b, c1b = fcn(bgr, want_f32=True)
    assert np.array_equal(c1, c1b)
    flipped = bgr[:, ::-1].copy()
    ref = None
    # 48 / 33: k_fcn_irbd2 as a persistent grid (r05: one workgroup per CU walks 3 / 2-or-3 tiles -- the ragged walk); 16: the last batch on the one-tile grid
    for nb in (128, 48, 33, 16, 8, 4, 2):
        batch = torch.from_numpy(np.stack([bgr if i % 2 == 0 else flipped for i in range(nb)])).to(dev)
        cf = torch.empty((nb,) + tuple(out_size), dtype=torch.float32, device=dev); cu = torch.empty((nb,) + tuple(out_size), dtype=torch.uint8, device=dev)
        fcn.forward_device(batch, cost_u8=cu, cost_f32=cf)
        torch.cuda.synchronize()
        c = cf.cpu().numpy()
        assert np.array_equal(c[0], c[nb - 2 - nb % 2]) and np.array_equal(c[1], c[nb - 1 - nb % 2])           # slot-independent
        if nb >= 16:
            for i in range(nb): assert np.array_equal(c[i], c[i % 2]), "batch %d: slot %d differs from slot %d (same input)" % (nb, i, i % 2)
            if ref is not None: assert np.array_equal(c[0], ref), "batch %d differs from batch 128 (same kernels, no split)" % nb
        if ref is None:
            ref = c[0]
            assert FC.check_against_golden(g, ref, cu[0].cpu().numpy(), tol=1e-3) < 3e-4
        # r06: 2e-5 -> 5e-5.  The decoder's correction product quantises block 17's output to 6-bit operands with a scale per (pixel, 16 channels) taken from
        # the data: where two schedules' activations differ in the last bit (summation order of the split ranges), a code can fall the other way -- a step of
        # 2^-4 of a term that is 2^-11 of the sum.  Measured 3.3e-5 (batch 8 vs 128; 1.1e-5 with three f16 products); each schedule stays within 3e-4 of the golden
        d = float(np.abs(c[0] - ref).max())
        assert d < 5e-5, "batch %d vs batch 128: %.3g" % (nb, d)
    d1 = float(np.abs(c1 - ref).max())
    assert d1 < 5e-5, "batch 1 vs batch 128: %.3g" % d1
    du = np.abs(u8_1.astype(int) - (ref * np.float32(255.0)).astype(np.uint8).astype(int))
    assert du.max() <= 1 and (du != 0).mean() < 0.01


def test_fcn_strided_input(iv):
    """the batched device path on padded rows / images (a view of a larger tensor, odd byte offsets): same result as contiguous"""
    import torch
    g, W, bgr, out_size = FC.load_case("kitti")
    dev = torch.device("cuda:0")
    h, w = bgr.shape[:2]
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), (h, w), out_size, max_batch=2)
    batch = torch.from_numpy(np.stack([bgr, bgr[::-1].copy()])).to(dev)
    a = torch.empty((2,) + tuple(out_size), dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    fcn.forward_device(batch, cost_u8=a)
    big = torch.zeros((2, h + 3, w + 11, 3), dtype=torch.uint8, device=dev)
    view = big[:, 1:1 + h, 5:5 + w]
    view.copy_(batch)
    assert not view.is_contiguous()
    fcn.forward_device(view, cost_u8=b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_f16_range_guard_flags_an_overflowing_activation_and_stays_quiet_below_the_edge(iv):
    """r05 (r04 verdict, weak #8): the split-f16 operands clamp at |x| = 65504.  The un-clamped activations are the linear-bottleneck
    outputs; a kernel that stores one >= 65504 raises a device-side flag and ivf_fcn_forward returns IVF_E_STATE instead of a plausible
    cost map (ivf_fcn_status after ivf_fcn_forward_device).  BatchNorm affine parameters of ONE projection scaled to put its output
      * above the edge (x 3e5): the error, from the per-call path, from the batched path, for a block of every kernel family;
      * just below it (max |x| in [2e4, 6.5e4)): no error, and the cost map still within 1e-3 of the numpy oracle -- the 22-bit products
        hold right up to the edge."""
    import torch
    import fcn_oracle
    from iv_slam_amd._lib import IvfError, IVF_E_STATE, IVF_E_INVALID
    g, W, bgr, out_size = FC.load_case("kitti")
    dev = torch.device("cuda:0")

    def scaled(block, c):
        # features.<block>.conv.7 = the projection's BatchNorm (t = 6 blocks): y = gamma * xhat + beta -> c * y
        V = dict(W)
        p = "encoder.features.%d.conv.7" % block
        V[p + ".weight"] = (W[p + ".weight"] * np.float32(c)).astype(np.float32)
        V[p + ".bias"] = (W[p + ".bias"] * np.float32(c)).astype(np.float32)
        return V

    batch = torch.from_numpy(np.stack([bgr, bgr[:, ::-1].copy()] * 10)).to(dev)           # 20 images: the batched kernels, no split
    for block in (2, 3, 6, 9, 14, 16, 17):          # k_fcn_irb (stride 2 / residual), k_fcn_irbd2 <DIL 1> / <DIL 2>, k_fcn_irbd4 <res> / <320>
        V = scaled(block, 3e5)
        f = iv.IntrospectionFCN(fcn_weights.pack_blob(V), bgr.shape[:2], out_size, max_batch=20)
        with pytest.raises(IvfError) as e:
            f(bgr)                                   # per-call path (batch 1: the small-batch split schedule + reduce kernels)
        assert e.value.code == IVF_E_STATE and "f16 range" in str(e.value), str(e.value)
        cu = torch.empty((20,) + tuple(out_size), dtype=torch.uint8, device=dev)
        f.forward_device(batch, cost_u8=cu)
        with pytest.raises(IvfError) as e:
            f.status()
        assert e.value.code == IVF_E_STATE
        f.status()                                   # cleared by the report
    # an ordinary network never raises it
    f = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out_size, max_batch=20)
    cu = torch.empty((20,) + tuple(out_size), dtype=torch.uint8, device=dev)
    f.forward_device(batch, cost_u8=cu); f.status()
    # just below the edge, on RE-PARAMETERISED networks: an un-clamped tensor that feeds exactly one BatchNorm-ed convolution is scaled
    # by c (its projection's BN affine x c) and the consumer's BN statistics compensate (running_mean x c, running_var x c^2): the same
    # function up to eps, well conditioned (the consumer's BN scale brings the magnitudes back -- no cancellation), but the tensor the
    # split-f16 operands are cut from now peaks at ~4e4.  Block 1 -> block 2's expansion (k_fcn_stem -> k_fcn_irb) and block 17 -> the
    # decoder 3x3 (k_fcn_irbd4 -> k_fcn_conv3x3_all) are the two such tensors of this architecture (every other one also feeds a residual).
    # (Scaling a tensor WITHOUT compensation is not a usable test: the consumer then cancels terms of 1e4 into [0, 6] and numpy's f32
    # sums themselves are only good to 6e-3 there -- measured: product and oracle 6.2e-3 apart at 4e4.)
    base_cost, _u, base_taps = fcn_oracle.forward(W, bgr, out_size, return_taps=True)
    for name, prod_bn, prod_block, cons_bn in (("block 1 -> block 2", "encoder.features.1.conv.4", 1, "encoder.features.2.conv.1"),
                                               ("block 17 -> decoder", "encoder.features.17.conv.7", 17, "decoder.cbr.1")):
        c = np.float32(4.0e4 / base_taps["block_absmax"][prod_block])
        V = dict(W)
        V[prod_bn + ".weight"] = (W[prod_bn + ".weight"] * c).astype(np.float32); V[prod_bn + ".bias"] = (W[prod_bn + ".bias"] * c).astype(np.float32)
        V[cons_bn + ".running_mean"] = (W[cons_bn + ".running_mean"] * c).astype(np.float32)
        V[cons_bn + ".running_var"] = (W[cons_bn + ".running_var"] * c * c).astype(np.float32)
        oc, ou8, taps = fcn_oracle.forward(V, bgr, out_size, return_taps=True)
        assert 3.0e4 < taps["block_absmax"][prod_block] < 6.5e4, (name, taps["block_absmax"][prod_block])
        assert float(np.abs(oc - base_cost).max()) < 2e-3, name + ": the re-parameterised network is not the same function"
        f = iv.IntrospectionFCN(fcn_weights.pack_blob(V), bgr.shape[:2], out_size, max_batch=20)
        u8, cost = f(bgr, want_f32=True)             # no error raised
        err = float(np.abs(cost - oc).max())
        assert err < 1e-3, "%s at %.3g: cost map %.3g from the oracle" % (name, taps["block_absmax"][prod_block], err)
        cf = torch.empty((20,) + tuple(out_size), dtype=torch.float32, device=dev)
        f.forward_device(batch, cost_f32=cf); f.status()         # the batched kernels: no flag either
        assert float(np.abs(cf[0].cpu().numpy() - oc).max()) < 1e-3, name + " (batched)"
    # non-finite weights never reach the device
    V = dict(W); V["encoder.features.5.conv.0.weight"] = W["encoder.features.5.conv.0.weight"].copy(); V["encoder.features.5.conv.0.weight"].flat[7] = np.inf
    with pytest.raises(IvfError) as e:
        iv.IntrospectionFCN(fcn_weights.pack_blob(V), bgr.shape[:2], out_size)
    assert e.value.code == IVF_E_INVALID

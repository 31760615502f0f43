"""Build profiles/*_pmc_hbm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: python tools/pmc_to_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <images_per_launch> <out.json> "<command>"

Per kernel symbol (and, for the FCN's fused depthwise+projection kernel, per launch shape) the per-launch average of
raw counter x 1024 bytes.  `fetch_correction` = 2.0 where the kernel's global reads are 16 B per lane: on gfx950 FETCH_SIZE
reports half the bytes of such streams (MI355X_MICROARCH.md, HBM section); other widths are uncalibrated and left at 1.0.
"""
import csv, json, sys, collections

# r02: tools/probe/fetch_calib.hip read 1 GiB with 4-, 8- and 16-byte loads per lane: FETCH_SIZE reported 512 MiB for ALL THREE
# (profiles/r02_fetch_calibration.json), so the x2 applies to every streaming read of this code base, not only to 16 B/lane ones.
WIDE = ("k_",)


def per_launch(path, counter):
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_fcn_dwpw<5, 4," in k:                      # the probed kernel: split its two launch shapes, stable key for bench.py
            k = "ivffcn::k_fcn_dwpw<5, 4>"
            wgs = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
            k += " 960->160" if wgs == 32 * (int(sys.argv[3]) // 2) else " 960->320"     # 32 row pairs per image (x2 channel halves)
        if k.endswith("k_fcn_irbd4<true>") or k.endswith("k_fcn_irbd4<true, false>") or k.endswith("k_fcn_irbd4<true, false, false>"): k = "ivffcn::k_fcn_irbd4<true> 160->960->160"        # the probe's name for blocks 15 / 16
        if k.endswith("k_fcn_irbd4<false>") or k.endswith("k_fcn_irbd4<false, false>") or k.endswith("k_fcn_irbd4<false, false, false>"): k = "ivffcn::k_fcn_irbd4<false> 160->960->320"     # block 17 (r03 / r04 form)
        if k.endswith("k_fcn_irbd4h") or k.endswith("k_fcn_irbd4h<false>"): k = "ivffcn::k_fcn_irbd4h 160->960->320"                                                              # block 17 (r05)
        acc[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    return {k: (v * 1024 / len(disp[k]), len(disp[k])) for k, v in acc.items()}


def main():
    fetch = per_launch(sys.argv[1], "FETCH_SIZE"); write = per_launch(sys.argv[2], "WRITE_SIZE")
    n_img = int(sys.argv[3])
    fcn_img = int(sys.argv[6]) if len(sys.argv) > 6 else n_img // 2
    import subprocess, os
    try:
        commit = subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        # the GPU box has no .git: the snapshot carries the commit it was cut from in .ivf_commit (written by tools/stamp_commit.sh before gpurun)
        commit = os.environ.get("IVF_COMMIT") or "unknown"
        try:
            commit = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".ivf_commit")).read().strip() or commit
        except OSError:
            pass
    # the build the counters were collected on: ivf_build_id() hashes the sources + flags, and the same hash recomputed from the tree --
    # always available, also on the GPU box (which has no .git) -- so the counter file is tied to a build like every bench line is
    build = {}
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from iv_slam_amd import _lib
        build = {"sources": _lib.source_build_id("")}
        import ctypes
        so = ctypes.CDLL(_lib.LIB_PATH)
        so.ivf_build_id.restype = ctypes.c_char_p; so.ivf_build_flags.restype = ctypes.c_char_p
        build["libivfront"] = so.ivf_build_id().decode(); build["flags"] = so.ivf_build_flags().decode()
    except Exception as e:
        build["error"] = str(e)
    out = {"command": sys.argv[5], "images_per_launch": n_img, "commit": commit, "build": build,
           "note": "raw counter x 1024 bytes per launch; fetch_correction = 2.0: gfx950 FETCH_SIZE reports half of the bytes of a streaming read "
                   "(MI355X_MICROARCH.md HBM section; calibrated here for 4 / 8 / 16 B per lane by tools/probe/fetch_calib.hip: 512 MiB "
                   "reported for a 1 GiB read at every width); copies / fills of the runtime are left uncorrected",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0)); w, nw = write.get(k, (0.0, 0))
        corr = 2.0 if any(t in k for t in WIDE) else 1.0
        e = {"fetch_bytes_per_launch": int(f * corr), "fetch_raw_bytes_per_launch": int(f), "fetch_correction": corr,
             "write_bytes_per_launch": int(w), "launches": max(nf, nw)}
        if "k_fcn" in k: e["images_per_launch"] = fcn_img             # the FCN sees the left images only, in sub-batches (r06: 64 per launch sequence)
        out["kernels"][k] = e
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print("wrote", sys.argv[4], len(out["kernels"]), "kernels")


if __name__ == "__main__":
    main()

"""bench.py's contract on a real GPU: one JSON line with the roofline and (optionally) cpu_baseline objects, and the
multi-GPU exchange step (RCCL all-gather of descriptor records on a side stream) exercised with a single rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", *extra],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    j = _run("--pairs", "16", "--stream", "32")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["value"] > 0 and j["unit"] == "pairs/s" and j["vs_baseline"] is None
    rf = j["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 * max(rf["frac"], 1e-9) + 1e-6
    assert "workload" in j["config"]
    # r05: the line names the FCN kernel with the largest TOTAL time per forward and prints the other whole-block probe beside it, both priced
    # against the f16 matrix peak as issued and as algorithmic flops; the tracker step covers every frame pair of the launch sequence
    assert rf["launches_per_forward"] in (1, 2) and rf["total_ms_per_forward"] > 0 and 0 < rf["frac_algorithmic"] < rf["frac_issued"] < 1
    # r06: `frac` IS SURVEY 8(d)'s algorithmic fraction (2 x MAC counted once); the three split-f16 products are `frac_issued` beside it
    assert rf["frac"] == rf["frac_algorithmic"] and abs(rf["frac_issued"] - 3 * rf["frac"]) < 1e-3
    ob = rf["other_block"]
    assert ob["kernel"] != rf["kernel"] and 0 < ob["frac_algorithmic"] < ob["frac_issued"] < 1 and ob["total_ms_per_forward"] <= rf["total_ms_per_forward"]
    assert {rf["kernel"].split(" ")[0], ob["kernel"].split(" ")[0]} == {"ivffcn::k_fcn_irbd4<true>", "ivffcn::k_fcn_irbd4h"}
    assert j["track"]["frame_pairs_per_launch_sequence"] == 16


def test_exchange_step_single_rank():
    j = _run("--pairs", "16", "--stream", "32", "--force-gather", "--no-introspect")
    assert j["value"] > 0
    # the exchange has a consumer: the batched tracker ran on the gathered records inside the timed region and equals the oracle
    assert j["exchange"]["consumed"] is True and j["track"]["parity_ok"] is True and j["track"]["in_timed_region"] is True
    # r05: 16 pairs per launch sequence of 16 frames -- the pair at the batch boundary comes through the carry record
    assert j["track"]["frame_pairs_per_launch_sequence"] == 16 and j["track"]["mean_matches"] > 50
    assert j["parity_spot_check"]["ok"] is True
    # the r04 behaviour stays selectable for A/B runs
    j = _run("--pairs", "16", "--stream", "32", "--no-introspect", "--no-carry")
    assert j["track"]["frame_pairs_per_launch_sequence"] == 15 and j["track"]["parity_ok"] is True


def test_tracker_step_is_timed_at_every_rank_count():
    """the 1-GPU line times the same work per frame as the N-GPU lines: pack + tracker step inside the timed region by default."""
    j = _run("--pairs", "16", "--stream", "32")
    assert "exchange" not in j and j["track_in_timed_region"] is True
    assert j["track"]["parity_ok"] is True and j["track"]["in_timed_region"] is True and j["track"]["us_per_frame_pair"] > 0
    assert j["build"]["libivfront"] == j["build"]["sources"] and len(j["build"]["libivfront"]) == 16
    j = _run("--pairs", "16", "--stream", "32", "--no-track", "--no-introspect")
    assert j["track_in_timed_region"] is False and j["track"]["in_timed_region"] is False and j["track"]["parity_ok"] is True


@pytest.mark.parametrize("cfg,pairs,w,h,n", [(3, 8, 1242, 375, 2000), (4, 4, 1920, 1200, 4000)])
def test_other_baseline_configs_have_bench_lines(cfg, pairs, w, h, n):
    """bench.py --config 3 / --config 4 (BASELINE.json configs[3] / configs[4]): one contract line each, oracle spot check of the last
    timed launch sequence, tracker parity, both rooflines."""
    j = _run("--config", str(cfg), "--pairs", str(pairs), "--stream", str(2 * pairs), "--batches-per-step", "2")
    c = j["config"]
    assert c["baseline_config_index"] == cfg and (c["width"], c["height"], c["nfeatures"]) == (w, h, n) and ("configs[%d]" % cfg) in c["workload"]
    assert j["value"] > 0 and j["parity_spot_check"]["ok"] is True and j["track"]["parity_ok"] is True
    assert j["roofline"]["bound"] == "mfma" and j["roofline_fast_nms"]["bound"] == "hbm" and j["roofline_fast_nms"]["frac"] > 0
    if cfg == 4:
        assert c["fast_thresholds"] == [12, 7]


@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_ranks_on_one_gpu_through_gloo(ranks):
    """The N > 1 path executes on a 1-GPU box: `bench.py --gpus N` spawns N ranks that share device 0 (IVF_BENCH_BACKEND=gloo,
    a test aid: blocks cross the host), all-gathers the record blocks, runs the tracker on frames extracted by ANOTHER rank (rank 0
    wraps to the last rank's previous slot), verifies the exchange and the MAX all-reduce, and prints ONE line with n_gpus == N."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", IVF_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--pairs", "16", "--stream", "32"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == ranks and j["value"] > 0 and j["scaling"] == "weak" and j["track_in_timed_region"] is True
    ex = j["exchange"]
    assert ex["world"] == ranks and ex["world_size_seen_by_backend"] == ranks and ex["backend"] == "gloo"
    assert ex["records_checked"] == 16 * ranks and ex["consumed"] is True
    assert ex["enqueued_on_stream"] == ex["batch_stream_of_that_run"]          # the collective sits on the batch's own stream, behind the pack
    # rank 0 tracked its frames j >= 1 against the LAST rank's frames j - 1 and its frame 0 against the last rank's LAST frame of the
    # previous batch (the carry record): 16 pairs; consecutive GLOBAL frames re-match
    assert j["track"]["parity_ok"] is True and j["track"]["frame_pairs_per_launch_sequence"] == 16 and j["track"]["mean_matches"] > 50
    assert j["parity_spot_check"]["ok"] is True


def _device_count():
    import iv_slam_amd
    return iv_slam_amd.load().ivf_device_count()


def test_bench_gpus_flag_spawns_ranks_nccl():
    """`bench.py --gpus 2` with WORLD_SIZE unset must start two ranks itself (RCCL all-gather between them) and print
    ONE line with n_gpus == 2.  Runs only where two devices are visible (the driver's 8-GPU node); skipped on 1-GPU boxes."""
    if _device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--pairs", "16", "--stream", "32"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["scaling"] == "weak"
    ex = j["exchange"]
    assert ex["world"] == 2 and ex["world_size_seen_by_backend"] == 2 and ex["records_checked"] > 0 and ex["consumed"] is True
    assert ex["enqueued_on_stream"] == ex["batch_stream_of_that_run"]


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)

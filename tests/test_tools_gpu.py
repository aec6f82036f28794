"""The tools/ entry points end to end on synthetic data (GPU): eval CLI produces metrics, training CLI lowers the loss
and writes a reference-format checkpoint that the eval CLI loads."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "vidsgg-big_amd", "tools")
CFG = os.path.join(ROOT, "vidsgg-big_amd", "experiments", "exp2", "config_.py")


def run(args):
    r = subprocess.run([sys.executable] + args, cwd=TOOLS, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_train_then_eval_cli_roundtrip():
    out_dir = tempfile.mkdtemp()
    log = run([os.path.join(TOOLS, "train_vidvrd.py"), "--cfg_path", CFG, "--output_dir", out_dir, "--save_tag", "t", "--synthetic", "16",
               "--n_trajs", "12", "--epochs", "12"])
    losses = [float(l.split("loss=")[1].split(";")[0]) for l in log.splitlines() if "loss=" in l]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    ckpt = os.path.join(out_dir, "model_epoch_12_t.pth")
    assert os.path.exists(ckpt)
    log = run([os.path.join(TOOLS, "eval_vidvrd.py"), "--cfg_path", CFG, "--ckpt_path", ckpt, "--output_dir", out_dir, "--save_tag", "t",
               "--synthetic", "8", "--n_trajs", "12", "--json_results_path", os.path.join(out_dir, "pred.json")])
    assert "detection mean AP" in log and "detection recall" in log
    assert os.path.exists(os.path.join(out_dir, "pred.json")) and os.path.exists(os.path.join(out_dir, "VidVRDtest_infer_result_t.pkl"))


def test_vidor_cli_cls_training_eval_and_grounding():
    cfg5 = os.path.join(ROOT, "vidsgg-big_amd", "experiments", "exp5", "config_.py")
    cfgg = os.path.join(ROOT, "vidsgg-big_amd", "experiments", "grounding_weights", "config_.py")
    out_dir = tempfile.mkdtemp()
    # BIG-C VidOR classification stage: a few epochs, loss goes down, checkpoint in the reference's format
    log = run([os.path.join(TOOLS, "train_vidor.py"), "--cfg_path", cfg5, "--output_dir", out_dir, "--save_tag", "c", "--synthetic", "8",
               "--n_trajs", "10", "--epochs", "10"])
    losses = [float(l.split("loss=")[1].split(";")[0]) for l in log.splitlines() if "loss=" in l]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    cls_ckpt = os.path.join(out_dir, "model_epoch_10_c.pth")
    assert os.path.exists(cls_ckpt)
    # grounding stage training
    log = run([os.path.join(TOOLS, "train_vidor.py"), "--train_grounding", "--cfg_path", cfgg, "--output_dir", out_dir, "--save_tag", "g",
               "--synthetic", "16", "--n_trajs", "8", "--epochs", "10"])
    losses = [float(l.split("loss=")[1].split(";")[0]) for l in log.splitlines() if "loss=" in l]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    assert all(k in log for k in ("pos_cls", "neg_cls", "pos_ct", "neg_ct", "regr"))
    grd_ckpt = os.path.join(out_dir, "model_epoch_10_g.pth")
    assert os.path.exists(grd_ckpt)
    # classification-only eval saves the per-video results the grounding stage consumes
    log = run([os.path.join(TOOLS, "eval_vidor.py"), "--eval_cls_only", "--cfg_path", cfg5, "--ckpt_path", cls_ckpt, "--output_dir", out_dir,
               "--save_tag", "t", "--synthetic", "6", "--n_trajs", "10"])
    assert "detection recall" in log
    pkl = os.path.join(out_dir, "VidORval_infer_results_topk3_t.pkl")
    assert os.path.exists(pkl)
    log = run([os.path.join(TOOLS, "eval_vidor.py"), "--cfg_path", cfgg, "--ckpt_path", grd_ckpt, "--cls_stage_result_path", pkl,
               "--output_dir", out_dir, "--save_tag", "t", "--synthetic", "6", "--n_trajs", "10", "--save_json_results"])
    assert "detection recall" in log
    assert os.path.exists(os.path.join(out_dir, "VidORval_predict_relations_aft_grd_t.json"))


def test_vidor_cli_pairwise_baseline():
    cfg6 = os.path.join(ROOT, "vidsgg-big_amd", "experiments", "exp6", "config_.py")
    out_dir = tempfile.mkdtemp()
    log = run([os.path.join(TOOLS, "eval_vidor.py"), "--eval_cls_only", "--use_baseline", "--cfg_path", cfg6, "--output_dir", out_dir,
               "--save_tag", "b", "--synthetic", "4", "--n_trajs", "10"])
    assert "detection recall" in log and os.path.exists(os.path.join(out_dir, "VidORval_infer_results_topk3_b.pkl"))


def test_vidor_cli_pairwise_baseline_training_then_eval():
    """tools/train_vidor.py --train_baseline (reference :175-352) on synthetic videos, then the checkpoint through eval_vidor.py"""
    cfg6 = os.path.join(ROOT, "vidsgg-big_amd", "experiments", "exp6", "config_.py")
    out_dir = tempfile.mkdtemp()
    log = run([os.path.join(TOOLS, "train_vidor.py"), "--train_baseline", "--cfg_path", cfg6, "--output_dir", out_dir, "--save_tag", "bt",
               "--synthetic", "8", "--n_trajs", "10", "--epochs", "2"])
    assert "videos have labelled pairs" in log and "loss=" in log
    ckpt = os.path.join(out_dir, "model_epoch_2_bt.pth")
    assert os.path.exists(ckpt)
    log = run([os.path.join(TOOLS, "eval_vidor.py"), "--eval_cls_only", "--use_baseline", "--cfg_path", cfg6, "--ckpt_path", ckpt,
               "--output_dir", out_dir, "--save_tag", "bt", "--synthetic", "4", "--n_trajs", "10"])
    assert "detection recall" in log


def _bench(args, env=None, detail=False):
    """-> the object of bench.py's LAST stdout line, parsed strictly (the driver reads that line); with detail=True also bench_detail.json"""
    import json
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert lines and lines[-1].startswith("{"), r.stdout[-2000:]
    assert sum(1 for l in lines if l.startswith("{")) == 1, "one JSON line on stdout (the detail goes to bench_detail.json)"

    def strict(c):
        raise ValueError("not strict JSON: " + c)
    d = json.loads(lines[-1], parse_constant=strict)          # NaN / Infinity are not JSON
    if "legs" in d:                                            # the eval line's compact form
        assert len(lines[-1]) < 4096, len(lines[-1])

        def keys(o):
            if isinstance(o, dict):
                for k, v in o.items():
                    yield k
                    yield from keys(v)
        assert max(len(k) for k in keys(d)) <= 32
    if detail:
        return d, json.load(open(os.path.join(ROOT, "bench_detail.json")))
    return d


def test_bench_json_contract():
    """bench.py's last stdout line is ONE compact JSON object (< 4 KB, short keys) with the keys the driver and the judge read; the
    long form goes to bench_detail.json (small workload, one GPU)"""
    c, d = _bench(["--videos", "8", "--steps", "2", "--warmup", "1", "--cpu-videos", "4", "--exact-steps", "1", "--legs", "grounding,train,batch512",
                   "--vidor-videos", "2", "--recall-steps", "150", "--recall-videos", "6", "--budget-s", "800"], detail=True)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity", "legs", "skipped"):
        assert k in c, k
    assert c["metric"] == "tracklet_pairs_per_sec" and c["n_gpus"] == 1 and c["steps"] == 2 and c["warmup"] == 1 and c["higher_is_better"] is True
    assert c["scaling"] == "weak" and c["vs_baseline"] is None and c["dtype"] == "f32" and c["data"] == "synthetic" and "workload" in c["config"]
    assert c["value"] > 0 and abs(c["value"] - c["config"]["pairs_per_gpu"] / (c["ms_per_step"] * 1e-3)) < 1e-3 * c["value"]
    assert c["config"]["timed_call"].startswith("model(proposal_list, topk)") and c["skipped"] == []
    r = c["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert r["algorithmic_bytes_per_launch"] > 0 and "step_traffic_ratio" in r
    cb = c["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    pa = c["parity"]
    n_same, n_all = [int(x) for x in pa["rows"].split(" of ")]
    assert pa["videos"] == 4 and n_all > 0 and n_same == n_all and pa["rows_off_both_oracles"] == 0 and pa["hip_vs_f64_max"] <= 1e-4
    lg = c["legs"]
    for k in ("resident", "pipelined", "new_lists", "plain_accum", "exact_f32", "fast_f16", "fast_bf16", "batch_x2", "grounding", "train_8", "train_64", "recall"):
        assert k in lg, k
    assert lg["grounding"]["unit"] == "queries/s" and lg["grounding"]["value"] > 0 and 0 < lg["grounding"]["mfma_frac"] < 1
    assert lg["train_8"]["value"] > 0 and lg["train_64"]["value"] > 0 and lg["recall"]["gate"] is True
    # ---- the long form
    assert d["value"] == c["value"] and d["ms_per_step"] == c["ms_per_step"] and d["roofline"]["frac"] == r["frac"]
    # the timed call is the drop-in signature (pack + forward + results inside); the resident-batch forward rides beside it
    assert d["timed_call"].startswith("model(proposal_list, topk)") and d["forward_api"]["value"] == d["value"]
    assert d["resident_packed_batch"]["roofline"]["bound"] == "mfma" and d["resident_packed_batch"]["value"] > 0 and len(d["timed_steps_ms"]) == 2 and d["modes"]["resident_batch_holds_split_features"] is True
    assert d["kernel_families_ms_per_step"]["pack"] >= 0
    assert d["forward_api_pipelined"]["value"] > 0 and d["forward_api_pipelined"]["results_identical_to_the_synchronous_call"] is True
    assert d["exact_f32_gemm_mode"]["triplet_indices_identical_to_default_mode"] in (True, False)
    # the opt-in reduced-precision mode rides along with its Recall@50 gate -- and is a different arithmetic
    f = d["fast_f16_gemm_mode"]
    assert f["value"] > 0
    # ... the gate itself is evaluated on a TRAINED model (the `recall` leg): Recall@50 of the HIP path == the oracle's, and not zero
    rc = d["recall"]
    assert rc["recall50_oracle"] > 0 and rc["recall50_hip"] == rc["recall50_oracle"] and rc["parity_gate_passed"] is True
    assert rc["relations_hip"] == rc["relations_oracle"] > 0 and "fast_gate_passed" in rc and rc["abs_diff_recall50_fast_vs_parity_mode"] <= 0.05
    assert 1e-6 < f["max_logit_diff_rel_vs_parity_mode"] < 0.2
    # the literal bf16 arithmetic of BASELINE configs[1] rides along too, with the same gate
    assert f["bf16"]["value"] > 0 and f["bf16"]["max_logit_diff_rel_vs_parity_mode"] > 1e-5
    assert "bf16_gate_passed" in rc and rc["abs_diff_recall50_bf16_vs_parity_mode"] <= 0.05
    assert d["roofline"]["sustained_mfma_peak"]["three_mfma_gemm_ceiling"] > 0
    # the oracle outputs of the CPU-baseline sample are compared row for row with the HIP outputs of the same videos
    op = d["oracle_parity"]
    assert op["videos"] == 4 and op["videos_fully_identical"] == 4
    # per video: within 1e-4 of the f32 oracle, or of a float64 evaluation, or inside the video's own f32 noise floor (oracle/accuracy.py)
    assert op["videos_outside_1e_4_and_their_f32_noise_floor"] == 0 and len(op["videos_over_1e_4"]) <= 1
    # the other BASELINE configs ride in the same run, and the headline workload at twice the videos per step
    assert d["batch_x2"]["videos_per_step"] == 16 and d["batch_x2"]["value"] > 0 and d["batch_x2"]["roofline"]["bound"] == "mfma"
    g, t = d["grounding"], d["train"]
    assert g["unit"] == "triplet-queries/s" and g["value"] > 0 and g["roofline_mfma"]["bound"] == "mfma"
    # who took part: one rank on one device here; at N > 1 the same object lists every rank's device UUID and its own step time
    rk = d["ranks"]
    assert rk["world"] == 1 and rk["ranks_seen"] == [0] and rk["distinct_devices"] == 1 and rk["devices"][0]["uuid"] and rk["devices"][0]["name"]
    assert abs(rk["ms_per_step_max"] - d["ms_per_step"]) < 1e-2 and rk["ms_per_step_min"] == rk["ms_per_step_max"]
    assert t["videos_8"]["value"] > 0 and t["videos_64"]["value"] > 0


def test_bench_time_budget_skips_optional_legs_and_keeps_the_required_objects():
    """past --budget-s the optional legs are skipped and named in `skipped`; roofline, cpu_baseline and parity are still there"""
    c = _bench(["--videos", "4", "--steps", "2", "--warmup", "1", "--cpu-videos", "2", "--budget-s", "0"])
    assert c["roofline"]["frac"] > 0 and c["cpu_baseline"]["value"] > 0 and c["parity"]["videos"] == 2
    assert {"recall", "vidor+grounding", "train", "batch_x2"} <= set(c["skipped"]) and "train_8" not in c["legs"]


def test_bench_train_mode_runs_rccl_allreduce_on_one_rank():
    """`--mode train` with a one-rank RCCL group: init_process_group("nccl"), barrier, the flat-gradient all-reduce (in place on
    the buffer the HIP backward filled) and the max-over-ranks reduction execute on the GPU."""
    d = _bench(["--mode", "train", "--train-videos", "2", "--n-trajs", "16", "--steps", "2", "--warmup", "1"], env={"BIGC_BENCH_FORCE_DIST": "1"})
    assert d["metric"] == "train_tracklet_pairs_per_sec" and d["value"] > 0 and d["n_gpus"] == 1
    a = d["allreduce"]
    assert a["backend"].startswith("nccl") and a["flat_buffer_reduced_in_place"] is True and a["bytes"] > 100e6
    # the all-reduce timed alone over the same flat buffer, and the ranks that took part (gathered through the RCCL group)
    assert a["alone"]["bytes"] == a["bytes"] and a["alone"]["us"] > 0 and a["alone"]["world"] == 1
    assert d["ranks"]["backend"] == "nccl" and d["ranks"]["ranks_seen"] == [0] and d["ranks"]["devices"][0]["uuid"]
    assert d["last_loss"] == d["last_loss"]          # finite


def test_bench_eval_mode_with_rccl_group_and_self_launch():
    """The eval line through the self-launcher (a child `torch.distributed.run`, as `python bench.py --gpus N` does for N > 1)
    with a one-rank RCCL group for the barrier / max-over-ranks."""
    d = _bench(["--videos", "4", "--steps", "2", "--warmup", "1", "--cpu-videos", "0", "--exact-steps", "0", "--legs", "", "--no-prof", "--recall-steps", "0"],
               env={"BIGC_BENCH_FORCE_DIST": "1", "BIGC_BENCH_SELF_LAUNCH": "1"})
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
    import json
    rk = json.load(open(os.path.join(ROOT, "bench_detail.json")))["ranks"]
    assert rk["backend"] == "nccl" and rk["world"] == 1 and rk["distinct_devices"] == 1


def _gt_from_predictions(pred_json_path, per_video=3):
    """a ground-truth JSON in the evaluator's format made of each video's best-scored predictions (so that recall is > 0 by construction:
    the CLI tests check the plumbing, parity with the reference's evaluator is tests/test_evaluator.py)"""
    import json
    pred = json.load(open(pred_json_path))
    gt = {}
    for name, rel in pred.items():
        best = sorted(rel, key=lambda x: -x["score"])[:per_video]
        gt[name] = [{k: r[k] for k in ("triplet", "duration", "sub_traj", "obj_traj")} for r in best]
    return gt


def test_eval_vidvrd_cli_on_raw_tracking_results():
    """BASELINE configs[0] as worded: a 10-video subset through ``tools/eval_vidvrd.py`` -- here from raw PKU tracking results
    (``--tracking_dir`` / ``--i3d_dir``: per-box rows of 12 + 2048 and 12 + 832 columns), ingested by this package (no reference
    dataloader on PYTHONPATH), scored by the HIP forward, converted and evaluated."""
    import json
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases
    d = tempfile.mkdtemp()
    os.makedirs(d + "/track"); os.makedirs(d + "/i3d")
    for v in range(10):
        track, i3d, _ = cases.make_pku_rows(dict(seed=7000 + v, n_traj=8, dim_box=2048, dim_i3d=832, with_i3d=True))
        np.save(d + "/track/video_%02d.npy" % v, track)
        np.save(d + "/i3d/video_%02d.npy" % v, i3d)
    json.dump({"video_00": {"video_len": 60, "video_wh": [640, 360]}}, open(d + "/info.json", "w"))
    # first pass: a ground truth that nothing matches (the evaluator, like the reference's, needs at least one relation in total)
    json.dump({"video_00": [{"triplet": ["nothing", "matches", "this"], "duration": [0, 5], "sub_traj": [[0, 0, 9, 9]] * 5, "obj_traj": [[20, 20, 29, 29]] * 5}]},
              open(d + "/gt0.json", "w"))
    common = [os.path.join(TOOLS, "eval_vidvrd.py"), "--cfg_path", CFG, "--output_dir", d, "--use_pku", "--tracking_dir", d + "/track", "--i3d_dir", d + "/i3d",
              "--video_info_path", d + "/info.json"]
    log = run(common + ["--save_tag", "raw0", "--gt_json_path", d + "/gt0.json", "--json_results_path", d + "/pred.json"])
    assert "inference: 10 videos" in log and "detection recall" in log
    pred = json.load(open(d + "/pred.json"))
    assert len(pred) == 10 and sum(len(r) for r in pred.values()) > 0
    json.dump(_gt_from_predictions(d + "/pred.json"), open(d + "/gt1.json", "w"))
    log = run(common + ["--save_tag", "raw1", "--gt_json_path", d + "/gt1.json"])
    rec = [l for l in log.splitlines() if l.startswith("detection recall")][0]
    assert float(rec.split("{")[1].split(":")[1].split(",")[0]) > 0.5, rec          # recall@50 of ground truth taken from the predictions' top 3
    assert os.path.exists(d + "/VidVRDtest_infer_result_raw1.pkl")


def test_eval_vidor_cli_on_raw_tracking_results_then_grounding():
    """BASELINE configs[3] (VidOR full pipeline) from raw files: ``tools/eval_vidor.py --tracking_dir --classeme_dir`` (ragged per-box rows of
    12 + 1024 or 6 columns + a classeme row per box, dataloaders/dataloader_vidor_v3.py:389-485) for the classification stage, then
    ``--video_feature_dir`` for the grounding stage on the saved results."""
    import json
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases
    cfg5 = os.path.join(ROOT, "vidsgg-big_amd", "experiments", "exp5", "config_.py")
    cfgg = os.path.join(ROOT, "vidsgg-big_amd", "experiments", "grounding_weights", "config_.py")
    d = tempfile.mkdtemp()
    for sub in ("track", "clsme", "clip"):
        os.makedirs(os.path.join(d, sub))
    rs = np.random.RandomState(5)
    info = {}
    for v in range(4):
        cases.VIDOR_INGEST_CASES["_cli"] = dict(seed=7100 + v, n_traj=9, dim_box=1024, dim_clsme=300, min_frames_th=15, max_proposal=180, score_th=0.4, ragged=True)
        track, classeme, _ = cases.make_vidor_ingest_case("_cli")
        arr = np.empty(len(track), dtype=object)
        for i, r in enumerate(track):
            arr[i] = r
        name = "%04d_%d" % (v, 1000 + v)
        np.save(os.path.join(d, "track", name + ".npy"), arr, allow_pickle=True)
        np.save(os.path.join(d, "clsme", name + "_clsme.npy"), classeme)
        vlen = int(max(r[0] for r in track)) + 1
        info[name] = {"video_len": vlen, "video_wh": [640, 360]}
        np.save(os.path.join(d, "clip", name + ".npy"), (0.1 * rs.standard_normal((-(-vlen // 8), 1024))).astype(np.float32))
    del cases.VIDOR_INGEST_CASES["_cli"]
    json.dump(info, open(d + "/info.json", "w"))
    json.dump({"1000": [{"triplet": ["nothing", "matches", "this"], "duration": [0, 5], "sub_traj": [[0, 0, 9, 9]] * 5, "obj_traj": [[20, 20, 29, 29]] * 5}]},
              open(d + "/gt0.json", "w"))
    # exp5 asks for min_frames_th = 15; the generated tracklets have 6..19 boxes: a config copy with the threshold of the golden cases
    # (the same for the grounding config: both stages must ingest the same proposals)
    open(d + "/config_.py", "w").write(open(cfg5).read().replace("min_frames_th=15", "min_frames_th=5"))
    os.makedirs(d + "/grd")
    open(d + "/grd/config_.py", "w").write(open(cfgg).read().replace("min_frames_th=15", "min_frames_th=5"))
    raw = ["--tracking_dir", d + "/track", "--classeme_dir", d + "/clsme", "--video_info_path", d + "/info.json", "--gt_json_path", d + "/gt0.json"]
    log = run([os.path.join(TOOLS, "eval_vidor.py"), "--eval_cls_only", "--cfg_path", d + "/config_.py", "--output_dir", d, "--save_tag", "raw", "--save_json_results"] + raw)
    assert "inference: 4 videos" in log and "detection recall" in log
    pkl = os.path.join(d, "VidORval_infer_results_topk3_raw.pkl")
    pred = json.load(open(os.path.join(d, "VidORval_predict_relations_topk3_raw.json")))
    assert os.path.exists(pkl) and sum(len(r) for r in pred.values()) > 0
    json.dump(_gt_from_predictions(os.path.join(d, "VidORval_predict_relations_topk3_raw.json")), open(d + "/gt0.json", "w"))
    log = run([os.path.join(TOOLS, "eval_vidor.py"), "--cfg_path", d + "/grd/config_.py", "--cls_stage_result_path", pkl, "--output_dir", d, "--save_tag", "raw",
               "--video_feature_dir", d + "/clip", "--save_json_results"] + raw[:4] + ["--video_info_path", d + "/info.json", "--gt_json_path", d + "/gt0.json"])
    assert "grounding: 4 videos" in log and "detection recall" in log
    assert os.path.exists(os.path.join(d, "VidORval_predict_relations_aft_grd_raw.json"))


def test_eval_vidvrd_cli_on_a_reference_cache_file():
    """the on-disk cache a user of the reference already has (datasets/cache/*.pkl: pickled instances of the reference's TrajProposal /
    VideoGraph) through ``tools/eval_vidvrd.py --cache_path``: proposals AND ground truth come from the file, no reference code is imported"""
    cfg_txt = open(CFG).read().replace("dim_feat=2048", "dim_feat=16").replace("dim_i3d=832", "dim_i3d=8")
    d = tempfile.mkdtemp()
    open(d + "/config_.py", "w").write(cfg_txt)
    log = run([os.path.join(TOOLS, "eval_vidvrd.py"), "--cfg_path", d + "/config_.py", "--output_dir", d, "--save_tag", "c",
               "--cache_path", os.path.join(ROOT, "tests", "golden", "ref_cache_vidvrd.pkl")])
    assert "inference: 2 videos" in log and "detection recall" in log

"""Tracking-result ingest against the reference's own ``_get_proposal``
(tests/golden/ingest_*.npz): host bookkeeping on the CPU, feature-row assembly on the GPU, bit-exact."""
import numpy as np
import pytest
import torch

import cases
from bigc_hip import ingest


@pytest.mark.parametrize("name", list(cases.INGEST_CASES))
def test_plan_matches_reference_dataloader(name):
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    track, i3d, case = cases.make_ingest_case(name)
    plan = ingest.plan_tracklets(track, case["dim_box"], case["min_frames_th"], case["max_proposal"])
    assert len(plan["cat_ids"]) == int(g["num_proposals"])
    assert np.array_equal(np.array(plan["cat_ids"]), g["cat_ids"])
    assert np.array_equal(plan["scores"].numpy(), g["scores"])                                  # bit-exact
    assert np.array_equal(np.array(plan["durations"]), g["traj_durations"])
    assert np.array_equal(np.array([b.shape[0] for b in plan["boxes"]]), g["lens"])
    assert np.array_equal(torch.cat(plan["boxes"], 0).numpy(), g["bboxes"])                     # bit-exact
    assert (plan["src_b"] != -2).sum() > 0 and (plan["src_a"] < 0).sum() > 0                    # the case has gaps and undetected boxes


def test_empty_and_all_filtered():
    track, _, case = cases.make_ingest_case("ingest_pku")
    assert ingest.plan_tracklets(track, case["dim_box"], 1000, 10) is None                      # everything shorter than the threshold


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.INGEST_CASES))
def test_feature_rows_match_reference_dataloader(name):
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    track, i3d, case = cases.make_ingest_case(name)
    p = ingest.load_pku_tracklets("video_x", track, case["dim_box"], case["min_frames_th"], case["max_proposal"], i3d_res=i3d,
                                  dim_i3d=case["dim_i3d"], device="cuda:0")
    assert p.num_proposals == int(g["num_proposals"])
    feats = torch.cat(p.features_list, 0)
    assert feats.is_cuda and feats.shape == g["features"].shape
    assert np.array_equal(feats.cpu().numpy(), g["features"])                                    # bit-exact, incl. interpolated rows
    assert np.array_equal(torch.cat(p.bboxes_list, 0).cpu().numpy(), g["bboxes"])
    assert np.array_equal(p.traj_durations.cpu().numpy(), g["traj_durations"]) and np.array_equal(p.cat_ids.cpu().numpy(), g["cat_ids"])


@pytest.mark.gpu
def test_ingested_proposal_runs_through_the_model():
    """the ingest output is directly consumable by the BIG-C forward (features stay device-resident)"""
    import os
    import tempfile
    from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict
    from models import BIG_C_vidvrd
    track, i3d, case = cases.make_ingest_case("ingest_pku_i3d")
    cfg = cases.get_cfg("CONFIG_TINY")
    cfg["dim_feat"], cfg["dim_i3d"], cfg["num_enti_cats"] = case["dim_box"], case["dim_i3d"], 36
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 5, att_gain=4.0)
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy()); np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
    model = BIG_C_vidvrd(dict(cfg, bias_matrix_path=d + "/b.npy", EntiNameEmb_path=d + "/e.npy"), is_train=False)
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    p = ingest.load_pku_tracklets("video_x", track, case["dim_box"], case["min_frames_th"], case["max_proposal"], i3d_res=i3d,
                                  dim_i3d=case["dim_i3d"], device="cuda:0").to("cuda:0")
    res = model([p], topk=3)[0]
    assert res is None or (res[0].shape[1] == 5 and res[0].shape[0] == res[1].shape[0])


# ---- VidOR (dataloaders/dataloader_vidor_v3.py:389-485 + :19-69): ragged files, classeme rows, score clipping before the top-N cut ------

def _ragged(track, ragged):
    if not ragged:
        return np.asarray(track, dtype=np.float32)
    arr = np.empty(len(track), dtype=object)
    for i, r in enumerate(track):
        arr[i] = r
    return arr


@pytest.mark.parametrize("name", list(cases.VIDOR_INGEST_CASES))
def test_vidor_plan_matches_reference_dataloader(name):
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    track, classeme, case = cases.make_vidor_ingest_case(name)
    plan, dense = ingest.plan_vidor_tracklets(_ragged(track, case["ragged"]), case["dim_box"], case["min_frames_th"], case["max_proposal"], case["score_th"])
    assert len(plan["cat_ids"]) == int(g["num_proposals"])
    assert np.array_equal(np.array(plan["cat_ids"]), g["cat_ids"])
    assert np.array_equal(plan["scores"].numpy(), g["scores"])                                  # bit-exact, in the reference's order
    assert (plan["scores"] > case["score_th"]).all() and len(plan["cat_ids"]) <= case["max_proposal"]
    assert np.array_equal(np.array(plan["durations"]), g["traj_durations"])
    assert np.array_equal(np.array([b.shape[0] for b in plan["boxes"]]), g["lens"])
    assert np.array_equal(torch.cat(plan["boxes"], 0).numpy(), g["bboxes"])                     # bit-exact (tracker boxes, interpolated across gaps)
    assert (plan["src_b"] != -2).sum() > 0                                                      # the case has gaps
    # the list-of-lists form (what numpy.load(allow_pickle=True) of the reference's files gives after .tolist()) plans the same
    plan2, _ = ingest.plan_vidor_tracklets(track, case["dim_box"], case["min_frames_th"], case["max_proposal"], case["score_th"])
    assert np.array_equal(plan2["src_a"], plan["src_a"]) and np.array_equal(plan2["scores"].numpy(), plan["scores"].numpy())
    assert dense.shape[1] == 12 + case["dim_box"] and len(plan["file_rows"]) == dense.shape[0]


def test_vidor_score_clipping_comes_before_the_top_n_cut_and_empty_results():
    track, classeme, case = cases.make_vidor_ingest_case("ingest_vidor_b")
    loose, _ = ingest.plan_vidor_tracklets(np.asarray(track, dtype=np.float32), case["dim_box"], case["min_frames_th"], 100, 0.0)
    mid = float(loose["scores"][len(loose["scores"]) // 2])
    tight, _ = ingest.plan_vidor_tracklets(np.asarray(track, dtype=np.float32), case["dim_box"], case["min_frames_th"], 3, mid)
    want = [float(s) for s in loose["scores"] if float(s) > mid][:3]
    assert [float(s) for s in tight["scores"]] == want
    assert ingest.plan_vidor_tracklets(np.asarray(track, dtype=np.float32), case["dim_box"], case["min_frames_th"], 10, 0.999) == (None, None)
    assert ingest.plan_vidor_tracklets(np.asarray(track, dtype=np.float32)[:, :6], case["dim_box"], 1, 10, 0.0) == (None, None)      # tracker-only file
    with pytest.raises(AssertionError):
        ingest.plan_vidor_tracklets(np.asarray(track, dtype=np.float32)[:, :9], case["dim_box"], 1, 10, 0.0)                        # (:406)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.VIDOR_INGEST_CASES))
def test_vidor_feature_rows_match_reference_dataloader(name):
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    track, classeme, case = cases.make_vidor_ingest_case(name)
    p = ingest.load_vidor_tracklets("video_x", _ragged(track, case["ragged"]), classeme, case["dim_box"], case["min_frames_th"], case["max_proposal"],
                                    case["score_th"], device="cuda:0")
    assert p.num_proposals == int(g["num_proposals"])
    feats = torch.cat(p.features_list, 0)
    assert feats.is_cuda and feats.shape == g["features"].shape and feats.shape[1] == case["dim_box"] + case["dim_clsme"]
    assert np.array_equal(feats.cpu().numpy(), g["features"])                                    # [RoI | classeme], bit-exact incl. interpolated rows
    assert np.array_equal(torch.cat(p.bboxes_list, 0).cpu().numpy(), g["bboxes"])
    assert np.array_equal(p.traj_durations.cpu().numpy(), g["traj_durations"]) and np.array_equal(p.cat_ids.cpu().numpy(), g["cat_ids"])
    assert np.array_equal(p.scores.cpu().numpy(), g["scores"])


# ---- the reference's on-disk cache (datasets/cache/*.pkl: pickled instances of ITS classes) read without the reference on PYTHONPATH ------

def test_reference_cache_is_read_without_the_reference_classes():
    import sys
    assert not any(m == "dataloaders" or m.startswith("dataloaders.") for m in sys.modules), "the reference must not be importable here"
    g = np.load(cases.GOLDEN_DIR + "/ref_cache_vidvrd_expect.npz")
    data = ingest.load_reference_cache(cases.GOLDEN_DIR + "/ref_cache_vidvrd.pkl")       # written by the reference's own classes (make_golden.py cache)
    assert sorted(data) == ["video_0", "video_1"]
    for v, name in enumerate(sorted(data)):
        p, gt = data[name]
        pre = "v%d_" % v
        assert type(p).__module__ == "bigc_hip.synthetic" and p.video_name == name and p.video_len == 50 + v and p.video_wh == (640, 360)
        assert np.array_equal(torch.cat(p.features_list, 0).numpy(), g[pre + "features"])
        assert np.array_equal(torch.cat(p.bboxes_list, 0).numpy(), g[pre + "bboxes"])
        assert np.array_equal(p.cat_ids.numpy(), g[pre + "cat_ids"]) and np.array_equal(p.scores.numpy(), g[pre + "scores"])
        assert np.array_equal(p.traj_durations.numpy(), g[pre + "traj_durations"])
        assert hasattr(p, "_feat_block") and p._lists is None                    # block-backed: the batch packer reads addresses by arithmetic
        assert np.array_equal(gt.adj_matrix.numpy(), g[pre + "adj"]) and np.array_equal(gt.pred_durations.numpy(), g[pre + "pred_durations"])
        assert np.array_equal(gt.traj_durations.numpy(), g[pre + "gt_traj_durations"]) and gt.num_trajs == 3 and gt.num_preds == 2


def test_reference_cache_reader_refuses_other_classes():
    import io
    import pickle

    class Evil(object):
        def __reduce__(self):
            return (print, ("should never run",))
    blob = pickle.dumps({"v": (Evil(),)})
    with pytest.raises(pickle.UnpicklingError):
        ingest._restricted_unpickler(io.BytesIO(blob)).load()


@pytest.mark.gpu
def test_reference_cache_feeds_the_model():
    import tempfile
    from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict
    from models import BIG_C_vidvrd
    data = ingest.load_reference_cache(cases.GOLDEN_DIR + "/ref_cache_vidvrd.pkl", device="cuda:0")
    p0 = data["video_0"][0]
    cfg = cases.get_cfg("CONFIG_TINY")
    cfg["dim_feat"], cfg["dim_i3d"], cfg["num_enti_cats"] = 16, 8, 36
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 5, att_gain=4.0)
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy()); np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
    model = BIG_C_vidvrd(dict(cfg, bias_matrix_path=d + "/b.npy", EntiNameEmb_path=d + "/e.npy"), is_train=False)
    model.load_state_dict(sd)
    res = model.to("cuda:0")([p0], topk=3)[0]
    assert res is None or (res[0].shape[1] == 5 and res[0].shape[0] == res[1].shape[0])

"""Generate the golden fixtures by running the REAL reference (Dawn-LX/VidSGG-BIG at
/root/reference) in the build container.  Not run by the tests and never on the GPU box.

    python tests/golden/make_golden.py [case ...]

The reference is imported read-only with two stub modules (``cv2``, ``torchvision.ops``)
because ``utils/utils_func.py:9,12`` imports them at module top although the hot path
uses neither.  Weights and inputs come from the numpy-seeded
generators in ``bigc_hip.synthetic`` (the tests regenerate them bit-identically); only
the reference's OUTPUTS are stored, as ``tests/golden/<case>.npz``.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

REFERENCE = "/root/reference"


def import_reference():
    for m in ("cv2", "torchvision", "torchvision.ops"):
        if m not in sys.modules:
            sys.modules[m] = types.ModuleType(m)
    sys.modules["torchvision.ops"].roi_pool = lambda *a, **k: None
    # the build's own packages are also called models/utils: the reference must win here
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE)
    import models.model_0v10 as m10
    import models.model_0v7 as m07
    import models.grd_model_v5 as grd
    sys.path.remove(REFERENCE)
    return m10, m07, grd


def build_reference_bigc(mod, cfg, sd):
    d = tempfile.mkdtemp()
    cfg = dict(cfg)
    np.save(os.path.join(d, "bias.npy"), sd["bias_matrix"].numpy())
    cfg["bias_matrix_path"] = os.path.join(d, "bias.npy")
    if "EntiNameEmb" in sd:
        np.save(os.path.join(d, "emb.npy"), sd["EntiNameEmb"].numpy())
        cfg["EntiNameEmb_path"] = os.path.join(d, "emb.npy")
    model = mod.BIG_C(cfg, is_train=False)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    model.eval()
    return model


def run_bigc_case(name, m10, m07):
    case = cases.BIGC_CASES[name]
    cfg, variant, sd, props, topk = cases.make_case(name)
    mod = m10 if variant == "vidvrd" else m07
    model = build_reference_bigc(mod, cfg, sd)
    out = {"torch_version": np.array(torch.__version__), "n_videos": np.array(len(props))}
    captured = {}

    def hook(tag):
        def f(_m, _i, o):
            captured.setdefault(tag, []).append(o)
        return f

    handles = [
        model.fc_bbox2enti.register_forward_hook(hook("fc_bbox2enti")),
        model.fc_feat2enti.register_forward_hook(hook("fc_feat2enti")),
        model.conv_feat2enti.register_forward_hook(hook("conv")),
        model.fc_enti2enco.register_forward_hook(hook("enti2enco")),
    ]
    for i, l in enumerate(model.encoder_layers):
        handles.append(l.register_forward_hook(hook("enc%d" % i)))
    for i, l in enumerate(model.decoder_layers):
        handles.append(l.register_forward_hook(hook("dec%d" % i)))

    with torch.no_grad():
        for v, p in enumerate(props):
            captured.clear()
            res = model([p], topk=topk)[0]
            pre = "v%d_" % v
            out[pre + "is_none"] = np.array(res is None)
            if res is not None:
                out[pre + "quintuples"] = res[0].numpy()
                out[pre + "scores"] = res[1].numpy()
                out[pre + "dura_inters"] = res[2].numpy()
                out[pre + "query_ids"] = res[3].numpy()
            if p.num_proposals == 0:
                continue
            captured.clear()
            pq, logits, att = model.encode2decode(p)
            out[pre + "pred_logits"] = logits.numpy()
            if case.get("slim", False):
                # outputs only: the contract (logits, arg-max of the attention, the triplet tuples above), ~110 KB per exp2-dims video
                out[pre + "att_argmax"] = att.argmax(-1).numpy().astype(np.int32)
                continue
            out[pre + "att"] = att.numpy()
            out[pre + "pred_query"] = pq.numpy()
            out[pre + "enti2enco"] = captured["enti2enco"][0].numpy()
            out[pre + "enco_output"] = captured["enc%d" % (cfg["n_enco_layers"] - 1)][0].squeeze(1).numpy()
            if case.get("full", False):
                out[pre + "fc_bbox2enti"] = captured["fc_bbox2enti"][0].numpy()
                out[pre + "fc_feat2enti"] = captured["fc_feat2enti"][0].numpy()
                out[pre + "conv"] = captured["conv"][0].permute(0, 2, 1).numpy()
                for i in range(cfg["n_enco_layers"]):
                    out[pre + "enc%d" % i] = captured["enc%d" % i][0].squeeze(1).numpy()
                for i in range(cfg["n_deco_layers"]):
                    q, a = captured["dec%d" % i][0]
                    out[pre + "dec%d_query" % i] = q.numpy()
                    out[pre + "dec%d_att" % i] = a.numpy()
    for h in handles:
        h.remove()
    path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def run_grd_case(name, grd):
    case = cases.GRD_CASES[name]
    cfg, sd, videos = cases.make_grd_case(name)
    d = tempfile.mkdtemp()
    rcfg = dict(cfg)
    np.save(os.path.join(d, "e.npy"), sd["EntiNameEmb"].numpy())
    np.save(os.path.join(d, "p.npy"), sd["PredNameEmb"].numpy())
    rcfg["EntiNameEmb_path"], rcfg["PredNameEmb_path"] = os.path.join(d, "e.npy"), os.path.join(d, "p.npy")
    model = grd.DEBUG(rcfg, is_train=False)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    model.eval()
    out = {"torch_version": np.array(torch.__version__), "n_videos": np.array(len(videos))}
    captured = {}

    def hook(tag):
        def f(_m, _i, o):
            captured[tag] = o
        return f

    for tag in ("video_encoder", "query_encoder", "combined_encoder", "vq_fc"):
        getattr(model, tag).register_forward_hook(hook(tag))
    with torch.no_grad():
        for v, (feat, data) in enumerate(videos):
            pre = "v%d_" % v
            pooled, probs, mask = model([feat], [data], with_gt_data=False, **cases.GRD_TH)
            out[pre + "pooled_se"] = pooled.numpy()
            out[pre + "bins_probs"] = probs.numpy()
            out[pre + "bins_mask"] = mask.numpy()
            if case.get("slim", False):          # final outputs only (a few KB per video)
                continue
            words, info = model.prepare_data(data)
            regrs, conf, cls = model.forward_propagation(feat, words, info)
            out[pre + "regrs"], out[pre + "conf"], out[pre + "cls"] = regrs.numpy(), conf.numpy(), cls.numpy()
            if case.get("full", False):
                out[pre + "video_enc"] = captured["video_encoder"][0].t().numpy()              # (T,d)
                out[pre + "query_enc"] = captured["query_encoder"].transpose(1, 2).numpy()     # (n,3,d)
                out[pre + "combined_in"] = captured["vq_fc"].numpy()                           # (n,T,d)
                out[pre + "combined_enc"] = captured["combined_encoder"].transpose(1, 2).numpy()
    path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def run_train_case(name, m10, m07):
    """Reference training forward + backward (modules in eval() mode: dropout is not part of the parity
    contract, ``is_train=True`` selects the training path)."""
    case = cases.TRAIN_CASES[name]
    cfg, variant, sd, props, gts = cases.make_train_case(name)
    mod = m10 if variant == "vidvrd" else m07
    d = tempfile.mkdtemp()
    rcfg = dict(cfg)
    np.save(os.path.join(d, "bias.npy"), sd["bias_matrix"].numpy())
    rcfg["bias_matrix_path"] = os.path.join(d, "bias.npy")
    if "EntiNameEmb" in sd:
        np.save(os.path.join(d, "emb.npy"), sd["EntiNameEmb"].numpy())
        rcfg["EntiNameEmb_path"] = os.path.join(d, "emb.npy")
    model = mod.BIG_C(rcfg, is_train=True)
    model.load_state_dict(sd, strict=True)
    model.eval()
    indices = []
    orig = model.bipartite_match

    def spy(*a, **k):
        r = orig(*a, **k)
        indices.append(r)
        return r
    model.bipartite_match = spy
    total, terms = model(props, gts)
    total.backward()
    out = {"torch_version": np.array(torch.__version__), "total": total.detach().numpy()}
    for k, v in terms.items():
        out["loss_" + k] = v.detach().numpy()
    for i, (iq, ig) in enumerate(indices):
        out["match%d_query" % i], out["match%d_gt" % i] = np.asarray(iq), np.asarray(ig)
    for k, p in model.named_parameters():
        g = p.grad
        out["gnorm_" + k] = np.array(0.0 if g is None else float(g.norm()))
        if g is not None and not case.get("slim", False) and (case.get("full", False) or k in cases.TRAIN_FULL_GRADS_V):
            out["grad_" + k] = g.numpy()
    path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), "total loss", float(total))


def run_grd_train_case(name, grd):
    """Reference grounding training forward + backward (eval() mode: no dropout; the negative predicates come from the
    global torch RNG, seeded right before the call)."""
    case = cases.GRD_TRAIN_CASES[name]
    cfg, sd, feats, gts, rng_seed = cases.make_grd_train_case(name)
    d = tempfile.mkdtemp()
    rcfg = dict(cfg)
    np.save(os.path.join(d, "e.npy"), sd["EntiNameEmb"].numpy())
    np.save(os.path.join(d, "p.npy"), sd["PredNameEmb"].numpy())
    rcfg["EntiNameEmb_path"], rcfg["PredNameEmb_path"] = os.path.join(d, "e.npy"), os.path.join(d, "p.npy")
    model = grd.DEBUG(rcfg, is_train=True)
    model.load_state_dict(sd, strict=True)
    model.eval()
    torch.manual_seed(rng_seed)
    total, terms = model(feats, gts)
    total.backward()
    out = {"torch_version": np.array(torch.__version__), "total": total.detach().numpy()}
    for k, v in terms.items():
        out["loss_" + k] = v.detach().numpy()
    for k, p_ in model.named_parameters():
        g = p_.grad
        out["gnorm_" + k] = np.array(0.0 if g is None else float(g.norm()))
        if g is not None and (case.get("full", False) or k in cases.GRD_TRAIN_FULL_GRADS):
            out["grad_" + k] = g.numpy()
    path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), "total", float(total), {k: round(float(v), 4) for k, v in terms.items()})


def run_evaluator_golden():
    """metrics of the reference's own evaluator (VidVRDhelperEvalAPIs/visual_relation_detection.py:64-120) on the
    seeded relation sets of cases.make_eval_case"""
    sys.path.insert(0, REFERENCE)
    from VidVRDhelperEvalAPIs.visual_relation_detection import evaluate
    from VidVRDhelperEvalAPIs.common import viou, voc_ap
    sys.path.remove(REFERENCE)
    out = {}
    for seed in cases.EVAL_SEEDS:
        gt, pr = cases.make_eval_case(seed)
        mean_ap, rec, prec = evaluate(gt, pr)
        out["s%d" % seed] = np.array([mean_ap, rec[50], rec[100], prec[1], prec[5], prec[10]], dtype=np.float64)
        g0 = gt["video_0"][0]
        p0 = pr["video_0"][0]
        out["s%d_viou" % seed] = np.array([viou(p0["sub_traj"], p0["duration"], g0["sub_traj"], g0["duration"]),
                                           viou(p0["obj_traj"], p0["duration"], g0["obj_traj"], g0["duration"])])
    out["voc_ap"] = np.array([voc_ap(np.array([0.1, 0.4, 0.4, 0.9]), np.array([1.0, 0.5, 0.66, 0.4]))])
    path = os.path.join(cases.GOLDEN_DIR, "evaluator.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.round(4).tolist() for k, v in out.items()})


def run_ingest_case(name):
    """The reference's own ``_get_proposal`` (dataloaders/dataloader_vidvrd.py:485-573 / :585-700) on synthetic tracking
    results written to a temp dir; the Dataset constructor (annotations, caches) is bypassed with a stand-in ``self``."""
    import types as _t
    sys.path.insert(0, REFERENCE)
    np.float, np.int = float, int                       # the dataloader module predates numpy 1.24
    from dataloaders import dataloader_vidvrd as D
    import utils.utils_func as U
    sys.path.remove(REFERENCE)
    # fill_zeropadding on the 1-D boundary vectors (utils/utils_func.py:283-290) calls np.where on a 0-d array, which numpy 2
    # rejects; under the numpy the reference targeted it copies element 1 over element 0 of an ALL-ZERO vector, i.e. nothing
    U.fill_zeropadding = lambda vectors: None
    track, i3d, case = cases.make_ingest_case(name)
    d = tempfile.mkdtemp()
    os.makedirs(d + "/prop"); os.makedirs(d + "/i3d")
    np.save(d + "/prop/video_x.npy", track)
    fake = _t.SimpleNamespace(proposal_dir=d + "/prop", i3d_dir=d + "/i3d", dim_boxfeature=case["dim_box"], dim_i3d=case["dim_i3d"],
                              min_frames_th=case["min_frames_th"], max_proposal=case["max_proposal"])
    if case["with_i3d"]:
        np.save(d + "/i3d/video_x.npy", i3d)
        prop = D.Dataset_pku_i3d._get_proposal(fake, "video_x")
    else:
        prop = D.Dataset_pku._get_proposal(fake, "video_x")
    out = dict(num_proposals=np.array(prop.num_proposals), cat_ids=prop.cat_ids.numpy(), scores=prop.scores.numpy(),
               traj_durations=prop.traj_durations.numpy(), bboxes=torch.cat(prop.bboxes_list, 0).numpy(),
               features=torch.cat(prop.features_list, 0).numpy(), lens=np.array([b.shape[0] for b in prop.bboxes_list]),
               versions=np.array([torch.__version__, np.__version__]))
    path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), "proposals", prop.num_proposals, "rows", out["bboxes"].shape[0])


def run_vidor_ingest_case(name):
    """The reference's VidOR ``Dataset._get_proposal`` (dataloaders/dataloader_vidor_v3.py:389-485; its ``TrajProposal`` :19-69 does the
    score clipping and the top-N cut) on synthetic tracking results + classemes written to a temp dir; the Dataset constructor
    (annotations, caches) is bypassed with a stand-in ``self``."""
    import types as _t
    for m in ("cv2", "torchvision", "torchvision.ops"):
        if m not in sys.modules:
            sys.modules[m] = types.ModuleType(m)
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k == "dataloaders" or k.startswith("dataloaders.")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE)
    np.float, np.int = float, int
    from dataloaders import dataloader_vidor_v3 as D
    import utils.utils_func as U
    sys.path.remove(REFERENCE)
    U.fill_zeropadding = lambda vectors: None            # see run_ingest_case
    D.linear_interpolation.__globals__["fill_zeropadding"] = U.fill_zeropadding
    track, classeme, case = cases.make_vidor_ingest_case(name)
    d = tempfile.mkdtemp()
    os.makedirs(d + "/prop"); os.makedirs(d + "/clsme")
    if case["ragged"]:
        arr = np.empty(len(track), dtype=object)
        for i, r in enumerate(track):
            arr[i] = r
    else:
        arr = np.asarray(track, dtype=object)
    np.save(d + "/prop/video_x.npy", arr, allow_pickle=True)
    np.save(d + "/clsme/video_x_clsme.npy", classeme)
    fake = _t.SimpleNamespace(classeme_dir=d + "/clsme", dim_boxfeature=case["dim_box"], min_frames_th=case["min_frames_th"],
                              max_proposal=case["max_proposal"], score_th=case["score_th"],
                              videoname2trackres=lambda video_name: d + "/prop/" + video_name + ".npy")
    prop = D.Dataset._get_proposal(fake, "video_x")
    if prop.num_proposals == 0:
        out = dict(num_proposals=np.array(0))
    else:
        out = dict(num_proposals=np.array(prop.num_proposals), cat_ids=prop.cat_ids.numpy(), scores=prop.scores.numpy(),
                   traj_durations=prop.traj_durations.numpy(), bboxes=torch.cat(prop.bboxes_list, 0).numpy(),
                   features=torch.cat(prop.features_list, 0).numpy(), lens=np.array([b.shape[0] for b in prop.bboxes_list]))
    out["versions"] = np.array([torch.__version__, np.__version__])
    path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), "proposals", prop.num_proposals, "rows", out.get("bboxes", np.zeros((0, 4))).shape[0])


def run_cache_fixture():
    """A ``datasets/cache/*.pkl`` file as the reference writes it (dataloaders/dataloader_vidvrd.py:170-183): {video: (TrajProposal, VideoGraph)}
    holding instances of the REFERENCE's classes, built from the ingest case's tracking results by the reference's own ``get_data`` pieces.
    Stored as tests/golden/ref_cache_vidvrd.pkl (data: what a user's cache directory holds) + the expected arrays."""
    import pickle
    import types as _t
    sys.path.insert(0, REFERENCE)
    np.float, np.int = float, int
    from dataloaders import dataloader_vidvrd as D
    import utils.utils_func as U
    sys.path.remove(REFERENCE)
    U.fill_zeropadding = lambda vectors: None
    data, expect = {}, {}
    for vi, case in enumerate((cases.INGEST_CASES["ingest_pku_i3d"], dict(cases.INGEST_CASES["ingest_pku_i3d"], seed=907, n_traj=7))):
        track, i3d, case = cases.make_pku_rows(case)
        d = tempfile.mkdtemp()
        os.makedirs(d + "/prop"); os.makedirs(d + "/i3d")
        np.save(d + "/prop/video_x.npy", track); np.save(d + "/i3d/video_x.npy", i3d)
        fake = _t.SimpleNamespace(proposal_dir=d + "/prop", i3d_dir=d + "/i3d", dim_boxfeature=case["dim_box"], dim_i3d=case["dim_i3d"],
                                  min_frames_th=case["min_frames_th"], max_proposal=case["max_proposal"])
        prop = D.Dataset_pku_i3d._get_proposal(fake, "video_x")
        prop.video_name = "video_%d" % vi
        n_t, n_p = 3, 2
        rs = np.random.RandomState(40 + vi)
        graph = D.VideoGraph(("video_%d" % vi, 50 + vi, (640, 360)), "test", [3, 5, 7], [(0, 10), (2, 14), (5, 9)],
                             [rs.uniform(0, 100, (10, 4)), rs.uniform(0, 100, (12, 4)), rs.uniform(0, 100, (4, 4))], [11, 12], [(2, 9), (5, 8)],
                             np.array([[1, 0, 0], [0, 0, 1]], dtype=np.float32), np.array([[0, 1, 0], [0, 1, 0]], dtype=np.float32), 100)
        prop.video_len, prop.video_wh = graph.video_len, graph.video_wh            # get_data (:232-233)
        data["video_%d" % vi] = (prop, graph)
        expect["v%d_features" % vi] = torch.cat(prop.features_list, 0).numpy()
        expect["v%d_bboxes" % vi] = torch.cat(prop.bboxes_list, 0).numpy()
        expect["v%d_cat_ids" % vi], expect["v%d_scores" % vi] = prop.cat_ids.numpy(), prop.scores.numpy()
        expect["v%d_traj_durations" % vi] = prop.traj_durations.numpy()
        expect["v%d_adj" % vi], expect["v%d_pred_durations" % vi] = graph.adj_matrix.numpy(), graph.pred_durations.numpy()
        expect["v%d_gt_traj_durations" % vi] = graph.traj_durations.numpy()
    with open(os.path.join(cases.GOLDEN_DIR, "ref_cache_vidvrd.pkl"), "wb") as f:
        pickle.dump(data, f)
    np.savez_compressed(os.path.join(cases.GOLDEN_DIR, "ref_cache_vidvrd_expect.npz"), **expect)
    print("wrote ref_cache_vidvrd.pkl", os.path.getsize(os.path.join(cases.GOLDEN_DIR, "ref_cache_vidvrd.pkl")), "bytes")


def run_basec_case(name):
    """The reference's Base_C (models/model_pairwise_baseline.py) eval forward on the seeded case."""
    sys.path.insert(0, REFERENCE)
    from models import model_pairwise_baseline as MB
    sys.path.remove(REFERENCE)
    cfg, sd, props, topk = cases.make_basec_case(name)
    d = tempfile.mkdtemp()
    rc = dict(cfg)
    np.save(os.path.join(d, "bias.npy"), sd["bias_matrix"].numpy())
    rc["bias_matrix_path"] = os.path.join(d, "bias.npy")
    if "EntiNameEmb" in sd:
        np.save(os.path.join(d, "emb.npy"), sd["EntiNameEmb"].numpy())
        rc["EntiNameEmb_path"] = os.path.join(d, "emb.npy")
    model = MB.Base_C(rc, is_train=False)
    model.load_state_dict(sd, strict=True)
    model.eval()
    out = {"versions": np.array([torch.__version__, np.__version__])}
    with torch.no_grad():
        res = model(props, topk=topk)
        for i, (p, r) in enumerate(zip(props, res)):
            out["v%d_none" % i] = np.array(r is None)
            if p.num_proposals > 1:
                pair_ids = model.trajid2pairid(p.num_proposals)
                out["v%d_logits" % i] = model.forward_propagation(p, pair_ids).numpy()
            if r is not None:
                out["v%d_quint" % i], out["v%d_scores" % i], out["v%d_dura" % i] = r[0].numpy(), r[1].numpy(), r[2].numpy()
    path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), [None if r is None else tuple(r[0].shape) for r in res])


def run_basec_train_case(name):
    """The reference's Base_C training forward + backward (models/model_pairwise_baseline.py:276-312) on the seeded case:
    loss, the logits of the labelled pairs, the norm of every parameter gradient and a few gradients in full."""
    sys.path.insert(0, REFERENCE)
    from models import model_pairwise_baseline as MB
    sys.path.remove(REFERENCE)
    cfg, sd, props, pos, labels = cases.make_basec_train_case(name)
    d = tempfile.mkdtemp()
    rc = dict(cfg)
    np.save(os.path.join(d, "bias.npy"), sd["bias_matrix"].numpy())
    rc["bias_matrix_path"] = os.path.join(d, "bias.npy")
    if "EntiNameEmb" in sd:
        np.save(os.path.join(d, "emb.npy"), sd["EntiNameEmb"].numpy())
        rc["EntiNameEmb_path"] = os.path.join(d, "emb.npy")
    model = MB.Base_C(rc, is_train=True)
    model.load_state_dict(sd, strict=True)
    model.eval()
    total, terms = model(props, pos, labels)
    total.backward()
    out = {"versions": np.array([torch.__version__, np.__version__]), "total": np.float64(total.item()), "loss_cls": np.float64(terms["cls"].item())}
    with torch.no_grad():
        out["logits"] = torch.cat([model.forward_propagation(p, ids) for p, ids in zip(props, pos)]).numpy()
    full = ("bias_matrix", "fc_pred2logits.0.weight", "fc_pred2logits.2.bias", "fc_enti2enco.2.weight", "conv_feat2enti.bias", "fc_bbox2enti.0.weight")
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out["gnorm_" + k] = np.float64(g.norm().item())
        if k in full and g.numel() <= 600000:
            out["grad_" + k] = g.numpy()
    path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), "loss", out["total"])


def run_category_tables():
    """The dataset vocabularies the result converter needs (utils/categories_v2.py via utils/evaluate.py:9): id -> name lists of
    VidVRD (standard and the PKU proposals' numbering), VidVRD predicates, VidOR objects and predicates.  Data, written as JSON
    next to the converter (vidsgg-big_amd/bigc_hip/data/categories.json)."""
    import contextlib
    import io
    import json
    sys.path.insert(0, REFERENCE)
    with contextlib.redirect_stdout(io.StringIO()):
        from utils import categories_v2 as C
    sys.path.remove(REFERENCE)
    out = {}
    for k in ("vidvrd_CatId2name", "vidvrd_PredId2name", "vidor_CatId2name", "vidor_PredId2name", "PKU_vidvrd_CatId2name"):
        d = getattr(C, k)
        assert sorted(d) == list(range(len(d)))
        out[k] = [d[i] for i in range(len(d))]
    path = os.path.join(os.path.dirname(os.path.dirname(cases.GOLDEN_DIR)), "vidsgg-big_amd", "bigc_hip", "data", "categories.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, {k: len(v) for k, v in out.items()})


def run_evalfmt_golden():
    """EvalFmtCvtor.to_eval_format_pr (utils/evaluate.py:72-151, traj_cutoff utils/utils_func.py:523-536) of the real reference on
    the seeded cases of cases.make_evalfmt_case: VidVRD with the standard and the PKU category numbering, VidOR (video name
    reset), a None result, background predicates; plus the grounded variant (score x bin probability, spans x video_len, rounded:
    tools/eval_vidor.py:236-262 restated by the caller).  The relation dicts are stored as JSON."""
    import contextlib
    import io
    import json
    sys.path.insert(0, REFERENCE)
    with contextlib.redirect_stdout(io.StringIO()):
        from utils.evaluate import EvalFmtCvtor
    sys.path.remove(REFERENCE)
    out = {}
    for name, (dataset, use_pku, proposal, result) in cases.make_evalfmt_cases().items():
        cvt = EvalFmtCvtor(dataset)
        if result is not None:
            quint, scores, dura = result
            result = (quint, scores.mean(dim=-1) if scores.dim() == 2 else scores, dura)
        out[name] = cvt.to_eval_format_pr(proposal, result, use_pku=use_pku)
    path = os.path.join(cases.GOLDEN_DIR, "evalfmt.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), {k: [len(v) for v in d.values()] for k, d in out.items()})


def run_ties_golden(m10):
    """the reference's construct_triplet (models/model_0v10.py:707-785) on logits with EXACT probability ties (cases.make_ties_case)"""
    import types as _t
    p, logits, att, topk, _ = cases.make_ties_case()
    fake = _t.SimpleNamespace(topk=topk, num_anchors=logits.shape[0])
    res = m10.BIG_C.construct_triplet(fake, p, logits, att)
    out = {"torch_version": np.array(torch.__version__), "quintuples": res[0].numpy(), "scores": res[1].numpy(), "dura_inters": res[2].numpy(),
           "query_ids": res[3].numpy(), "topk_classes": torch.topk(torch.softmax(logits, -1), topk, -1)[1].numpy()}
    path = os.path.join(cases.GOLDEN_DIR, "bigc_ties.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, out["quintuples"].shape, out["topk_classes"][4:8].tolist())


def main():
    torch.manual_seed(0)
    m10, m07, grd = import_reference()
    names = sys.argv[1:] or (list(cases.BIGC_CASES) + list(cases.GRD_CASES) + list(cases.TRAIN_CASES) + list(cases.GRD_TRAIN_CASES) + ["evaluator", "categories", "evalfmt", "ties", "cache"] +
                             list(cases.INGEST_CASES) + list(cases.VIDOR_INGEST_CASES) + list(cases.BASEC_CASES) + list(cases.BASEC_TRAIN_CASES))
    for n in names:
        if n == "evaluator":
            run_evaluator_golden()
            continue
        if n == "ties":
            run_ties_golden(m10)
            continue
        if n == "categories":
            run_category_tables()
            continue
        if n == "evalfmt":
            run_evalfmt_golden()
            continue
        if n == "cache":
            run_cache_fixture()
            continue
        if n in cases.INGEST_CASES:
            run_ingest_case(n)
            continue
        if n in cases.VIDOR_INGEST_CASES:
            run_vidor_ingest_case(n)
            continue
        if n in cases.BASEC_CASES:
            run_basec_case(n)
            continue
        if n in cases.BASEC_TRAIN_CASES:
            run_basec_train_case(n)
            continue
        if n in cases.BIGC_CASES:
            run_bigc_case(n, m10, m07)
        elif n in cases.GRD_CASES:
            run_grd_case(n, grd)
        elif n in cases.TRAIN_CASES:
            run_train_case(n, m10, m07)
        elif n in cases.GRD_TRAIN_CASES:
            run_grd_train_case(n, grd)
        else:
            raise SystemExit("unknown case " + n)


if __name__ == "__main__":
    main()

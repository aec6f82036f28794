"""Seeded weights and synthetic tracklet proposals for parity tests and the benchmark.

Everything here is generated from ``numpy.random.RandomState`` so that the GPU box
regenerates bit-identical weights and inputs without the reference being present
(SURVEY.md section 8c/8d).  The synthetic-video recipe is the one SURVEY.md section 8(d)
prescribes: per video v (seed ``base_seed + v``) N tracklets, ``L_i = randint(8,64)``,
``start_i = randint(0,64)``, boxes ``xy~U(0,600)``, ``wh~U(10,110)`` (xyxy), frame
1280x720, features ``~N(0,1)``, ``cat_ids = randint(1,num_enti_cats)``, scores ``U(0,1)``
sorted descending (the order ``dataloaders/dataloader_vidvrd.py:40-46`` produces).
"""
from collections import OrderedDict

import numpy as np
import torch


class _ViewList(list):
    """The per-tracklet views of a block-backed ``TrajProposal``, made on demand.  Reading is a plain list; ASSIGNING an element
    (``proposal.features_list[i] = x``, what the reference's own ``.to()`` does, dataloaders/dataloader_vidvrd.py:73-75) must not be lost:
    it turns the container into one that owns explicit lists (the blocks stop being the source of truth) and writes there."""

    def __init__(self, owner, which, items):
        super().__init__(items)
        self._owner, self._which = owner, which

    def __setitem__(self, i, value):
        self._owner._own_lists()[self._which][i] = value
        super().__setitem__(i, value)


class TrajProposal(object):
    """Duck-typed stand-in for the reference's per-video tracklet container.

    Attribute names, dtypes and the closed-interval duration convention follow
    ``dataloaders/dataloader_vidvrd.py:14-77`` (``TrajProposal``): ``num_proposals``,
    ``video_len``, ``video_wh``, ``traj_durations`` (int64 (N,2), closed),
    ``bboxes_list`` (N x f32 (L_i,4) xyxy pixels), ``features_list`` (N x f32 (L_i,D)),
    ``cat_ids`` (int64 (N,)), ``scores`` (f32 (N,)), ``video_name``; ``to(device)`` moves
    the tensors in place and returns ``self``.
    """

    def __init__(self, video_name, video_len, video_wh, cat_ids, scores, traj_durations,
                 bboxes_list, features_list):
        self.video_name = video_name
        self.video_len = int(video_len)
        self.video_wh = tuple(video_wh)
        self.num_proposals = len(bboxes_list)
        if self.num_proposals == 0:
            return
        self.cat_ids = torch.as_tensor(cat_ids, dtype=torch.long)
        self.scores = torch.as_tensor(scores, dtype=torch.float32)
        self.traj_durations = torch.as_tensor(traj_durations, dtype=torch.long)
        self.bboxes_list = [torch.as_tensor(b, dtype=torch.float32) for b in bboxes_list]
        self.features_list = [torch.as_tensor(f, dtype=torch.float32) for f in features_list]
        self.dim_feat = self.features_list[0].shape[1]
        for i in range(self.num_proposals):
            n = int(self.traj_durations[i, 1] - self.traj_durations[i, 0] + 1)
            assert self.bboxes_list[i].shape == (n, 4)
            assert self.features_list[i].shape[0] == n

    @classmethod
    def from_blocks(cls, video_name, video_len, video_wh, cat_ids, scores, traj_durations, lens, box_block, feat_block):
        """The same container with the tracklets as views of ONE (sum L, 4) box tensor and ONE (sum L, D) feature tensor per
        video (what ``bigc_hip.ingest`` assembles and what ``.to()`` produces): the batch packer then gets every tracklet's
        address by arithmetic instead of visiting the tensors."""
        p = cls.__new__(cls)
        p.video_name, p.video_len, p.video_wh = video_name, int(video_len), tuple(video_wh)
        p.num_proposals = len(lens)
        if p.num_proposals == 0:
            return p
        p.cat_ids = torch.as_tensor(cat_ids, dtype=torch.long)
        p.scores = torch.as_tensor(scores, dtype=torch.float32)
        p.traj_durations = torch.as_tensor(traj_durations, dtype=torch.long)
        p._set_blocks(box_block, feat_block, [int(l) for l in lens])
        p.dim_feat = int(feat_block.shape[1])
        dev = feat_block.device
        p.cat_ids, p.scores, p.traj_durations = p.cat_ids.to(dev), p.scores.to(dev), p.traj_durations.to(dev)
        return p

    def _set_blocks(self, box_block, feat_block, lens):
        import numpy as np
        assert box_block.dtype == torch.float32 and feat_block.dtype == torch.float32 and box_block.is_contiguous() and feat_block.is_contiguous()
        assert box_block.shape == (sum(lens), 4) and feat_block.shape[0] == sum(lens)
        # The per-tracklet tensors are views of the blocks, made when somebody asks for the lists (the properties below) and not kept:
        # 256 resident videos would otherwise keep 33 000 tensor objects alive, and every full pass of Python's cyclic garbage collector
        # over them stalls the calling thread for ~70 ms (measured: one model(proposal_list) call in 40 took 100 instead of 25 ms).
        self._lists = None
        self._box_block, self._feat_block, self._lens = box_block, feat_block, [int(l) for l in lens]
        ln = np.asarray(lens, dtype=np.int64)
        off = np.zeros(len(lens), dtype=np.uint64)
        np.cumsum(ln[:-1], out=off[1:])
        width = int(feat_block.shape[1])
        fptr = np.uint64(feat_block.data_ptr()) + off * np.uint64(4 * width)
        bptr = np.uint64(box_block.data_ptr()) + off * np.uint64(16)
        # the source table the batch packer reads (bigc_hip.packed.tracklet_table): same fields, no per-tensor visit
        self._bigc_pack = dict(n=len(lens), device=feat_block.device, width=width, lens=ln, fptr=fptr, bptr=bptr, fld=None,
                               aligned=bool(feat_block.data_ptr() % 16 == 0 and box_block.data_ptr() % 16 == 0 and width % 4 == 0),
                               blocks=(feat_block.data_ptr(), box_block.data_ptr()), keep=[])

    @property
    def features_list(self):
        if self._lists is not None:
            return self._lists[1]
        return _ViewList(self, 1, torch.split(self._feat_block, self._lens, 0))

    @features_list.setter
    def features_list(self, value):
        self._own_lists()[1] = value

    @property
    def bboxes_list(self):
        if self._lists is not None:
            return self._lists[0]
        return _ViewList(self, 0, torch.split(self._box_block, self._lens, 0))

    @bboxes_list.setter
    def bboxes_list(self, value):
        self._own_lists()[0] = value

    def _own_lists(self):
        """explicit lists (assigned by the caller) replace the blocks as the source of truth"""
        if getattr(self, "_lists", None) is None:
            blocks = getattr(self, "_feat_block", None)
            self._lists = [None, None] if blocks is None else [list(torch.split(self._box_block, self._lens, 0)), list(torch.split(blocks, self._lens, 0))]
            self._feat_block = self._box_block = None
            if hasattr(self, "_bigc_pack"):
                del self._bigc_pack
        return self._lists

    def to(self, device):
        """As the reference's ``TrajProposal.to`` (dataloaders/dataloader_vidvrd.py:66-77), which copies tensor by tensor; here the
        boxes and the features of a video travel as ONE block each and the per-tracklet tensors become views of the blocks."""
        if self.num_proposals == 0:
            return self
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.cat_ids = self.cat_ids.to(device)
        self.scores = self.scores.to(device)
        self.traj_durations = self.traj_durations.to(device)
        if self._lists is None:
            if self._feat_block.device == device:
                return self
            lens = self._lens
            bb, fb = self._box_block.to(device), self._feat_block.to(device)
        else:
            bl, fl = self._lists
            lens = [int(b.shape[0]) for b in bl]
            bb = torch.cat([b.to(torch.float32) for b in bl], 0).to(device)
            fb = torch.cat([f.to(torch.float32) for f in fl], 0).to(device)
        self._set_blocks(bb.contiguous(), fb.contiguous(), lens)
        return self

    def __repr__(self):
        return "TrajProposal[{},num_proposals={}]".format(self.video_name, self.num_proposals)


def synthetic_proposal(seed, n_trajs=64, feat_width=2048 + 832, num_enti_cats=36,
                       len_range=(8, 64), start_range=(0, 64), video_wh=(1280, 720),
                       name=None, feat_scale=1.0):
    """One synthetic video following SURVEY.md section 8(d)."""
    rs = np.random.RandomState(seed)
    if n_trajs == 0:
        return TrajProposal(name or "synth_%d" % seed, 0, video_wh, [], [], [], [], [])
    lens = rs.randint(len_range[0], len_range[1], size=n_trajs)
    starts = rs.randint(start_range[0], start_range[1], size=n_trajs)
    video_len = int((starts + lens).max())
    duras = np.stack([starts, starts + lens - 1], axis=1).astype(np.int64)
    cat_ids = rs.randint(1, num_enti_cats, size=n_trajs).astype(np.int64)
    scores = np.sort(rs.uniform(0, 1, size=n_trajs).astype(np.float32))[::-1].copy()
    bboxes, feats = [], []
    for i in range(n_trajs):
        xy = rs.uniform(0, 600, size=(lens[i], 2))
        wh = rs.uniform(10, 110, size=(lens[i], 2))
        bboxes.append(np.concatenate([xy, xy + wh], axis=1).astype(np.float32))
        feats.append((feat_scale * rs.standard_normal((lens[i], feat_width))).astype(np.float32))
    return TrajProposal(name or "synth_%d" % seed, video_len, video_wh, cat_ids, scores, duras,
                        bboxes, feats)


# --------------------------------------------------------------------------------------
# seeded weights
# --------------------------------------------------------------------------------------

def _mha_shapes(prefix, d):
    return [
        (prefix + ".in_proj_weight", (3 * d, d)),
        (prefix + ".in_proj_bias", (3 * d,)),
        (prefix + ".out_proj.weight", (d, d)),
        (prefix + ".out_proj.bias", (d,)),
    ]


def _linear_shapes(prefix, n_out, n_in, bias=True):
    out = [(prefix + ".weight", (n_out, n_in))]
    if bias:
        out.append((prefix + ".bias", (n_out,)))
    return out


def _ln_shapes(prefix, d):
    return [(prefix + ".weight", (d,)), (prefix + ".bias", (d,))]


def bigc_state_shapes(cfg, variant="vidvrd"):
    """Ordered (key, shape) list of the BIG-C ``state_dict``.

    Key names and shapes are the reference's (``models/model_0v10.py:266-334`` for
    ``variant="vidvrd"``, ``models/model_0v7.py:265-341`` for ``variant="vidor"``; dump in
    SURVEY.md section 8b).
    """
    de, dp, da, df = cfg["dim_enti"], cfg["dim_pred"], cfg["dim_att"], cfg["dim_ffn"]
    q = cfg["num_querys"]
    ne, npred = cfg["num_enti_cats"], cfg["num_pred_cats"]
    shapes = []
    if variant == "vidvrd" or cfg.get("EntiNameEmb_path", None) is not None or cfg.get("_with_EntiNameEmb", False):
        shapes.append(("EntiNameEmb", (ne, cfg["dim_clsme"])))
    shapes += [("pos_embedding", (q, dp)), ("pred_query_init", (q, dp)),
               ("bias_matrix", (ne, ne, npred))]
    shapes += _linear_shapes("fc_feat2enti.0", de, cfg["dim_feat"])
    shapes += _linear_shapes("fc_feat2enti.2", de, de)
    shapes += _linear_shapes("fc_bbox2enti.0", de, 8)
    shapes += _linear_shapes("fc_bbox2enti.2", de, de)
    shapes += [("conv_feat2enti.weight", (de, 2 * de, 3)), ("conv_feat2enti.bias", (de,))]
    shapes += _linear_shapes("fc_enti2enco.0", de, de * cfg["enco_pool_len"])
    shapes += _linear_shapes("fc_enti2enco.2", de, de)
    if variant == "vidvrd" and cfg.get("dim_i3d", None):
        shapes += _linear_shapes("fc_i3d.0", de, cfg["dim_i3d"])
    for i in range(cfg["n_enco_layers"]):
        p = "encoder_layers.%d" % i
        shapes += _mha_shapes(p + ".self_attn", de)
        shapes += _linear_shapes(p + ".linear1", df, de)
        shapes += _linear_shapes(p + ".linear2", de, df)
        shapes += _ln_shapes(p + ".norm1", de) + _ln_shapes(p + ".norm2", de)
    for i in range(cfg["n_deco_layers"]):
        p = "decoder_layers.%d" % i
        shapes += _mha_shapes(p + ".self_attn", dp)
        for r in range(2):
            shapes += _linear_shapes(p + ".fc_rolewise.%d.0" % r, dp, de)
            shapes += _linear_shapes(p + ".fc_rolewise.%d.2" % r, dp, dp)
        shapes += _linear_shapes(p + ".fc_enti2att", da, de)
        shapes += _linear_shapes(p + ".fc_pred2att", da, dp)
        shapes += _linear_shapes(p + ".fc2.0", df, dp)
        shapes += _linear_shapes(p + ".fc2.3", dp, df)
        shapes += _ln_shapes(p + ".norm1", dp) + _ln_shapes(p + ".norm2", dp) + _ln_shapes(p + ".norm3", dp)
    if variant == "vidvrd":
        if cfg.get("dim_i3d", None):
            k = dp + 2 * cfg["dim_clsme"] + 4 * de
        else:
            k = dp + 2 * cfg["dim_clsme"] + 2 * de
        shapes += _linear_shapes("fc_pred2logits", npred, k)
    else:
        k = dp + 2 * de + (2 * cfg["dim_clsme"] if cfg["use_clsme"] else 0)
        shapes += _linear_shapes("fc_pred2logits.0", df, k)
        shapes += _linear_shapes("fc_pred2logits.2", npred, df)
    return shapes


def basec_state_shapes(cfg):
    """Ordered (key, shape) list of the pairwise baseline's ``state_dict`` (models/model_pairwise_baseline.py:28-71)."""
    de, df, ne, npred = cfg["dim_enti"], cfg["dim_ffn"], cfg["num_enti_cats"], cfg["num_pred_cats"]
    shapes = []
    if cfg.get("EntiNameEmb_path", None) is not None or cfg.get("_with_EntiNameEmb", False):
        shapes.append(("EntiNameEmb", (ne, cfg["dim_clsme"])))
    shapes.append(("bias_matrix", (ne, ne, npred)))
    shapes += _linear_shapes("fc_feat2enti.0", de, cfg["dim_feat"]) + _linear_shapes("fc_feat2enti.2", de, de)
    shapes += _linear_shapes("fc_bbox2enti.0", de, 8) + _linear_shapes("fc_bbox2enti.2", de, de)
    shapes += [("conv_feat2enti.weight", (de, 2 * de, 3)), ("conv_feat2enti.bias", (de,))]
    shapes += _linear_shapes("fc_enti2enco.0", de, de * cfg["enco_pool_len"]) + _linear_shapes("fc_enti2enco.2", de, de)
    k = 2 * de + (2 * cfg["dim_clsme"] if cfg["use_clsme"] else 0)
    shapes += _linear_shapes("fc_pred2logits.0", df, k) + _linear_shapes("fc_pred2logits.2", npred, df)
    return shapes


def grd_state_shapes(cfg):
    """Ordered (key, shape) list of the grounding scorer's ``state_dict``
    (reference: models/grd_model_v5.py:158-193; the two embedding tables are hard-wired to
    81 entity / 51 predicate rows there, :162,168)."""
    d, k, de = cfg["dim_hidden"], cfg["num_bins"], cfg["dim_clsme"]
    shapes = [("EntiNameEmb", (81, de)), ("PredNameEmb", (51, de))]
    shapes += _linear_shapes("video_fc", d, cfg["dim_feat"])
    shapes += _linear_shapes("query_fc", d, de)
    shapes += _linear_shapes("temp_fc", d, 2)
    shapes += _linear_shapes("vq_fc", d, 4 * d)

    def dws(prefix, c_out, ksize):
        return [(prefix + ".depth_wise.weight", (d, 1, ksize)), (prefix + ".depth_wise.bias", (d,)),
                (prefix + ".point_wise.weight", (c_out, d, 1)), (prefix + ".point_wise.bias", (c_out,))]

    for enc, ksize in (("video_encoder", 7), ("query_encoder", 3), ("combined_encoder", 7)):
        for i in range(4):
            shapes += dws("%s.convs.%d" % (enc, i), d, ksize)
        shapes += _mha_shapes(enc + ".mh_attn", d)
        shapes += _linear_shapes(enc + ".fc", d, d)
        shapes += _ln_shapes(enc + ".normb", d)
        for i in range(4):
            shapes += _ln_shapes("%s.norm_seq.%d" % (enc, i), d)
        shapes += _ln_shapes(enc + ".norme", d)
    shapes.append(("proj2sim.weight", (d, d)))
    for hname, c_out in (("cls_head", k), ("conf_head", k), ("regr_head", 2 * k)):
        for i in range(4):
            shapes += dws("%s.%d.0" % (hname, i), d, 3)
        shapes += dws("%s.4" % hname, c_out, 3)
    return shapes


def _fill(rs, key, shape, att_gain=1.0):
    """Seeded value for one state_dict entry (SURVEY.md section 8d weight recipe)."""
    leaf = key.split(".")[-1]
    if key in ("pos_embedding", "pred_query_init"):
        return rs.normal(0.0, 0.1, size=shape)
    if key == "bias_matrix":
        return rs.uniform(-7.0, 0.0, size=shape)
    if key in ("EntiNameEmb", "PredNameEmb"):
        return rs.normal(0.0, 0.4, size=shape)
    if ".norm" in key:
        if leaf == "weight":
            return 1.0 + rs.normal(0.0, 0.05, size=shape)
        return rs.normal(0.0, 0.02, size=shape)
    if len(shape) == 1:
        return rs.normal(0.0, 0.02, size=shape)
    recept = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_out, fan_in = shape[0] * recept, shape[1] * recept
    std = np.sqrt(2.0 / (fan_in + fan_out))
    if "depth_wise.weight" in key:
        # the reference uses kaiming-normal (2/fan_in, models/grd_model_v5.py:47); 0.5/fan_in keeps the
        # five un-normalised conv stacks of the heads from saturating every sigmoid
        std = np.sqrt(0.5 / recept)
    if "fc_enti2att" in key or "fc_pred2att" in key or "_head.4.point_wise.weight" in key or key == "proj2sim.weight":
        std *= att_gain
    return rs.normal(0.0, std, size=shape)


def seeded_state_dict(shapes, seed, att_gain=1.0):
    """numpy-seeded fill of a list of (key, shape): xavier-scaled normal matrices,
    small biases, LayerNorm weights near one, query/pos embeddings N(0,0.1),
    ``bias_matrix`` in [-7,0] (the range of log(p+1e-3))."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for key, shape in shapes:
        sd[key] = torch.from_numpy(np.asarray(_fill(rs, key, shape, att_gain), dtype=np.float32))
    return sd


def sine_pos_emb(length, d_model):
    """Fixed sine table used as ``pos_embedding`` by the VidOR model.

    Same values as the reference's ``SinePosEmb`` (models/model_0v7.py:229-238):
    entry (p, i) = sin(p * f_i + ph_i) evaluated in float32, with
    f_i = 10000^(-i/d) (i even) or -(10000^((1-i)/d)) (i odd) and ph_i = 0 / pi/2.
    """
    idx = np.arange(d_model)
    even = (idx % 2) == 0
    f = np.where(even, 10000.0 ** (-idx / d_model), -(10000.0 ** ((1 - idx) / d_model)))
    ph = np.where(even, 0.0, np.pi / 2)
    f32 = torch.from_numpy(f.astype(np.float32))
    ph32 = torch.from_numpy(ph.astype(np.float32))
    pos = torch.arange(length, dtype=torch.float32)
    return torch.sin(pos[:, None] * f32[None, :] + ph32[None, :])


# --------------------------------------------------------------------------------------
# synthetic ground truth for the training step
# --------------------------------------------------------------------------------------

class VideoGraph(object):
    """Duck-typed stand-in for the reference's GT container (dataloaders/dataloader_vidvrd.py:84-140):
    ``traj_cat_ids`` i64 (n_traj,), ``traj_durations`` i64 (n_traj,2) closed, ``traj_bboxes`` list of f32
    (frames,4), ``pred_cat_ids`` i64 (n_pred,), ``pred_durations`` f32 (n_pred,2) closed, ``adj_matrix`` f32
    (2,n_pred,n_traj), ``num_trajs``, ``num_preds``, ``video_len``, ``max_frames``; ``to(device)`` in place."""

    def __init__(self, video_name, video_len, video_wh, traj_cat_ids, traj_durations, traj_bboxes, pred_cat_ids,
                 pred_durations, adj_matrix):
        self.video_name, self.video_len, self.video_wh = video_name, int(video_len), tuple(video_wh)
        self.traj_cat_ids = torch.as_tensor(traj_cat_ids, dtype=torch.long)
        self.traj_durations = torch.as_tensor(traj_durations, dtype=torch.long)
        self.traj_bboxes = [torch.as_tensor(b, dtype=torch.float32) for b in traj_bboxes]
        self.pred_cat_ids = torch.as_tensor(pred_cat_ids, dtype=torch.long)
        self.pred_durations = torch.as_tensor(pred_durations, dtype=torch.float32)
        self.adj_matrix = torch.as_tensor(adj_matrix, dtype=torch.float32)
        self.num_trajs, self.num_preds = len(self.traj_bboxes), int(self.pred_cat_ids.shape[0])
        self.n_frames_list = [int(b.shape[0]) for b in self.traj_bboxes]
        self.max_frames = max(self.n_frames_list)

    def to(self, device):
        self.traj_cat_ids = self.traj_cat_ids.to(device)
        self.pred_cat_ids = self.pred_cat_ids.to(device)
        self.traj_durations = self.traj_durations.to(device)
        self.pred_durations = self.pred_durations.to(device)
        self.adj_matrix = self.adj_matrix.to(device)
        self.traj_bboxes = [b.to(device) for b in self.traj_bboxes]
        return self


def synthetic_gt_graph(proposal, seed, n_gt_trajs=6, n_preds=5, num_pred_cats=133, jitter=6.0):
    """GT for one synthetic video: ``n_gt_trajs`` of the proposals (boxes jittered by a few pixels so that
    vIoU with the source proposal is high but not 1, the last one shifted far away so that it matches no
    proposal above the threshold and takes the best-proposal fallback of models/model_0v10.py:587-590) and
    ``n_preds`` predicates between temporally overlapping pairs of them."""
    rs = np.random.RandomState(seed)
    n = proposal.num_proposals
    picks = rs.choice(n, size=min(n_gt_trajs, n), replace=False)
    duras = proposal.traj_durations[picks].clone()
    boxes = []
    for j, p in enumerate(picks):
        b = proposal.bboxes_list[int(p)].clone() + torch.from_numpy(rs.uniform(-jitter, jitter, size=(1, 4)).astype(np.float32))
        if j == len(picks) - 1:
            b = b + 400.0
        boxes.append(b)
    cats = proposal.cat_ids[picks].clone()
    pairs = [(a, b) for a in range(len(picks)) for b in range(len(picks)) if a != b and
             max(int(duras[a, 0]), int(duras[b, 0])) <= min(int(duras[a, 1]), int(duras[b, 1]))]
    assert pairs, "no overlapping GT pair: choose another seed"
    sel = [pairs[i] for i in rs.choice(len(pairs), size=min(n_preds, len(pairs)), replace=False)]
    adj = np.zeros((2, len(sel), len(picks)), dtype=np.float32)
    pd = []
    for k, (a, b) in enumerate(sel):
        adj[0, k, a] = 1.0
        adj[1, k, b] = 1.0
        pd.append([max(int(duras[a, 0]), int(duras[b, 0])), min(int(duras[a, 1]), int(duras[b, 1]))])
    pred_cats = rs.randint(1, num_pred_cats, size=len(sel))
    return VideoGraph(proposal.video_name, proposal.video_len, proposal.video_wh, cats, duras, boxes, pred_cats, pd, adj)


# --------------------------------------------------------------------------------------
# a LEARNABLE synthetic relation set (Recall@50 that is not zero)
# --------------------------------------------------------------------------------------

def relational_video(seed, n_trajs=64, feat_width=2048 + 832, num_enti_cats=36, num_pred_cats=133, n_actors=4, marker=3.0,
                     marker_width=128, name=None, label_noise=0.0, motion=False):
    """One synthetic video (SURVEY.md section 8d recipe) whose ground truth a model can LEARN, so that Recall@50 on held-out
    videos is far from zero after a few hundred training steps (with section 8d's random predicates every recall is 0 and a
    recall-parity gate compares 0 with 0).

    ``n_actors`` tracklets that share a frame are the "actors": actor k carries a marker, ``+marker`` on feature columns
    ``[k * marker_width, (k + 1) * marker_width)`` of every one of its frames.  The ground truth is the ``n_actors *
    (n_actors - 1)`` ordered actor pairs (12 relations, as section 8d), predicate of (actor a, actor b) = ``1 + a * n_actors + b``,
    GT trajectories = the actors' own proposal boxes (vIoU = 1 when the model picks the pair), GT span = the overlap span.

    Harder variants, so that a trained model's Recall@50 lands well inside (0, 1) and a change of arithmetic can move it:
    ``motion``: every actor's boxes become a coherent left-to-right or right-to-left walk (a random direction bit per actor) and the
    predicate also encodes whether subject and object walk the same way -- ``1 + a * n_actors + b + n_actors^2 * (dir_a != dir_b)`` --
    which the model can only read off the BOX features (fc_bbox2enti, the 8 relative coordinates of models/model_0v10.py:401-420);
    ``label_noise``: each ground-truth predicate is replaced by a random other class with this probability (irreducible error: a
    perfect model's recall is about ``1 - label_noise``).
    -> (TrajProposal, VideoGraph)"""
    p = synthetic_proposal(seed, n_trajs=n_trajs, feat_width=feat_width, num_enti_cats=num_enti_cats, name=name)
    rs = np.random.RandomState(seed + 500009)
    duras = p.traj_durations.numpy()
    assert n_actors * marker_width <= feat_width and 1 + n_actors * n_actors * (2 if motion else 1) <= num_pred_cats
    # the frame most tracklets cover: all tracklets that cover it overlap pairwise
    cover = np.zeros(p.video_len + 1, dtype=np.int64)
    for s, e in duras:
        cover[s:e + 1] += 1
    t_star = int(np.argmax(cover))
    cand = np.flatnonzero((duras[:, 0] <= t_star) & (duras[:, 1] >= t_star))
    assert len(cand) >= n_actors, "fewer than %d tracklets share a frame: choose another seed" % n_actors
    actors = np.sort(rs.choice(cand, size=n_actors, replace=False))
    rs.shuffle(actors)                                         # actor slot k is not tied to the score order
    for k, t in enumerate(actors):
        p.features_list[int(t)][:, k * marker_width:(k + 1) * marker_width] += marker
    dirs = rs.randint(0, 2, size=n_actors) if motion else np.zeros(n_actors, dtype=np.int64)
    if motion:
        for k, t in enumerate(actors):
            b = p.bboxes_list[int(t)]
            L = b.shape[0]
            x0 = float(rs.uniform(50, 250)) if dirs[k] == 0 else float(rs.uniform(850, 1050))
            step = (1.0 if dirs[k] == 0 else -1.0) * float(rs.uniform(6, 12))
            y0, w, h = float(rs.uniform(100, 500)), float(rs.uniform(40, 100)), float(rs.uniform(40, 100))
            xs = x0 + step * torch.arange(L, dtype=torch.float32)
            b[:, 0], b[:, 1], b[:, 2], b[:, 3] = xs, y0, xs + w, y0 + h
    pairs = [(a, b) for a in range(n_actors) for b in range(n_actors) if a != b]
    adj = np.zeros((2, len(pairs), n_actors), dtype=np.float32)
    spans, preds = [], []
    for j, (a, b) in enumerate(pairs):
        adj[0, j, a] = 1.0
        adj[1, j, b] = 1.0
        da, db = duras[actors[a]], duras[actors[b]]
        spans.append([max(int(da[0]), int(db[0])), min(int(da[1]), int(db[1]))])
        cls = 1 + a * n_actors + b + (n_actors * n_actors if (motion and dirs[a] != dirs[b]) else 0)
        if label_noise > 0 and rs.uniform() < label_noise:
            cls = 1 + (cls - 1 + int(rs.randint(1, num_pred_cats - 1))) % (num_pred_cats - 1)       # any OTHER class in [1, num_pred_cats)
        preds.append(cls)
    g = VideoGraph(p.video_name, p.video_len, p.video_wh, p.cat_ids[actors].clone(), p.traj_durations[actors].clone(),
                   [p.bboxes_list[int(t)].clone() for t in actors], np.asarray(preds, dtype=np.int64), spans, adj)
    return p, g

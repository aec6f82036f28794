#!/usr/bin/env python3
"""Which stage's float32 round-off is behind the outlier videos of the benchmark model (CPU only; uses oracle/ as the checker).
The oracle's op sequence evaluated in float64 EXCEPT for one named group of stages, which runs in float32 (inputs rounded to f32,
f32 arithmetic, result widened again) -- and the converse, everything in float32 except one group in float64.  Logit error is
max |logit - logit64| / max |logit64| per video.
    python benchmarks/accuracy_mixed.py 28 29 172"""
import copy
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT):
    sys.path.insert(0, p)
from bigc_hip import configs  # noqa: E402
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal  # noqa: E402
from oracle import bigc_oracle as O  # noqa: E402

GROUPS = ["frontend", "encoder", "self_attn", "ln", "att_proj", "att_scores", "att_values", "role_mlp", "ffn", "head"]


class P(object):
    """precision policy: group -> dtype"""

    def __init__(self, default, special=(), special_dtype=None):
        self.default, self.special, self.special_dtype = default, set(special), special_dtype

    def dt(self, group):
        return self.special_dtype if group in self.special else self.default


def run(fn, dtype, *tensors):
    """fn on tensors cast to dtype; result widened to f64"""
    out = fn(*[t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t for t in tensors])
    if isinstance(out, tuple):
        return tuple(o.double() for o in out)
    return out.double()


def sdcast(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def forward(sd64, cfg, p64, pol):
    sds = {torch.float32: sdcast(sd64, torch.float32), torch.float64: sd64}
    de, nhead = cfg["dim_enti"], cfg["n_att_head"]

    def lin(x, prefix, group, relu=False):
        dt = pol.dt(group)
        return O.linear(x.to(dt), sds[dt], prefix, relu).double()

    def ln(x, prefix):
        dt = pol.dt("ln")
        return O.layer_norm(x.to(dt), sds[dt], prefix).double()

    def mha(xq, xk, xv, prefix, group):
        dt = pol.dt(group)
        return O.mha(xq.to(dt), xk.to(dt), xv.to(dt), sds[dt], prefix, nhead).double()

    dt = pol.dt("frontend")
    tb, tf = O.preprocess_proposal(p64)
    tb, tf = tb.to(dt), tf.to(dt)
    s = sds[dt]
    visual, extra = tf[:, :, :cfg["dim_feat"]], tf[:, :, cfg["dim_feat"]:]
    fb = O.linear(O.linear(tb, s, "fc_bbox2enti.0", True), s, "fc_bbox2enti.2", True)
    fv = O.linear(O.linear(visual, s, "fc_feat2enti.0", True), s, "fc_feat2enti.2", True)
    enti = torch.cat([fb, fv], -1).permute(0, 2, 1)
    conv = F.conv1d(enti, s["conv_feat2enti.weight"], s["conv_feat2enti.bias"], stride=2, padding=1)
    pooled = F.adaptive_max_pool1d(conv, cfg["enco_pool_len"]).reshape(tb.shape[0], -1)
    e2e = O.linear(O.linear(pooled, s, "fc_enti2enco.0", True), s, "fc_enti2enco.2", True).double()
    i3d_avg = extra.mean(dim=1).double()

    x = e2e
    for i in range(cfg["n_enco_layers"]):
        pfx = "encoder_layers.%d" % i
        dte = pol.dt("encoder")
        x = O.encoder_layer(x.to(dte), sds[dte], pfx, nhead).double()
    enco = x

    query, pos = sd64["pred_query_init"], sd64["pos_embedding"]
    half = cfg["dim_att"] // 2
    att = None
    for i in range(cfg["n_deco_layers"]):
        pfx = "decoder_layers.%d" % i
        qk = query + pos
        query = ln(query + mha(qk, qk, query, pfx + ".self_attn", "self_attn"), pfx + ".norm1")
        query = query + pos
        e2a = lin(enco, pfx + ".fc_enti2att", "att_proj")
        p2a = lin(query, pfx + ".fc_pred2att", "att_proj")
        dts = pol.dt("att_scores")
        pa, ea = p2a.to(dts), e2a.to(dts)
        scores = torch.stack([pa[:, :half] @ ea[:, :half].t(), pa[:, half:] @ ea[:, half:].t()], 0) / np.sqrt(de)
        att = (torch.softmax(scores, dim=2) * torch.softmax(scores, dim=0)).double()
        dtv = pol.dt("att_values")
        role = 0
        for r in range(2):
            vals = (att[r].to(dtv) @ enco.to(dtv)).double()
            fr = pfx + ".fc_rolewise.%d" % r
            role = role + lin(lin(vals, fr + ".0", "role_mlp", True), fr + ".2", "role_mlp")
        query = ln(query + role, pfx + ".norm2")
        y = lin(lin(query, pfx + ".fc2.0", "ffn", True), pfx + ".fc2.3", "ffn")
        query = ln(query + y, pfx + ".norm3")

    soid = torch.argmax(att, dim=-1)
    socat = p64.cat_ids[soid]
    dth = pol.dt("head")
    s = sds[dth]
    bias = s["bias_matrix"][socat[0], socat[1], :]
    sub_feat, obj_feat = e2e[soid[0]].to(dth), e2e[soid[1]].to(dth)
    sub_c, obj_c = s["EntiNameEmb"][socat[0]], s["EntiNameEmb"][socat[1]]
    sub_i = O.linear(i3d_avg[soid[0]].to(dth), s, "fc_i3d.0", True)
    obj_i = O.linear(i3d_avg[soid[1]].to(dth), s, "fc_i3d.0", True)
    head_in = torch.cat([query.to(dth), sub_i, obj_i, sub_feat, obj_feat, sub_c, obj_c], -1)
    return (O.linear(head_in, s, "fc_pred2logits") + bias).double(), query


def survey(n):
    """worst / mean logit error over the first n benchmark videos for a few precision policies"""
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    torch.set_num_threads(8)
    pols = {"all f32": P(torch.float32),
            "f32, att_proj+att_scores f64": P(torch.float32, ["att_proj", "att_scores"], torch.float64),
            "f32, frontend+encoder f64": P(torch.float32, ["frontend", "encoder"], torch.float64),
            "f32, frontend+encoder+att_proj+att_scores f64": P(torch.float32, ["frontend", "encoder", "att_proj", "att_scores"], torch.float64),
            "f32, att_proj f64": P(torch.float32, ["att_proj"], torch.float64)}
    errs = {k: [] for k in pols}
    for v in range(n):
        p = synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
        p64 = copy.deepcopy(p)
        p64.bboxes_list = [b.double() for b in p64.bboxes_list]
        p64.features_list = [f.double() for f in p64.features_list]
        with torch.no_grad():
            ref, _ = forward(sd64, cfg, p64, P(torch.float64))
            for k, pol in pols.items():
                out, _ = forward(sd64, cfg, p64, pol)
                errs[k].append(float((out - ref).abs().max() / ref.abs().max()))
        print("video %d: " % v + "  ".join("%.1e" % errs[k][-1] for k in pols), flush=True)
    for k in pols:
        a = np.array(errs[k])
        print("%-50s worst %.2e  mean %.2e  over 1e-4: %d of %d" % (k, a.max(), a.mean(), int((a > 1e-4).sum()), n))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--survey":
        return survey(int(sys.argv[2]))
    vids = [int(a) for a in sys.argv[1:]] or [28, 29, 172]
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    torch.set_num_threads(8)
    for v in vids:
        p = synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
        p64 = copy.deepcopy(p)
        p64.bboxes_list = [b.double() for b in p64.bboxes_list]
        p64.features_list = [f.double() for f in p64.features_list]
        with torch.no_grad():
            ref, qref = forward(sd64, cfg, p64, P(torch.float64))
            e = lambda x: float((x - ref).abs().max() / ref.abs().max())
            all32, _ = forward(sd64, cfg, p64, P(torch.float32))
            print("video %d: everything f32 %.2e" % (v, e(all32)), flush=True)
            for g in GROUPS:
                only32, _ = forward(sd64, cfg, p64, P(torch.float64, [g], torch.float32))
                but64, _ = forward(sd64, cfg, p64, P(torch.float32, [g], torch.float64))
                print("   %-11s only this group in f32: %.2e    everything f32 but this group in f64: %.2e" % (g, e(only32), e(but64)), flush=True)


if __name__ == "__main__":
    main()

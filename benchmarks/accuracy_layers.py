#!/usr/bin/env python3
"""Which decoder layer an outlier video's logit error comes from.  For each video, the decoder query after layer k (k = 1..n) of
  * the HIP path (models truncated to k decoder layers sharing the full model's weights: pred_query of the k-layer model IS the
    layer-k query of the full one),
  * the float32 CPU oracle,
against a float64 evaluation of the oracle; and, per layer, the INTRINSIC error of the HIP layer: a one-layer model whose
pred_query_init is the float64 oracle's layer-(k-1) query (rounded to f32) and whose decoder layer is layer k -- what the layer adds
when its query input is exact.  Error = max |x - x64| / max |x64|.
    python benchmarks/accuracy_layers.py 28 172 0 1        (GPU box; uses oracle/ as the checker)"""
import copy
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT):
    sys.path.insert(0, p)
from bigc_hip import configs  # noqa: E402
from bigc_hip.packed import PackedBatch  # noqa: E402
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal  # noqa: E402
from oracle import bigc_oracle as O  # noqa: E402


def build(cfg, sd):
    from models import BIG_C_vidvrd
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy()); np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
    model = BIG_C_vidvrd(dict(cfg, bias_matrix_path=d + "/b.npy", EntiNameEmb_path=d + "/e.npy"), is_train=False)
    model.load_state_dict(sd)
    return model.to("cuda:0")


def truncated(cfg, sd, k):
    c = dict(cfg, n_deco_layers=k)
    keep = {key: v for key, v in sd.items() if not key.startswith("decoder_layers.") or int(key.split(".")[1]) < k}
    return c, keep


def one_layer(cfg, sd, k, query0):
    """decoder layer k alone, started from `query0` instead of pred_query_init"""
    c = dict(cfg, n_deco_layers=1)
    keep = {key: v for key, v in sd.items() if not key.startswith("decoder_layers.")}
    for key, v in sd.items():
        if key.startswith("decoder_layers.%d." % k):
            keep["decoder_layers.0." + key.split(".", 2)[2]] = v
    keep["pred_query_init"] = query0
    return c, keep


def err(x, ref):
    return float((x.double().cpu().reshape(ref.shape) - ref).abs().max() / ref.abs().max())


def main():
    vids = [int(a) for a in sys.argv[1:]] or [28, 172, 0]
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    n = cfg["n_deco_layers"]
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    torch.set_num_threads(16)
    models = {k: build(*truncated(cfg, sd, k)) for k in range(1, n + 1)}
    for v in vids:
        p = synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
        with torch.no_grad():
            o32 = O.encode2decode(sd, cfg, p, "vidvrd", keep=True)
            p64 = copy.deepcopy(p)
            p64.bboxes_list = [b.double() for b in p64.bboxes_list]
            p64.features_list = [f.double() for f in p64.features_list]
            p64.scores = p64.scores.double()
            o64 = O.encode2decode(sd64, cfg, p64, "vidvrd", keep=True)
        packed = PackedBatch([copy.deepcopy(p).to("cuda:0")], "cuda:0")
        print("video %d: enti2enco hip %.1e cpu32 %.1e | enco_output hip %.1e cpu32 %.1e" % (
            v, err(models[n].forward_packed(packed, 10, keep_intermediates=True)["enti2enco"], o64["enti2enco"]), err(o32["enti2enco"], o64["enti2enco"]),
            err(models[n].forward_packed(packed, 10, keep_intermediates=True)["enco_output"], o64["enco_output"]), err(o32["enco_output"], o64["enco_output"])), flush=True)
        for k in range(1, n + 1):
            out = models[k].forward_packed(packed, 10, keep_intermediates=True)
            q64 = o64["dec_states"][k - 1][0]
            a64 = o64["dec_states"][k - 1][1]
            e_hip, e_cpu = err(out["pred_query"], q64), err(o32["dec_states"][k - 1][0], q64)
            # intrinsic: layer k alone from the float64 query of layer k-1
            q_prev = (sd64["pred_query_init"] if k == 1 else o64["dec_states"][k - 2][0]).float()
            m1 = build(*one_layer(cfg, sd, k - 1, q_prev))
            out1 = m1.forward_packed(packed, 10, keep_intermediates=True)
            e_int = err(out1["pred_query"], q64)
            # the same for the float32 oracle
            with torch.no_grad():
                enco32 = o32["enco_output"]
                q1, _ = O.decoder_layer(q_prev, sd["pos_embedding"], enco32, sd, "decoder_layers.%d" % (k - 1), cfg["n_att_head"], cfg["dim_enti"], cfg["dim_att"])
            e_int32 = err(q1, q64)
            # how sharp / how tied the float64 attention is: smallest margin between the two largest scores-softmax entries
            top2 = torch.topk(a64, 2, dim=-1).values
            margin = float((top2[..., 0] - top2[..., 1]).min())
            print("  layer %d: query hip %.1e cpu32 %.1e | intrinsic (exact query in) hip %.1e cpu32 %.1e | min att margin %.2e" % (k, e_hip, e_cpu, e_int, e_int32, margin), flush=True)
            del m1
        print("  logits: hip %.1e cpu32 %.1e" % (err(models[n].forward_packed(packed, 10)["pred_logits"], o64["pred_logits"]), err(o32["pred_logits"], o64["pred_logits"])), flush=True)


if __name__ == "__main__":
    main()

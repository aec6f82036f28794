import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import cases
from bigc_hip import _lib
from bigc_hip.packed import PackedBatch
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal
from test_hip_bigc import build_model
DEV = "cuda:0"
lib = _lib.load()
cfg = cases.get_cfg("CONFIG_TINY")
cfg = dict(cfg, dim_enti=64, dim_pred=64, dim_att=64, dim_ffn=64, dim_feat=128, dim_i3d=32, n_att_head=4, num_querys=24)
sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 91, att_gain=3.0)
props = [synthetic_proposal(4400 + v, n_trajs=20, feat_width=160, num_enti_cats=cfg["num_enti_cats"], len_range=(6, 40), start_range=(0, 20)).to(DEV) for v in range(6)]
packed = PackedBatch(props, DEV)
keys = ("pred_logits", "quint", "scores", "n_out")
def run(model):
    out = model.forward_packed(packed, 3, keep_intermediates=True)
    model.counts(out)
    torch.cuda.synchronize()
    return {k: out[k].clone() for k in keys + ("enti2enco", "enco_output", "pred_query")}
def diff(a, b):
    return {k: float((a[k].float() - b[k].float()).abs().max()) for k in a}
base = build_model(cfg, "vidvrd", sd)
r = [run(base) for _ in range(4)]
print("base default, 4 runs vs first:", [diff(x, r[0]) for x in r[1:]])
m2 = build_model(cfg, "vidvrd", sd)
print("second model vs base:", diff(run(m2), r[0]))
print("second model again:", diff(run(m2), r[0]))
m2.engine().set_mode(fast=1)
f1 = run(m2)
prev = lib.bigc_linear_set_fast(1); f2 = run(base); lib.bigc_linear_set_fast(prev)
print("handle-fast vs global-fast:", diff(f1, f2), " fast vs default:", diff(f1, r[0]))
print("base default again:", diff(run(base), r[0]))

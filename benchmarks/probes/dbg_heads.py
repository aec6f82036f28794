import os, sys, gc
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, "vidsgg-big_amd"); sys.path.insert(0, ".")
import torch, numpy as np
import cases
from test_hip_grounding import build_model, DEV
seq = sys.argv[1].split(",")
cfg = dict(cases.GRD_CASES["grd_g_n64"]["cfg"])
_, sd0, _ = cases.make_grd_case("grd_g_n64")
feat, (quint, dura, vl) = cases.make_grd_inputs(cfg, dict(seed=31, T=70, n=9, video_len=555))
model = None
for case in seq:
    sd = {k: v.clone() for k, v in sd0.items()}
    if case == "ha":
        sd["conf_head.0.0.point_wise.weight"] *= 3e5
        sd["conf_head.4.point_wise.weight"] *= 1e-5
    elif case == "hw":
        sd["cls_head.2.0.point_wise.weight"][5, 7] = 5e4
        sd["cls_head.4.point_wise.weight"] *= 1e-4
    elif case == "nan":
        sd["conf_head.1.0.point_wise.weight"][3, 3] = float("nan")
    new = build_model(dict(cfg, loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0)), sd)
    torch.cuda.synchronize(); print(case, "built", flush=True)
    model = None; gc.collect(); torch.cuda.synchronize()
    print(case, "destroyed previous", flush=True)
    model = new
    got = model.forward_raw(feat.to(DEV), quint, dura, vl, keep_intermediates=True, **cases.GRD_TH)
    torch.cuda.synchronize(); print(case, "chain ok", flush=True)
    eng = model.engine(); eng.set_mode(1, -1)
    want = model.forward_raw(feat.to(DEV), quint, dura, vl, keep_intermediates=True, **cases.GRD_TH)
    eng.set_mode(-1, -1)
    torch.cuda.synchronize(); print(case, "exact ok", flush=True)
print("done", flush=True)

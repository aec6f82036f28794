import os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, "vidsgg-big_amd"); sys.path.insert(0, ".")
import cases
from test_hip_bigc import build_model, rel
from bigc_hip.packed import PackedBatch
for name in cases.BIGC_CASES:
    g = np.load(os.path.join(cases.GOLDEN_DIR, name + ".npz"))
    cfg, variant, sd, props, topk = cases.make_case(name)
    model = build_model(cfg, variant, sd)
    Q = cfg["num_querys"]
    for v, p in enumerate(props):
        pre = "v%d_" % v
        if p.num_proposals == 0 or pre + "att" not in g.files: continue
        out = model.forward_packed(PackedBatch([p], "cuda:0"), topk, keep_intermediates=True)
        n = p.num_proposals
        print(name, v, "att %.2e pred_query %.2e enco %.2e e2e %.2e logits %.2e" % (rel(out["att"].cpu().numpy().reshape(2, Q, n), g[pre + "att"]),
              rel(out["pred_query"].cpu().numpy(), g[pre + "pred_query"]), rel(out["enco_output"].cpu().numpy(), g[pre + "enco_output"]),
              rel(out["enti2enco"].cpu().numpy(), g[pre + "enti2enco"]), rel(out["pred_logits"].cpu().numpy(), g[pre + "pred_logits"])))

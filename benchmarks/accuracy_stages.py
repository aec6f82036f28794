#!/usr/bin/env python3
"""Where the f32 round-off of the eval forward comes from: error of each stage (enti2enco, enco_output, every decoder layer's
query, logits) against a float64 evaluation of the CPU oracle, for the HIP path (default split-operand mode and exact-f32 mode)
and for the float32 CPU oracle (= the reference's arithmetic).  Error = max |x - x64| / max |x64|.
    python benchmarks/accuracy_stages.py [n_videos]     (GPU box; uses oracle/ as the checker)"""
import copy
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT):
    sys.path.insert(0, p)
from bigc_hip import _lib, configs  # noqa: E402
from bigc_hip.packed import PackedBatch  # noqa: E402
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal  # noqa: E402
from oracle import bigc_oracle as O  # noqa: E402


def main():
    n_vid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    from models import BIG_C_vidvrd
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy()); np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
    model = BIG_C_vidvrd(dict(cfg, bias_matrix_path=d + "/b.npy", EntiNameEmb_path=d + "/e.npy"), is_train=False)
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    lib = _lib.load()
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    sd64 = {k: v.double() for k, v in sd.items()}
    torch.set_num_threads(16)
    worst = {}
    for v in range(n_vid):
        p = synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
        with torch.no_grad():
            o32 = O.encode2decode(sd, cfg, p, "vidvrd", keep=True)
            p64 = copy.deepcopy(p)
            p64.bboxes_list = [b.double() for b in p64.bboxes_list]
            p64.features_list = [f.double() for f in p64.features_list]
            p64.scores = p64.scores.double()
            o64 = O.encode2decode(sd64, cfg, p64, "vidvrd", keep=True)
        packed = PackedBatch([copy.deepcopy(p).to("cuda:0")], "cuda:0")
        line = []
        for name, mode in (("split", 0), ("exact", 1)):
            lib.bigc_linear_set_exact(mode)
            out = model.forward_packed(packed, 10, keep_intermediates=True)
            for st in ("enti2enco", "enco_output", "pred_query", "pred_logits"):
                ref = o64[st]
                e = float((out[st].double().cpu().reshape(ref.shape) - ref).abs().max() / ref.abs().max())
                worst[(name, st)] = max(worst.get((name, st), 0.0), e)
                line.append("%s/%s %.1e" % (name, st[:6], e))
            hip_vs_ref = float((out["pred_logits"].cpu().reshape(o32["pred_logits"].shape) - o32["pred_logits"]).abs().max() / o32["pred_logits"].abs().max())
            worst[(name, "vs_cpu32")] = max(worst.get((name, "vs_cpu32"), 0.0), hip_vs_ref)
            line.append("%s/vs_cpu32 %.1e" % (name, hip_vs_ref))
        lib.bigc_linear_set_exact(0)
        for st in ("enti2enco", "enco_output", "pred_query", "pred_logits"):
            e = float((o32[st].double() - o64[st]).abs().max() / o64[st].abs().max())
            worst[("cpu32", st)] = max(worst.get(("cpu32", st), 0.0), e)
            line.append("cpu32/%s %.1e" % (st[:6], e))
        print("video %d: " % v + "  ".join(line), flush=True)
    print("worst over %d videos:" % n_vid)
    for k in sorted(worst):
        print("  %-8s %-12s %.2e" % (k[0], k[1], worst[k]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""(A checker script, not a pytest module: `python tests/accuracy_vs_fp64.py [n_videos]` on a GPU box.)
Accuracy of the HIP path's predicate logits on the benchmark model (exp2 dims, seeded weights with sharp attention):
split-operand GEMM mode and exact-f32 GEMM mode against a float64 evaluation of the CPU oracle, with the float32 CPU oracle
(= the reference's arithmetic) alongside.  Error = max |x - x64| / max |x64| over all queries of a video."""
import copy
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # tests/ -> repo root
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
    props = [synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"]) for v in range(n_vid)]
    sd64 = {k: v.double() for k, v in sd.items()}
    torch.set_num_threads(16)
    rows = []
    for v, p in enumerate(props):
        with torch.no_grad():
            o32 = O.encode2decode(sd, cfg, p, "vidvrd")
            p64 = copy.deepcopy(p)
            p64.bboxes_list = [b.double() for b in p64.bboxes_list]
            p64.features_list = [f.double() for f in p64.features_list]
            p64.scores = p64.scores.double()
            o64 = O.encode2decode(sd64, cfg, p64, "vidvrd")
        ref = o64["pred_logits"]
        scale = float(ref.abs().max())
        packed = PackedBatch([copy.deepcopy(p).to("cuda:0")], "cuda:0")
        res = {}
        for mode in (0, 1):
            lib.bigc_linear_set_exact(mode)
            out = model.forward_packed(packed, 10)
            res[mode] = out["pred_logits"].double().cpu()
            soid = out["soid"][0].cpu().long()
            res["soid%d" % mode] = bool(torch.equal(soid, torch.argmax(o64["att"], dim=-1)))
        lib.bigc_linear_set_exact(0)
        e = lambda x: float((x.double() - ref).abs().max()) / scale
        rows.append((e(res[0]), e(res[1]), e(o32["pred_logits"]), res["soid0"], res["soid1"],
                     bool(torch.equal(torch.argmax(o32["att"], dim=-1), torch.argmax(o64["att"], dim=-1)))))
        print("video %d: split %.2e  exact-f32 %.2e  torch-cpu-f32 %.2e   argmax(att)==fp64: split %s exact %s cpu-f32 %s" % ((v,) + rows[-1]), flush=True)
    a = np.array([r[:3] for r in rows])
    print("max over %d videos: split %.2e  exact-f32 %.2e  torch-cpu-f32 %.2e" % (n_vid, a[:, 0].max(), a[:, 1].max(), a[:, 2].max()))


if __name__ == "__main__":
    main()

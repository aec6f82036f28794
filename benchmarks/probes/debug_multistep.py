import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch, numpy as np
import cases
from test_hip_train import build_train_model
from oracle import bigc_train_oracle as T
name = sys.argv[1]; lr = float(sys.argv[2])
cfg, variant, sd, props, _ = cases.make_train_case(name)
model = build_train_model(cfg, variant, sd)
opt = torch.optim.Adam(model.parameters(), lr=lr)
for step in range(3):
    gts = cases.make_train_case(name)[4]
    opt.zero_grad()
    loss, _ = model(props, gts)
    loss.backward()
    torch.cuda.synchronize()
    sdg = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    want, _, _ = T.forward_train(sdg, cfg, props, cases.make_train_case(name)[4], variant)
    want.backward()
    worst = {}
    for k, p in model.named_parameters():
        if p.grad is None: continue
        ref = sdg[k].grad
        worst[k] = (float((p.grad.cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-12)), float(ref.abs().max()), float(p.grad.abs().max()))
    top = sorted(worst.items(), key=lambda kv: -kv[1][0])[:6]
    print("step", step, "loss hip %.6f oracle %.6f" % (float(loss), float(want)), "median err %.2e" % np.median([v[0] for v in worst.values()]))
    for k, v in top: print("    %-45s err %.2e  |ref| %.2e |hip| %.2e" % ((k,) + v))
    zeros = [k for k, v in worst.items() if v[2] == 0.0]
    print("    zero hip grads: %d of %d; nonzero:" % (len(zeros), len(worst)), [k for k in worst if k not in zeros][:12])
    front = {k: v[0] for k, v in worst.items() if k.startswith(("fc_bbox2enti", "fc_feat2enti"))}
    print("    front-end:", {k: "%.1e" % v for k, v in front.items()})
    opt.step()

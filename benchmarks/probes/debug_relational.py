import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch, copy
from bigc_hip import configs
from bigc_hip.packed import PackedBatch
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, relational_video
from oracle import bigc_oracle as O
from models import BIG_C_vidvrd
dev = "cuda:0"
cfg = dict(configs.CONFIG_V)
width = cfg["dim_feat"] + cfg["dim_i3d"]
sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 11, att_gain=2.0)
d = tempfile.mkdtemp()
np.save(d + "/b.npy", sd["bias_matrix"].numpy()); np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
mcfg = dict(cfg, bias_matrix_path=d + "/b.npy", EntiNameEmb_path=d + "/e.npy")
model = BIG_C_vidvrd(mcfg, is_train=True); model.load_state_dict(sd); model = model.to(dev).eval()
ev = BIG_C_vidvrd(mcfg, is_train=False).to(dev)
mk = lambda s: relational_video(s, n_trajs=64, feat_width=width, marker=6.0)
train = [mk(20000 + v) for v in range(32)]
cpu0 = mk(20000)
for p, _ in train: p.to(dev)
duras = {id(g): g.traj_durations.clone() for _, g in train}
opt = torch.optim.Adam(model.parameters(), lr=3e-4, fused=True)
rs = np.random.RandomState(0)
for it in range(400):
    pick = rs.choice(len(train), size=16, replace=False)
    props, gts = [train[i][0] for i in pick], [train[i][1] for i in pick]
    for g in gts: g.traj_durations = duras[id(g)].clone()
    opt.zero_grad(); loss, terms = model(props, gts); loss.mean().backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5); opt.step()
print("final loss", float(loss), {k: float(v) for k, v in terms.items()})
p0, g0 = train[0]
g0.traj_durations = duras[id(g0)].clone()
loss, terms = model([p0], [g0])
iq, ig = model.last_match_indices[0]
print("video 0 alone: loss", float(loss), {k: float(v) for k, v in terms.items()})
print("matched queries", list(iq), "-> gt", list(ig))
ev.load_state_dict(model.state_dict())
out = ev.forward_packed(PackedBatch([p0], dev), 10, keep_intermediates=True)
soid = out["soid"][0].cpu().numpy()
logits = out["pred_logits"].cpu()
so_gt = g0.adj_matrix.argmax(-1).t().numpy()       # (n_pred, 2) actor slots
print("GT predicates", g0.pred_cat_ids.tolist())
sdt = {k: v.detach().cpu() for k, v in model.state_dict().items()}
o = O.encode2decode(sdt, cfg, cpu0[0])
so_or = o["att"].argmax(-1).numpy()
print("oracle vs hip eval: soid equal", bool((so_or == soid).all()), " logits rel diff %.2e" % float((o["pred_logits"] - logits).abs().max() / o["pred_logits"].abs().max()))
# which proposals are the actors?  match GT traj boxes to proposals
acts = []
for b in cpu0[1].traj_bboxes:
    acts.append([i for i, pb in enumerate(cpu0[0].bboxes_list) if pb.shape == b.shape and torch.equal(pb, b)][0])
print("actor proposals", acts)
for q, gi in zip(iq, ig):
    print("  query %3d gt %2d (actors %s pred %d): hip eval so=(%d,%d) argmax pred %d p=%.3f | oracle so=(%d,%d) pred %d" % (
        q, gi, [acts[so_gt[gi][0]], acts[so_gt[gi][1]]], int(g0.pred_cat_ids[gi]), soid[0][q], soid[1][q], int(logits[q].argmax()), float(torch.softmax(logits[q], -1).max()),
        so_or[0][q], so_or[1][q], int(o["pred_logits"][q].argmax())))
res = ev([p0], topk=10)[0]
print("eval relations:", None if res is None else res[0].shape, None if res is None else res[0][:12].tolist())
print("nan/inf in weights:", [k for k, v in sdt.items() if not torch.isfinite(v).all()])
print("largest |w|:", sorted(((float(v.abs().max()), k) for k, v in sdt.items() if v.is_floating_point()), reverse=True)[:6])
ok = O.encode2decode(sdt, cfg, cpu0[0], keep=True)
for name in ("enti2enco", "enco_output", "pred_query", "pred_logits", "att"):
    t = ok[name]
    print("oracle %-12s finite %s  absmax %.3e  mean %.3e" % (name, bool(torch.isfinite(t).all()), float(t.abs().max()), float(t.mean())))
for name in ("enti2enco", "enco_output", "pred_query", "pred_logits"):
    t = out[name].cpu()
    print("hip    %-12s finite %s  absmax %.3e  mean %.3e  rel diff to oracle %.2e" % (name, bool(torch.isfinite(t).all()), float(t.abs().max()), float(t.mean()),
          float((t.reshape(ok[name].shape) - ok[name]).abs().max() / ok[name].abs().max())))
from oracle import bigc_train_oracle as T
gcpu = cpu0[1]
want, terms_o, _ = T.forward_train({k: v.clone() for k, v in sdt.items()}, cfg, [cpu0[0]], [gcpu], "vidvrd")
print("oracle train forward: loss", float(want), terms_o)
att = ok["att"]
print("oracle att row sums (role 0, first 3 queries):", att[0, :3].sum(-1).tolist(), " max per query:", att[0, :3].max(-1).values.tolist())

#!/usr/bin/env python3
"""Per-video distance of the HIP predicate logits to a float64 evaluation of the oracle over the WHOLE benchmark set (256 videos of
BASELINE configs[1]), for several arithmetic variants of the GEMM kernels (each variant is a fresh process: the switches are read once).

    python benchmarks/accuracy_sweep.py --videos 256 --variants default,acc2_conv,acc2_all,blk4,exact     (GPU box; oracle/ is the checker)

Phase 1 (this process, CPU): the f32 oracle and the float64 oracle of every video -> /tmp/acc_sweep_ref.npz.
Phase 2 (one child per variant, GPU): model.forward_packed on the same videos, in-place feature rows as the timed call reads them ->
per-video max |logit - logit64| / max |logit64| on the queries whose attention arg-max agrees with the float64 one.
One JSON line per variant (max, mean, videos above 1e-4, the five worst videos, videos where HIP is farther from float64 than the
f32 oracle AND above 1e-4) on stdout and appended to gpurun_out/acc_sweep.jsonl."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT):
    sys.path.insert(0, p)

VARIANTS = {
    "default": {},
    "acc2_conv": {"BIGC_PS_ACC2": "conv"},
    "acc2_all": {"BIGC_PS_ACC2": "all"},
    "blk2": {"BIGC_PS_BLK": "2"},
    "blk4": {"BIGC_PS_BLK": "4"},
    "blk8": {"BIGC_PS_BLK": "8"},
    "exact": {"BIGC_LINEAR_EXACT": "1"},
    "feat0_splitloop": {"BIGC_FEAT0_SPLITLOOP": "1"},
    "blk4_k1024": {"BIGC_PS_BLK": "4", "BIGC_PS_BLK_MINK": "1024"},
    "blk4_front": {"BIGC_PS_BLK": "4", "BIGC_PS_BLK_MINM": "100000"},
    "blk4_small": {"BIGC_PS_BLK": "4", "BIGC_PS_BLK_MAXM": "100000"},
    "blk4_decoder": {"BIGC_PS_BLK": "4", "BIGC_PS_BLK_MINM": "20000", "BIGC_PS_BLK_MAXM": "100000"},
    "rb2": {"BIGC_PS_RB": "2"},
    "ln_fused": {"BIGC_LN_FUSED": "1"},
    "ln_fused_resident": {"BIGC_LN_FUSED": "1", "ACC_SWEEP_RESIDENT": "1"},
    "plain": {"BIGC_ACCUM": "plain"},
    "k1024": {"BIGC_ACCUM": "k1024"},
    "front": {"BIGC_ACCUM": "front"},
    "nodecoder": {"BIGC_ACCUM": "nodecoder"},
    "blocked": {"BIGC_ACCUM": "blocked"},
    "blocked_splitloop": {"BIGC_ACCUM": "blocked", "BIGC_FEAT0_SPLITLOOP": "1"},
    "resident": {"ACC_SWEEP_RESIDENT": "1"},
    "resident_blk4": {"ACC_SWEEP_RESIDENT": "1", "BIGC_PS_BLK": "4"},
}
REF = "/tmp/acc_sweep_ref.npz"


def usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def setup():
    import torch
    from bigc_hip import configs
    from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    return torch, cfg, sd, cfg["dim_feat"] + cfg["dim_i3d"]


def phase_ref(args):
    torch, cfg, sd, width = setup()
    from bigc_hip.synthetic import synthetic_proposal
    from oracle import accuracy as A
    from oracle import bigc_oracle as O
    torch.set_num_threads(usable_cores())
    sd64 = {k: v.double() for k, v in sd.items()}
    l32, l64, so32, so64 = [], [], [], []
    t0 = time.time()
    with torch.no_grad():
        for v in range(args.videos):
            p = synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
            o = O.encode2decode(sd, cfg, p)
            _, p64 = A.to_f64({}, p)
            o64 = O.encode2decode(sd64, cfg, p64)
            l32.append(o["pred_logits"].numpy()); so32.append(o["att"].argmax(-1).numpy())
            l64.append(o64["pred_logits"].numpy()); so64.append(o64["att"].argmax(-1).numpy())
    np.savez(REF, l32=np.stack(l32), l64=np.stack(l64), so32=np.stack(so32), so64=np.stack(so64))
    print("reference: %d videos, f32 + f64 oracle in %.1f s on %d threads" % (args.videos, time.time() - t0, torch.get_num_threads()), flush=True)


def summarise(name, err, err_ref, extra):
    err, err_ref = np.asarray(err), np.asarray(err_ref)
    worst = np.argsort(-err)[:5]
    return dict(variant=name, videos=int(len(err)), max=float(err.max()), mean=float(err.mean()), over_1e_4=int((err > 1e-4).sum()),
                over_1e_4_and_farther_than_f32_oracle=int(((err > 1e-4) & (err > err_ref)).sum()),
                worst=[dict(video=int(i), hip_vs_f64=float(err[i]), f32_oracle_vs_f64=float(err_ref[i])) for i in worst],
                f32_oracle=dict(max=float(err_ref.max()), mean=float(err_ref.mean()), over_1e_4=int((err_ref > 1e-4).sum())), **extra)


def phase_variant(args, name):
    torch, cfg, sd, width = setup()
    import tempfile
    from bigc_hip.packed import PackedBatch
    from bigc_hip.synthetic import synthetic_proposal
    from models import BIG_C_vidvrd
    ref = np.load(REF)
    dev = torch.device("cuda", 0)
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy()); np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
    model = BIG_C_vidvrd(dict(cfg, bias_matrix_path=d + "/b.npy", EntiNameEmb_path=d + "/e.npy"), is_train=False)
    model.load_state_dict(sd)
    model = model.to(dev)
    props = [synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"]).to(dev) for v in range(args.videos)]
    packed = PackedBatch(props, dev, "lazy")
    if os.environ.get("ACC_SWEEP_RESIDENT") == "1":
        packed.materialize("hr8", cfg["dim_feat"])
    out = model.forward_packed(packed, 10)
    model.counts(out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = model.forward_packed(packed, 10)
        model.counts(out)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 3
    Q, P = cfg["num_querys"], cfg["num_pred_cats"]
    lg = out["pred_logits"].cpu().numpy().reshape(args.videos, Q, P).astype(np.float64)
    so = out["soid"].cpu().numpy().reshape(args.videos, 2, Q)           # (B, 2, Q): attention arg-max (subject, object) per query
    err, err_ref, other = [], [], 0
    for v in range(args.videos):
        r64 = ref["l64"][v]
        same = (so[v] == ref["so64"][v]).all(0)
        same32 = (ref["so32"][v] == ref["so64"][v]).all(0)
        other += int((~same).sum())
        sc = np.abs(r64).max()
        err.append(float(np.abs(lg[v][same] - r64[same]).max() / sc) if same.any() else 0.0)
        err_ref.append(float(np.abs(ref["l32"][v][same32].astype(np.float64) - r64[same32]).max() / sc) if same32.any() else 0.0)
    line = summarise(name, err, err_ref, dict(ms_per_step=round(ms, 3), queries_with_other_argmax_than_f64=other, env=VARIANTS[name]))
    print(json.dumps(line), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "acc_sweep.jsonl"), "a") as f:
        f.write(json.dumps(dict(line, per_video=[float(e) for e in err])) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=256)
    ap.add_argument("--variants", default="default,acc2_conv,acc2_all,blk4,exact")
    ap.add_argument("--child", default="")
    args = ap.parse_args()
    if args.child:
        return phase_variant(args, args.child)
    if not os.path.exists(REF) or np.load(REF)["l64"].shape[0] != args.videos:
        phase_ref(args)
    for name in args.variants.split(","):
        env = dict(os.environ)
        env.update(VARIANTS[name])
        subprocess.run([sys.executable, os.path.abspath(__file__), "--videos", str(args.videos), "--child", name], env=env, check=False)


if __name__ == "__main__":
    main()

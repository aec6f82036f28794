"""BASELINE configs[4]: the BIG-C training step (`bench.py --mode train` and the `train` leg of the eval line)."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

from .common import (EXACT_F32, F16_MFMA_PEAK_TFLOPS, F32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT, PackedBatch, TrajProposal, _lib, bigc_state_shapes,
                     build_model, collect_families, configs, device_proposal, family_rooflines, gemm_roofline, init_dist, print_last, seeded_state_dict,
                     synthetic_proposal, timed_with_events, usable_cores)


def make_train_job(args, rank, device, nvid):
    from bigc_hip.synthetic import synthetic_gt_graph
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=3.0)
    model = build_model(cfg, sd, device, "vidvrd", is_train=True)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    props = [synthetic_proposal(5000 + rank * nvid + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
             for v in range(nvid)]
    gts = [synthetic_gt_graph(p, 6000 + rank * nvid + v, n_gt_trajs=8, n_preds=10, num_pred_cats=cfg["num_pred_cats"]) for v, p in enumerate(props)]
    props = [p.to(device) for p in props]            # GT stays on the host: alignment + Hungarian run there, as in the reference
    gt_duras = [g.traj_durations.clone() for g in gts]
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)     # the same update as the reference's Adam, one kernel
    pairs = sum(p.num_proposals * (p.num_proposals - 1) for p in props)
    return model, props, gts, gt_duras, opt, pairs


def train_step_fn(model, props, gts, gt_duras, opt, dist):
    from bigc_hip.shard import allreduce_gradients, clip_grad_norm_
    from bigc_hip.train_step import mark
    params = list(model.parameters())

    def step():
        for g, du in zip(gts, gt_duras):
            g.traj_durations = du.clone()            # the forward decrements them in place (reference quirk, model_0v10.py:567)
        opt.zero_grad()
        loss, terms = model(props, gts)
        loss.mean().backward()
        mark("autograd glue after the backward")
        flat = allreduce_gradients(params, dist)
        clip_grad_norm_(params, 5, flat=flat)
        opt.step()
        mark("all-reduce + clip + Adam launches")
        # the loss stays on the device: like the reference's loop (tools/train_vidvrd.py:160-164, which reads it every 10th
        # iteration) the step does not wait for the GPU, so the next step's packing overlaps this step's backward
        return loss.detach()
    return step


def train_main(args):
    """`--mode train`: the training step of BASELINE configs[4] (BIG-C DP training) on synthetic videos:
    zero_grad, HIP forward + matching (scipy's assignment algorithm as a device kernel), HIP backward, one flat-gradient all-reduce over RCCL (N>1),
    clip_grad_norm_(5), Adam -- the loop body of tools/train_vidvrd.py:151-164. Not the headline metric."""
    from bigc_hip.shard import allreduce_gradients, max_over_ranks
    rank, world, device, dist = init_dist(args)
    nvid = args.train_videos
    model, props, gts, gt_duras, opt, pairs = make_train_job(args, rank, device, nvid)
    step = train_step_fn(model, props, gts, gt_duras, opt, dist)
    for _ in range(args.warmup):
        step()
    lib = _lib.load()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed_this_rank = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_this_rank, dist, device)
    from bigc_hip.shard import flat_gradient_view, rank_report, time_allreduce
    ranks = rank_report(dist, device, elapsed_this_rank, args.steps)
    # the flat-gradient all-reduce alone (every rank takes part): microseconds, algorithmic and bus GB/s over the 111 MB buffer
    allreduce_alone = time_allreduce(flat_gradient_view([p for p in model.parameters() if p.grad is not None]), dist)
    families = {}
    if not args.no_prof:        # per-family kernel time from a few extra steps: the event pairs around every launch slow the host down
        n_prof = min(args.steps, 5)
        lib.bigc_prof_reset()
        lib.bigc_prof_enable(1)
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        lib.bigc_prof_enable(0)
        families = {k: round(v["ms"] / n_prof, 3) for k, v in collect_families(lib, n_prof).items()}
    line = None
    if rank == 0:
        line = json.dumps({"metric": "train_tracklet_pairs_per_sec", "value": round(pairs * world * args.steps / elapsed, 1), "unit": "pairs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "BIG-C model_0v10 training step (exp2 dims), %d synthetic videos x N=%d per GPU per step, "
                                                 "forward + Hungarian + backward + flat-gradient all-reduce + clip + Adam" % (nvid, args.n_trajs),
                                     "parallelism": "dp%d" % world}, "last_loss": round(float(last), 4),
                          "allreduce": {"backend": "nccl (RCCL)" if dist is not None else None,
                                        "flat_buffer_reduced_in_place": allreduce_gradients.last_in_place,
                                        "bytes": 4 * sum(p.numel() for p in model.parameters() if p.requires_grad),
                                        "alone": allreduce_alone},
                          "ranks": ranks, "kernel_families_ms_per_step": families})
    if dist is not None:
        dist.destroy_process_group()
    if line is not None:
        if world > 1:
            time.sleep(1.0)                # the other ranks (whose output the launcher merges into the same stream) have exited by then
        print_last(line)



def train_pmc_traffic(args, nvid):
    """HBM bytes per GEMM launch and per whole training step from the committed rocprofv3 --pmc passes of `bench.py --mode train
    --train-videos nvid` (profiles/pmc_traffic.json, section "train"; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes) -- only when they
    were measured on these kernel sources and this workload; else (None, reason)"""
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc_path) or args.n_trajs != 64:
        return None, "no committed counter pass for this workload"
    pmc = json.load(open(pmc_path))
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    from pmc_summary import csrc_sha16
    if pmc.get("csrc_sha16") != csrc_sha16():
        return None, "profiles/pmc_traffic.json was measured on other kernel sources (csrc fingerprint differs): not reported"
    t = (pmc.get("train") or {}).get("videos_%d" % nvid)
    if not t or "error" in t:
        return None, "no training pass in profiles/pmc_traffic.json"
    return t, "not measured in this run: profiles/pmc_traffic.json (%s), rocprofv3 --pmc passes of `bench.py --mode train --train-videos %d` on these kernel sources" % (pmc.get("tag", ""), nvid)


def train_leg(args, lib, device, budget=None):
    """BASELINE configs[4] on one GPU: the BIG-C training step at 8 and 64 videos per step (`budget()` true = skip what is left)"""
    out = {}
    for nvid in (8, 64):
        if budget is not None and budget() and out:
            out["skipped"] = "videos_%d (time budget)" % nvid
            break
        model, props, gts, gt_duras, opt, pairs = make_train_job(args, 0, device, nvid)
        step = train_step_fn(model, props, gts, gt_duras, opt, None)
        steps = 8
        el, el_prof, fam = timed_with_events(lib, step, 4, steps)
        repeats_ms = timed_with_events.last_repeats_ms
        kern_ms = sum(v["ms"] for v in fam.values()) / steps       # the library's kernels (HIP events); torch's Adam / clip kernels come on top
        # roofline of the step's GEMMs: forward + dX = dY W (family linear_f32) and dW = dY^T X (family linear_dw), both on the split-operand
        # f16 MFMAs (three per product): algorithmic flops / HIP-event time against the dense f16 peak / 3
        peak_f = F32_MFMA_PEAK_TFLOPS if EXACT_F32 else F16_MFMA_PEAK_TFLOPS / 3
        roof = gemm_roofline(fam, steps, 1e3 * el_prof / steps)
        dw = fam.get("linear_dw")
        if roof is not None and dw:
            lin = fam["linear_f32"]
            both_fl, both_ms = lin["flops"] + dw["flops"], lin["ms"] + dw["ms"]
            roof = dict(bound="mfma", achieved=round(both_fl / both_ms / 1e9, 2), peak=round(peak_f, 1), unit="TFLOP/s",
                        frac=round(both_fl / both_ms / 1e9 / peak_f, 4), kernel="forward + dX + dW GEMMs of the training step (split-operand f16 MFMAs)",
                        forward_and_dx=dict(tflops=roof["achieved"] if roof["unit"] == "TFLOP/s" else roof.get("tflops"), frac=round(lin["flops"] / lin["ms"] / 1e9 / peak_f, 4),
                                            launches_per_step=lin["launches"] // steps, gflop_per_step=round(lin["flops"] / steps / 1e9, 1),
                                            ms_per_step=round(lin["ms"] / steps, 3)),
                        dw=dict(tflops=round(dw["flops"] / dw["ms"] / 1e9, 2), frac=round(dw["flops"] / dw["ms"] / 1e9 / peak_f, 4),
                                launches_per_step=dw["launches"] // steps, gflop_per_step=round(dw["flops"] / steps / 1e9, 1), ms_per_step=round(dw["ms"] / steps, 3)),
                        share_of_step=round(both_ms / steps / (1e3 * el_prof / steps), 3), traffic=None,
                        measured="HIP events around each launch over %d instrumented steps" % steps)
            tpmc, tsrc = train_pmc_traffic(args, nvid)
            roof["traffic_source"] = tsrc
            if tpmc:
                roof["traffic"] = round(tpmc["gemm_hbm_bytes_per_launch"])
                roof["traffic_launches_per_step"] = tpmc["gemm_launches_per_step"]
                roof["algorithmic_bytes_per_launch"] = round((lin["bytes"] + dw["bytes"]) / max(1, lin["launches"] + dw["launches"]))
                roof["step_traffic_all_kernels"] = dict(hbm_bytes=tpmc["whole_step_fetch_bytes_all_kernels"] + tpmc["whole_step_write_bytes_all_kernels"])
        by_fam, share, gflop_step, _ = family_rooflines(fam, steps, 1e3 * el_prof / steps)
        out["videos_%d" % nvid] = dict(value=round(pairs * steps / el, 1), unit="pairs/s", ms_per_step=round(1e3 * el / steps, 3), steps=steps, ms_per_step_of_each_repeat=repeats_ms,
                                       kernel_ms_per_step=round(kern_ms, 3), kernel_share_of_step=round(kern_ms / (1e3 * el / steps), 3),
                                       instrumented_ms_per_step=round(1e3 * el_prof / steps, 3), roofline=roof,
                                       rooflines_by_family={k: v for k, v in by_fam.items() if "bound" in v}, roofline_share_of_step=share,
                                       kernel_families_ms_per_step={k: round(v["ms"] / steps, 3) for k, v in fam.items()})
        del model, props, gts, opt
        torch.cuda.empty_cache()
    out["workload"] = ("BIG-C model_0v10 training step (exp2 dims, dropout on), synthetic videos x N=%d: zero_grad + forward + matching on the device (GT alignment + scipy's assignment algorithm, no host round trip) + "
                       "backward + clip_grad_norm_ + Adam (tools/train_vidvrd.py:151-164)" % args.n_trajs)
    return out



def grd_train_leg(args, lib, device):
    """BASELINE configs[5], grounding stage: the training step of the temporal grounding scorer (models/grd_model_v5.py:224-306,375-527;
    tools/train_vidor.py:497-636) at the reference's batch of 8 videos (experiments/grounding_weights/config_.py:66), config G dims:
    zero_grad + forward (GT queries + drawn negatives) + the three losses + backward + Adam, on synthetic clip features and GT graphs."""
    import tempfile
    from models import DEBUG
    from bigc_hip.synthetic import grd_state_shapes, synthetic_gt_graph
    gcfg = dict(configs.CONFIG_G)
    gcfg["loss_factor"] = dict(classification=1.0, centerness=1.0, regression=1.0)
    gsd = seeded_state_dict(grd_state_shapes(gcfg), 12, 2.0)
    d = tempfile.mkdtemp()
    np.save(d + "/ge.npy", gsd["EntiNameEmb"].numpy())
    np.save(d + "/gp.npy", gsd["PredNameEmb"].numpy())
    model = DEBUG(dict(gcfg, EntiNameEmb_path=d + "/ge.npy", PredNameEmb_path=d + "/gp.npy"), is_train=True)
    model.load_state_dict(gsd)
    model = model.to(device)
    rs = np.random.RandomState(4711)
    feats, gts, n_pred = [], [], 0
    for v, L in enumerate([480, 800, 1100, 1400, 1800, 2200, 2700, 3200]):          # video lengths of the eval grounding leg: T = ceil(L / 8) in [60, 400]
        p = synthetic_proposal(9100 + v, n_trajs=16, feat_width=4, num_enti_cats=gcfg.get("num_enti_cats", 81), len_range=(L // 4, L // 2), start_range=(0, L // 2))
        g = synthetic_gt_graph(p, 9200 + v, n_gt_trajs=8, n_preds=10, num_pred_cats=gcfg.get("num_pred_cats", 51))
        T = -(-int(p.video_len) // 8)
        feats.append(torch.from_numpy((0.1 * rs.standard_normal((T, gcfg["dim_feat"]))).astype(np.float32)).to(device))
        gts.append(g)
        n_pred += int(g.num_preds)
    opt = torch.optim.Adam(model.parameters(), lr=5e-5, fused=True)

    def step():
        opt.zero_grad()
        torch.manual_seed(1000)                      # (the negative predicates are drawn from the global RNG, models/grd_model_v5.py:285-303)
        total, terms = model(feats, gts)
        total.backward()
        opt.step()
        return total.detach()

    steps = 8
    el, el_prof, fam = timed_with_events(lib, step, 4, steps)
    kern_ms = sum(v["ms"] for v in fam.values()) / steps
    out = dict(value=round(len(feats) * steps / el, 1), unit="videos/s", ms_per_step=round(1e3 * el / steps, 3), steps=steps,
               ms_per_step_of_each_repeat=timed_with_events.last_repeats_ms, videos=len(feats), gt_predicates=n_pred,
               clips=[int(f.shape[0]) for f in feats], kernel_ms_per_step=round(kern_ms, 3),
               kernel_families_ms_per_step={k: round(v["ms"] / steps, 3) for k, v in fam.items()},
               launches_per_step={k: v["launches"] // steps for k, v in fam.items()},
               workload="grounding scorer training step (BASELINE configs[5], grd_model_v5 DEBUG config G, batch 8 as experiments/grounding_weights/config_.py:66): "
                        "zero_grad + forward on the GT queries and drawn negatives + classification / centerness / regression losses + backward + Adam")
    del model, opt
    torch.cuda.empty_cache()
    return out

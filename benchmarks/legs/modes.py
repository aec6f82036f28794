"""The headline step in the other arithmetic modes: exact f32 MFMA, plain accumulation, reduced precision (f16 / bf16, Recall-gated)."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

from .common import (EXACT_F32, F16_MFMA_PEAK_TFLOPS, F32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT, PackedBatch, TrajProposal, _lib, bigc_state_shapes,
                     build_model, collect_families, configs, device_proposal, family_rooflines, gemm_roofline, init_dist, seeded_state_dict,
                     synthetic_proposal, timed_with_events, usable_cores)


def fast_mode_leg(args, lib, model, packed, step, ref_out, cfg, base, width, pairs_per_step, device):
    """The opt-in reduced-precision mode (bigc_linear_set_fast: every GEMM operand rounded to one f16, one MFMA per product; the
    "bf16 fast mode" of SURVEY.md section 7 step 10 / BASELINE configs[1]) on the same workload, outside the timed region.  It does
    not meet the 1e-4 logit tolerance by design: its gate is SURVEY.md 8(d)'s |R@50(fast) - R@50(parity)| <= 1e-3, evaluated by the
    `recall` leg on a TRAINED model (seeded random weights recall nothing, 0 = 0)."""
    from bigc_hip import evaluate as E
    lib.bigc_linear_set_fast(1)
    try:
        fa_out = model.forward_packed(packed, args.topk)          # also the warm-up of this mode
        model.counts(fa_out)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.fast_steps):
            step()
        torch.cuda.synchronize()
        fa_elapsed = time.perf_counter() - t0
    finally:
        lib.bigc_linear_set_fast(0)
    # the literal bf16 arithmetic of BASELINE configs[1] (operands rounded to bf16, v_mfma_f32_32x32x16_bf16; no GEMM -> GEMM chains)
    lib.bigc_linear_set_fast(2)
    try:
        bf_out = model.forward_packed(packed, args.topk)
        model.counts(bf_out)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.fast_steps):
            step()
        torch.cuda.synchronize()
        bf_elapsed = time.perf_counter() - t0
    finally:
        lib.bigc_linear_set_fast(0)
    bf_same = (ref_out["soid"] == bf_out["soid"]).all(1).reshape(-1)
    bf16 = dict(value=round(pairs_per_step * args.fast_steps / bf_elapsed, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * bf_elapsed / args.fast_steps, 3),
                arithmetic="GEMM operands rounded to one bf16 (8-bit mantissa), one v_mfma_f32_32x32x16_bf16 per product, f32 accumulate; every GEMM on the "
                           "f32-operand kernel (the split form that chains GEMMs carries f16 heads); opt-in (BIGC_FAST=bf16 / bigc_linear_set_fast(2))",
                max_logit_diff_rel_vs_parity_mode=float((((ref_out["pred_logits"] - bf_out["pred_logits"]).abs().max(1).values * bf_same).max()
                                                         / ref_out["pred_logits"].abs().max()).item()),
                queries_with_other_argmax_entity=int((~bf_same).sum()), gate="recall.bf16_gate_passed")
    same_so = (ref_out["soid"] == fa_out["soid"]).all(1).reshape(-1)
    logit_rel = float((((ref_out["pred_logits"] - fa_out["pred_logits"]).abs().max(1).values * same_so).max() / ref_out["pred_logits"].abs().max()).item())
    nv = min(32, args.videos)
    host = {k: {n: o[n][:nv].cpu() for n in ("quint", "scores", "dura", "n_out")} for k, o in (("parity", ref_out), ("fast", fa_out))}
    preds, common, total = {"parity": {}, "fast": {}}, 0, 0
    for v in range(nv):
        p = synthetic_proposal(base + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
        p.video_name = "synth_%d" % v
        top = {}
        for k, h in host.items():
            n = int(h["n_out"][v])
            res = (h["quint"][v, :n].long(), h["scores"][v, :n], h["dura"][v, :n].long()) if n else None
            preds[k].update(E.to_eval_format(p, res))
            # the 50 best-scored triplets of the video (what Recall@50 looks at), as (quintuple, span) keys
            if n:
                order = torch.argsort(h["scores"][v, :n].mean(-1), descending=True, stable=True)[:50]
                top[k] = {tuple(h["quint"][v, i].tolist()) + tuple(h["dura"][v, i].tolist()) for i in order.tolist()}
            else:
                top[k] = set()
        common += len(top["parity"] & top["fast"])
        total += len(top["parity"])
    # a sensitivity figure on the benchmark model itself (seeded random weights find no relation of a random-predicate ground truth):
    # GT relations taken FROM the parity mode's output -- 12 of each video's 100 best-scored relations -- so that the parity
    # mode's R@50 is about one half and every relation the fast mode ranks differently shows.  The GATE is the `recall` leg's.
    rs = np.random.RandomState(77)
    gts2 = {}
    for name, rel in preds["parity"].items():
        best = sorted(rel, key=lambda x: -x["score"])[:100]
        pick = rs.choice(len(best), size=min(12, len(best)), replace=False) if best else []
        gts2[name] = [{k: best[i][k] for k in ("triplet", "duration", "sub_traj", "obj_traj")} for i in sorted(pick)]
    _, r2_p, _ = E.evaluate(gts2, preds["parity"])
    _, r2_f, _ = E.evaluate(gts2, preds["fast"])
    return dict(value=round(pairs_per_step * args.fast_steps / fa_elapsed, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * fa_elapsed / args.fast_steps, 3),
                steps=args.fast_steps, arithmetic="GEMM operands rounded to one f16 (11-bit mantissa), one v_mfma_f32_32x32x16_f16 per product, f32 accumulate; "
                                                  "attention kernels unchanged; opt-in (BIGC_FAST=1 / bigc_linear_set_fast), never the headline",
                max_logit_diff_rel_vs_parity_mode=logit_rel, queries_with_other_argmax_entity=int((~same_so).sum()),
                top50_triplets_in_common_on_the_benchmark_model="%d of %d" % (common, total), bf16=bf16,
                recall_on_model_derived_gt=dict(gt_relations=sum(len(x) for x in gts2.values()), recall50_parity=r2_p[50], recall50_fast=r2_f[50],
                                                recall100_parity=r2_p[100], recall100_fast=r2_f[100], abs_diff_recall50=abs(r2_p[50] - r2_f[50]),
                                                note="GT = 12 of the parity mode's 100 best relations per video: a sensitivity figure; the gate is in `recall`"))



def _rows_vs(ref_out, other):
    """(valid-row mask on the padded result arrays, rows of `other` identical to `ref_out`'s, row counts equal)"""
    n_a, n_b = ref_out["n_out"].cpu(), other["n_out"].cpu()
    cap = ref_out["quint"].shape[1]
    valid = (torch.arange(cap)[None, :] < torch.minimum(n_a, n_b)[:, None]).to(ref_out["quint"].device)
    same = ((ref_out["quint"] == other["quint"]).all(-1) & (ref_out["dura"] == other["dura"]).all(-1)) & valid
    return valid, same, bool(torch.equal(n_a, n_b)), int(n_b.sum())


def exact_mode_leg(args, lib, model, packed, packed_step, ref_out, pairs_per_step):
    """the same step with every GEMM on the exact f32 MFMA kernel (bigc_linear_set_exact), outside the timed region"""
    lib.bigc_linear_set_exact(1)
    try:
        ex_out = model.forward_packed(packed, args.topk)          # also the warm-up of this mode
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.exact_steps):
            packed_step()
        torch.cuda.synchronize()
        ex_elapsed = time.perf_counter() - t0
    finally:
        lib.bigc_linear_set_exact(0)
    valid, same_rows, n_equal, n_rows = _rows_vs(ref_out, ex_out)
    same = n_equal and int(same_rows.sum()) == int(valid.sum())
    return dict(value=round(pairs_per_step * args.exact_steps / ex_elapsed, 1), unit="pairs/s (this GPU)",
                ms_per_step=round(1e3 * ex_elapsed / args.exact_steps, 3), steps=args.exact_steps,
                triplet_indices_identical_to_default_mode=bool(same),
                identical_triplet_rows="%d of %d" % (int(same_rows.sum()), n_rows),
                # a query whose attention arg-max (subject / object entity) is a tie at f32 resolution may pick the
                # other entity in the other mode and then gathers different features: counted, and left out of the
                # logit comparison
                queries_with_other_argmax_entity=int((ref_out["soid"] != ex_out["soid"]).any(1).sum()),
                max_logit_diff_rel_between_modes=float((((ref_out["pred_logits"] - ex_out["pred_logits"]).abs().max(1).values
                                                         * (ref_out["soid"] == ex_out["soid"]).all(1).reshape(-1)).max()
                                                        / ex_out["pred_logits"].abs().max()).item()))


def plain_accumulation_leg(args, lib, model, make_batch, step, ref_out, pairs_per_step, shape):
    """the same call with PLAIN accumulation (bigc_set_accum_policy(0): one f32 accumulator per element on the largest tiles -- round 1's default):
    what the blocked accumulation of the default costs, and what it buys (its distance to float64 is filled in by the oracle_parity loop)
    -> (leg, logits of this policy as (videos, Q, P) numpy)"""
    prev_policy = lib.bigc_set_accum_policy(0)
    try:
        pl_out = model.forward_packed(make_batch(), args.topk)     # also the warm-up of this policy
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.exact_steps):
            step()
        torch.cuda.synchronize()
        pl_elapsed = time.perf_counter() - t0
    finally:
        lib.bigc_set_accum_policy(prev_policy)
    valid, same_rows_pl, _, n_rows = _rows_vs(ref_out, pl_out)
    plain_logits = pl_out["pred_logits"].cpu().numpy().reshape(*shape)
    leg = dict(value=round(pairs_per_step * args.exact_steps / pl_elapsed, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * pl_elapsed / args.exact_steps, 3),
               steps=args.exact_steps, policy="0 = plain: one f32 accumulator per element, 256- / 192-row tiles, fc_feat2enti.0 in place on the DMA kernel",
               identical_triplet_rows_vs_default="%d of %d" % (int(same_rows_pl.sum()), n_rows),
               queries_with_other_argmax_entity=int((ref_out["soid"] != pl_out["soid"]).any(1).sum()),
               max_logit_diff_rel_vs_default=float(((ref_out["pred_logits"] - pl_out["pred_logits"]).abs().max() / ref_out["pred_logits"].abs().max()).item()))
    return leg, plain_logits

"""BASELINE configs[2] / [3]: VidOR-sized classification and the grounding scorer on its triplet queries."""
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


def vidor_grounding_legs(args, lib, device, want_vidor, want_grounding):
    """BASELINE configs[3]: VidOR-sized classification (model_0v7 at the exp5 dims, N=180 tracklets, L in [15,1000),
    topk=3, SURVEY.md 8d "config 4") and configs[2]/[3] second stage: the grounding scorer (config G) on the triplet
    queries the classification stage produced for the same videos (rt_triplets_topk = 200 per video)."""
    from models import DEBUG
    from bigc_hip.synthetic import grd_state_shapes
    cfg = dict(configs.CONFIG_O)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidor"), 11, att_gain=8.0)
    big = build_model(cfg, sd, device, "vidor")
    width = cfg["dim_feat"] + cfg["dim_clsme"]
    nv, ntr, max_len = args.vidor_videos, 180, 1000
    # the first `--vidor-oracle-videos` videos are generated on the HOST (numpy-seeded, SURVEY.md 8d recipe) so that the CPU oracle can
    # score the very same inputs; the others draw their (large) feature block on the device
    n_host = min(args.vidor_oracle_videos, nv) if want_vidor else 0
    host_props = [synthetic_proposal(4321 + v, n_trajs=ntr, feat_width=width, num_enti_cats=cfg["num_enti_cats"], len_range=(15, max_len),
                                     start_range=(0, max_len)) for v in range(n_host)]
    import copy
    props = [copy.deepcopy(p).to(device) for p in host_props] + \
            [device_proposal(4321 + v, ntr, width, cfg["num_enti_cats"], (15, max_len), (0, max_len), device) for v in range(n_host, nv)]
    video_len = [p.video_len for p in props]
    packed = PackedBatch(props, device)
    pairs, rows = packed.n_pairs(), packed.n_rows
    props, packed.proposals = None, None
    out_holder = {}

    def cls_step():
        out_holder["o"] = big.forward_packed(packed, 3)
        big.counts(out_holder["o"])

    steps = 3
    el, el_prof, fam = timed_with_events(lib, cls_step, 1, steps)
    legs = {}
    if want_vidor:
        legs["vidor"] = dict(value=round(pairs * steps / el, 1), unit="pairs/s", ms_per_step=round(1e3 * el / steps, 3),
                             ms_per_video=round(1e3 * el / steps / nv, 3), steps=steps,
                             workload="VidOR classification stage (BASELINE configs[3]): BIG-C model_0v7 eval forward incl. construct_triplet, "
                                      "exp5 dims (RoI 1024 + classeme 300, 6 enc / 4 dec), %d synthetic videos x N=%d, L in [15,%d), topk=3, "
                                      "%d rows, %d pairs per step" % (nv, ntr, max_len, rows, pairs),
                             roofline=gemm_roofline(fam, steps, 1e3 * el_prof / steps),
                             kernel_families_ms_per_step={k: round(v["ms"] / steps, 3) for k, v in fam.items()})
    if want_vidor and n_host:
        # the host-seeded videos through the CPU oracle (models/model_0v7.py:434-513 restated), outside the timing: rows identical, logits 1e-4
        from oracle import bigc_oracle as O
        Q, P = cfg["num_querys"], cfg["num_pred_cats"]
        o = out_holder["o"]
        hip_logits = o["pred_logits"][:n_host * Q].cpu().numpy().reshape(n_host, Q, P)
        hip_res = big.unpack_results(nv, list(range(nv)), o)[:n_host]
        torch.set_num_threads(usable_cores())
        rows_same, rows_all, max_rel, t0 = 0, 0, 0.0, time.perf_counter()
        with torch.no_grad():
            for v, p in enumerate(host_props):
                oo = O.encode2decode(sd, cfg, p, "vidor")
                want = O.construct_triplet(p, oo["pred_logits"], oo["att"], 3)
                got = hip_res[v]
                n_w = 0 if want is None else int(want[0].shape[0])
                rows_all += n_w
                if want is not None and got is not None and got[0].shape[0] == n_w:
                    rows_same += int(((got[0].cpu() == want[0]).all(1) & (got[2].cpu() == want[2]).all(1) & (got[3].cpu() == want[3])).sum())
                lg = oo["pred_logits"].numpy()
                max_rel = max(max_rel, float(np.abs(hip_logits[v] - lg).max() / np.abs(lg).max()))
        legs["vidor"]["oracle_parity"] = dict(videos=n_host, identical_triplet_rows="%d of %d" % (rows_same, rows_all), max_logit_rel=max_rel,
                                              oracle_seconds=round(time.perf_counter() - t0, 1),
                                              note="host-seeded videos of this leg (N = 180, L in [15, 1000)) through oracle/bigc_oracle.py, variant vidor")
        host_props = None
    if not want_grounding:
        return legs
    # the grounding leg scores clip sequences of T = ceil(video_len / 8) in [60, 400]: its queries come from the classification
    # stage run (untimed) on videos whose lengths are spread accordingly -- the 16 videos above all have video_len ~ 2000, T ~ 245
    del packed
    g_lens = [480, 800, 1100, 1400, 1800, 2200, 2700, 3200]
    g_props = [device_proposal(5321 + v, ntr, width, cfg["num_enti_cats"], (15, max(30, L // 2)), (0, max(30, L // 2)), device) for v, L in enumerate(g_lens)]
    video_len = [p.video_len for p in g_props]
    nv = len(g_props)
    g_packed = PackedBatch(g_props, device)
    g_props, g_packed.proposals = None, None
    out_holder["o"] = big.forward_packed(g_packed, 3)
    big.counts(out_holder["o"])
    results = big.unpack_results(nv, list(range(nv)), out_holder["o"])
    del g_packed
    gcfg = dict(configs.CONFIG_G)
    gcfg.setdefault("loss_factor", {})
    gsd = seeded_state_dict(grd_state_shapes(gcfg), 12, 4.0)
    d = tempfile.mkdtemp()
    np.save(d + "/ge.npy", gsd["EntiNameEmb"].numpy())
    np.save(d + "/gp.npy", gsd["PredNameEmb"].numpy())
    grd = DEBUG(dict(gcfg, EntiNameEmb_path=d + "/ge.npy", PredNameEmb_path=d + "/gp.npy"), is_train=False)
    grd.load_state_dict(gsd)
    grd = grd.to(device)
    jobs = []
    rs = np.random.RandomState(99)
    for v, r in enumerate(results):
        if r is None:
            continue
        quint, scores, dura = r[0], r[1], r[2]
        if quint.shape[0] > 200:                     # rt_triplets_topk (experiments/exp6/config_rt200.py:14)
            keep = torch.sort(torch.argsort(scores.mean(dim=-1), descending=True)[:200])[0]
            quint, dura = quint[keep], dura[keep]
        T = -(-video_len[v] // 8)
        clip = torch.from_numpy((0.1 * rs.standard_normal((T, gcfg["dim_feat"]))).astype(np.float32)).to(device)
        jobs.append((clip, quint.to(device), dura.to(device), video_len[v]))
    del big, out_holder
    if not jobs:
        return legs
    n_q = sum(int(j[1].shape[0]) for j in jobs)

    def grd_step():
        if args.grd_one_video_per_call:
            for clip, quint, dura, vl in jobs:
                last = grd.forward_raw(clip, quint, dura, vl, 0.9, 0.5, 0.2, 0.8)
        else:
            last = grd.forward_raw_batch(jobs, 0.9, 0.5, 0.2, 0.8)[-1]
        if args.grd_one_video_per_call:
            last["status"].cpu()         # the reference's loop reads every video's result before the next forward
        # batched: results stay on the device; like the headline loop, the timed region waits for the GPU once at its end, so
        # the host's packing of the next batch overlaps this one's kernels

    # ---- parity of THIS leg's workload (outside the timing): the two videos with the fewest query x clip rows through the CPU oracle
    # (oracle/grd_oracle.py: models/grd_model_v5.py:310-373,530-576,667-737 restated) -- the three heads of forward_propagation against the
    # HIP path's, and the final pooled spans / bin probabilities / bin masks of the batched call
    grd_parity = None
    if args.grd_oracle_videos > 0:
        from oracle import grd_oracle as G
        torch.set_num_threads(usable_cores())
        order = sorted(range(len(jobs)), key=lambda i: int(jobs[i][0].shape[0]) * int(jobs[i][1].shape[0]))[:args.grd_oracle_videos]
        outs = grd.forward_raw_batch(jobs, 0.9, 0.5, 0.2, 0.8)
        k = gcfg["num_bins"]
        hm, pm, bm, m_same, m_all, t0 = 0.0, 0.0, 0.0, 0, 0, time.perf_counter()
        shapes = []
        with torch.no_grad():
            for i in order:
                clip, quint, dura, vl = jobs[i]
                n, T = int(quint.shape[0]), int(clip.shape[0])
                shapes.append((n, T))
                one = grd.forward_raw(clip, quint, dura, vl, 0.9, 0.5, 0.2, 0.8, keep_intermediates=True)
                words, info = G.prepare_data(gsd, quint.cpu(), dura.cpu(), vl)
                regrs, conf, cls = G.forward_propagation(gsd, clip.cpu(), words, info)
                for key, want, w in (("regrs", regrs, 2 * k), ("conf", conf, k), ("cls", cls, k)):
                    got = one[key].cpu().numpy().reshape(n, T, w)
                    hm = max(hm, float(np.abs(got - want.numpy()).max() / np.abs(want.numpy()).max()))
                pooled, probs, mask = G.forward(gsd, gcfg, clip.cpu(), (quint.cpu(), dura.cpu(), vl), 0.9, 0.5, 0.2, 0.8)
                o = outs[i]
                pm = max(pm, float((o["pooled_se"].cpu() - pooled).abs().max()))
                bm = max(bm, float((o["bins_probs"].cpu() - probs).abs().max()))
                m_same += int((o["bins_mask"].cpu().bool() == mask.bool()).sum())
                m_all += int(mask.numel())
        grd_parity = dict(videos=len(order), queries_x_clips=["%d x %d" % s for s in shapes], heads_max_rel=hm, pooled_se_max_abs=pm, bins_probs_max_abs=bm,
                          bins_mask_identical="%d of %d" % (m_same, m_all), oracle_seconds=round(time.perf_counter() - t0, 1),
                          note="oracle/grd_oracle.py on videos of this leg: heads relative to max |head output|, spans (normalised to [0, 1]) and probabilities absolute")
        del outs
    gsteps = 5
    el, el_prof, fam = timed_with_events(lib, grd_step, 2, gsteps)
    by_fam, share, gflop_step, gbyte_step = family_rooflines(fam, gsteps, 1e3 * el_prof / gsteps)
    clip_bytes = sum(int(j[0].numel()) * 4 for j in jobs)
    weight_bytes = 4 * sum(int(v.numel()) for v in gsd.values())
    out_bytes = n_q * (gcfg["num_bins"] + 1) * (2 * 4 + 4 + 1)
    # ~23 GFLOP per video of 200 queries x 120 clips; this leg's videos have other n x T, so the figure is scaled by sum n T
    nt_sum = sum(int(j[1].shape[0]) * int(j[0].shape[0]) for j in jobs)
    grd_step = dict(gflop_per_step_counted=round(gflop_step, 1), gflop_per_step_survey_8d_scaled=round(23.0 * nt_sum / (200 * 120), 1),
                    algorithmic_gbytes_per_step_all_launches=round(gbyte_step, 3),
                    compulsory_bytes=dict(clip_features=clip_bytes, weights=weight_bytes, outputs=out_bytes, total=clip_bytes + weight_bytes + out_bytes),
                    roofline_share_of_step=share,
                    note="every family of the leg with its own bound (rooflines_by_family): the d = 128 GEMMs, depth-wise stages, LayerNorm and attention are "
                         "HBM-bound on the INTERMEDIATES they exchange (32 flop per byte at d = 128); against the compulsory bytes alone (clip features + weights + outputs) "
                         "the step would take %.1f us at 8 TB/s" % (1e6 * (clip_bytes + weight_bytes + out_bytes) / (HBM_PEAK_GBS * 1e9)))
    legs["grounding"] = dict(value=round(n_q * gsteps / el, 1), unit="triplet-queries/s", ms_per_step=round(1e3 * el / gsteps, 3), rooflines_by_family=by_fam, step=grd_step,
                             ms_per_video=round(1e3 * el / gsteps / len(jobs), 3), steps=gsteps, ms_per_step_of_each_repeat=timed_with_events.last_repeats_ms,
                             workload="grounding stage (BASELINE configs[2]/[3]): grd_model_v5 DEBUG eval forward incl. temporal pooling + NMS, "
                                      "config G, %d videos, %d queries (<=200 per video), T = ceil(video_len/8) in [%d,%d]"
                                      % (len(jobs), n_q, min(int(j[0].shape[0]) for j in jobs), max(int(j[0].shape[0]) for j in jobs)),
                             roofline=gemm_roofline(fam, gsteps, 1e3 * el_prof / gsteps), oracle_parity=grd_parity,
                             # the leg as a whole against the MFMA roof (its flops are all split-f16 products: dense f16 peak / 3): the honest
                             # figure for a step whose compulsory bytes are 270 x smaller than the intermediates the per-family HBM fractions price
                             roofline_mfma=dict(bound="mfma", achieved=round(gflop_step / (1e3 * el / gsteps), 2), peak=round(F16_MFMA_PEAK_TFLOPS / 3, 1), unit="TFLOP/s",
                                                frac=round(gflop_step / (1e3 * el / gsteps) / (F16_MFMA_PEAK_TFLOPS / 3), 4), gflop_per_step=round(gflop_step, 1),
                                                note="all counted flops of the step / the un-instrumented step time"),
                             kernel_families_ms_per_step={k: round(v["ms"] / gsteps, 3) for k, v in fam.items()})
    return legs


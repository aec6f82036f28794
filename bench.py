#!/usr/bin/env python3
"""Benchmark of the BIG classification hot path (BIG-C eval forward incl. construct_triplet).

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json / SURVEY.md section 8d): tracklet-pairs/sec = videos/s x N(N-1), whole job.
Workload = BASELINE.json configs[1]: VidVRD-test classification-only, model_0v10 at the exp2 dimensions (PKU RoI 2048-d + I3D 832-d),
synthetic videos of N=64 tracklets (SURVEY.md 8d recipe, video v uses numpy RandomState(1234+v)), numpy-seeded weights.  One step = ONE
CALL OF THE DROP-IN SIGNATURE, `model(proposal_list, topk)` (models/model_0v10.py:359-388), on `--videos` (default 256) device-resident
TrajProposals per GPU: the ragged batch's tables and boxes (the feature rows are read in place through a row-address table), the forward,
construct_triplet, the compaction to the reference's return types, the one D2H copy and the per-video result tuples are all inside the
timed region.  N>1: one process per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks timing -- videos are
independent, no data-path collective), weak scaling.  Started without a torch.distributed.run environment, `--gpus N` (N>1) launches the
N ranks itself (a child `python -m torch.distributed.run`, started before this process touches the GPU).

OUTPUT.  The LAST line of stdout is ONE compact JSON object (< 4 KB, key names <= 32 characters): the contract keys, `roofline` (the GEMM
family: algorithmic flops of its launches / their HIP-event time on the launch stream, against the dense f16 MFMA peak / 3),
`cpu_baseline` (the CPU oracle, kind "port", on a bounded sample of the same videos), `parity` (rows and logits of that sample against the
oracle and against a float64 evaluation), and one {value, ms} pair per other leg.  Everything else -- per-family kernel times, the legs'
own rooflines, notes -- goes to `bench_detail.json` (repo root, and gpurun_out/ when that directory exists); `--print-detail` also prints
it as one line BEFORE the compact line.

The legs (benchmarks/legs/*.py; rank 0, N=1, all outside the timed region, each skipped once `--budget-s` seconds have passed):
  parity.py           cpu_baseline + oracle_parity          recall.py   Recall@50 parity on a trained model (the metric's second half)
  vidor_grounding.py  BASELINE configs[2] / [3]             train.py    BASELINE configs[4] on one GPU (8 and 64 videos per step)
  modes.py            exact f32 MFMA / plain accumulation / reduced precision (f16, bf16; Recall-gated)
  batch512.py         the headline workload at twice the videos per step
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--videos", type=int, default=256, help="videos per GPU per step")
    ap.add_argument("--n-trajs", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--cpu-videos", type=int, default=256, help="videos of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="stop the CPU-baseline sample after this much oracle time")
    ap.add_argument("--f64-videos", type=int, default=32, help="of the CPU sample, the videos that are also evaluated in float64 (the truth the HIP and the "
                                                                 "f32-oracle logits are measured against; ~0.15 s per video, outside every timing)")
    ap.add_argument("--headline", choices=["api", "packed"], default="api", help="what `value` times: api = model(proposal_list, topk) on device-resident "
                    "TrajProposals (pack + forward + results: the drop-in call), packed = the forward on a resident PackedBatch")
    ap.add_argument("--no-prof", action="store_true", help="do not record HIP events around the kernels")
    ap.add_argument("--mode", choices=["eval", "train"], default="eval", help="eval = headline metric; train = training-step throughput")
    ap.add_argument("--train-videos", type=int, default=8, help="videos per GPU per training step (reference batch size 8)")
    ap.add_argument("--exact-steps", type=int, default=3, help="extra steps timed with the exact f32-MFMA GEMM kernel, reported beside (0 = skip)")
    ap.add_argument("--fast-steps", type=int, default=3, help="extra steps timed in the opt-in reduced-precision mode (one f16 MFMA per product), "
                                                             "reported beside with its Recall@50 gate (0 = skip)")
    ap.add_argument("--recall-steps", type=int, default=300, help="training iterations of the Recall@50 leg (0 = skip the leg)")
    ap.add_argument("--recall-videos", type=int, default=16, help="held-out videos the Recall@50 leg scores (HIP, HIP reduced precision, CPU oracle)")
    ap.add_argument("--legs", default="vidor,grounding,train,batch512", help="extra workloads reported in the same line at N=1 ('' = none)")
    ap.add_argument("--vidor-videos", type=int, default=16, help="videos of the VidOR leg (SURVEY.md 8d: 16 x N = 180, L in [15, 1000))")
    ap.add_argument("--vidor-oracle-videos", type=int, default=2, help="of those, host-seeded ones that are also run through the CPU oracle (outside the timing)")
    ap.add_argument("--grd-oracle-videos", type=int, default=2, help="videos of the grounding leg that are also run through the CPU oracle (outside the timing)")
    ap.add_argument("--grd-one-video-per-call", action="store_true", help="grounding leg: the reference's one-video forward instead of the batched one")
    ap.add_argument("--budget-s", type=float, default=110.0, help="seconds after process start past which the remaining optional legs are skipped (listed in `skipped`); "
                                                                  "the headline, roofline, cpu_baseline and parity are never skipped")
    ap.add_argument("--print-detail", action="store_true", help="also print bench_detail.json's content as one line BEFORE the compact final line")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `torch.distributed.run` (never an
    exec, and before this process has initialised the GPU), relay its output, return its exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


ARGS = parse() if __name__ == "__main__" else None
if ARGS is not None and (ARGS.gpus > 1 or os.environ.get("BIGC_BENCH_SELF_LAUNCH") == "1") and "RANK" not in os.environ:
    sys.exit(self_launch(ARGS))

T_START = time.time()

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
from legs import common as C  # noqa: E402
from legs.common import (EXACT_F32, F32_MFMA_PEAK_TFLOPS, PackedBatch, _lib, bigc_state_shapes, build_model, collect_families, configs,  # noqa: E402
                         gemm_roofline, generate_on_device, init_dist, seeded_state_dict, synthetic_proposal, throttled_periods, usable_cores)
from legs.train import make_train_job, train_main, train_step_fn  # noqa: E402,F401  (re-exported: benchmarks/train_*.py use them)

COMPACT_LIMIT = 4096        # bytes of the final stdout line (a 20 KB line was not parsed by the driver)


def over_budget(args):
    return time.time() - T_START > args.budget_s


def pmc_traffic(args, lin, rows_per_step, width, sd):
    """HBM bytes per GEMM launch / per step: they cannot be sampled from inside this process and come from the committed rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of this same command (profiles/pmc_traffic.json, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950); null when the workload or the kernel sources differ from the ones the passes ran on"""
    traffic, source, step_traffic = None, None, None
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(path) and args.videos == 256 and args.n_trajs == 64:
        pmc = json.load(open(path))
        from pmc_summary import csrc_sha16
        if pmc.get("csrc_sha16") == csrc_sha16():
            traffic = round(pmc["hbm_bytes_per_launch"])
            step_traffic = pmc.get("whole_step_fetch_bytes_all_kernels", 0) + pmc.get("whole_step_write_bytes_all_kernels", 0)
            source = "not measured in this run: profiles/pmc_traffic.json (%s), rocprofv3 --pmc passes of this command on these kernel sources" % pmc.get("tag", "")
        else:
            source = "profiles/pmc_traffic.json was measured on other kernel sources (csrc fingerprint differs): not reported"
    # SURVEY.md 8(d)'s algorithmic bytes of the STEP: features + boxes read once (6.7 kB per pair) + the weights once per batch
    alg_8d = rows_per_step * (width + 4) * 4 + 4 * sum(int(v.numel()) for v in sd.values())
    return dict(traffic=traffic, traffic_source=source, algorithmic_bytes_per_launch=round(lin["bytes"] / lin["launches"]),
                step_traffic_all_kernels=dict(hbm_bytes=step_traffic, algorithmic_bytes_survey_8d=alg_8d,
                                              ratio=round(step_traffic / alg_8d, 2) if step_traffic else None,
                                              note="HBM bytes of every kernel of one step (same PMC passes) against features + boxes + weights read once"))


def compact_line(d):
    """The final stdout line: the contract keys + roofline + cpu_baseline + parity + one {value, ms} pair per leg.  Short keys, numbers only
    where possible; everything else lives in bench_detail.json."""
    def pick(o, *keys):
        return {k: o[k] for k in keys if o is not None and k in o} if o else None

    def vm(o):
        return {"value": o["value"], "ms": o["ms_per_step"]} if o and "value" in o else None

    r, op = d.get("roofline"), d.get("oracle_parity")
    line = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    c = d["config"]
    line["config"] = {"workload": "BASELINE configs[1]: BIG-C model_0v10 eval forward + construct_triplet, exp2 dims, f32 (split-f16 MFMA)",
                      "videos_per_gpu": c["videos_per_gpu_per_step"], "tracklets": c["tracklets_per_video"], "pairs_per_gpu": c["pairs_per_gpu_per_step"],
                      "rows": c["rows_per_step"], "topk": c["topk"], "parallelism": c["parallelism"], "timed_call": "model(proposal_list, topk)"}
    if r:
        line["roofline"] = pick(r, "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "launches_per_step", "avg_launch_us", "share_of_step")
        line["roofline"]["kernel"] = "split-f16 GEMM family"
        st = r.get("step_traffic_all_kernels") or {}
        line["roofline"]["step_traffic_ratio"] = st.get("ratio")
    else:
        line["roofline"] = None
    line["cpu_baseline"] = pick(d.get("cpu_baseline"), "value", "unit", "cores", "kind")
    if d.get("cpu_baseline"):
        line["cpu_baseline"]["sample"] = "%d videos, %.1f s" % (op["videos"], d["cpu_baseline_seconds"])
    if op:
        f64 = op.get("distance_to_float64") or {}
        line["parity"] = {"videos": op["videos"], "rows": op["identical_triplet_rows"], "rows_off_both_oracles": op["rows_differing_from_both_oracles"],
                          "f64_videos": f64.get("videos"), "hip_vs_f64_max": f64.get("hip_max"), "per_row_max": (f64.get("per_query_row") or {}).get("hip_max"),
                          "f32_oracle_vs_f64_max": f64.get("f32_oracle_max"), "videos_over_1e_4": f64.get("hip_videos_over_1e_4")}
    else:
        line["parity"] = None
    legs = {"resident": vm(d.get("resident_packed_batch")), "pipelined": vm(d.get("forward_api_pipelined")), "new_lists": vm(d.get("forward_api_new_containers")),
            "plain_accum": vm(d.get("plain_accumulation")), "exact_f32": vm(d.get("exact_f32_gemm_mode")), "fast_f16": vm(d.get("fast_f16_gemm_mode")),
            "fast_bf16": vm((d.get("fast_f16_gemm_mode") or {}).get("bf16")), "vidor": vm(d.get("vidor")), "batch_x2": vm(d.get("batch_x2"))}
    g = d.get("grounding")
    if g and "value" in g:
        legs["grounding"] = {"value": g["value"], "ms": g["ms_per_step"], "unit": "queries/s", "mfma_frac": (g.get("roofline_mfma") or {}).get("frac")}
    t = d.get("train") or {}
    for k in ("videos_8", "videos_64"):
        if k in t:
            legs["train_" + k[7:]] = {"value": t[k]["value"], "ms": t[k]["ms_per_step"], "gemm_frac": (t[k].get("roofline") or {}).get("frac"),
                                      "attn_bwd_ms": (t[k].get("kernel_families_ms_per_step") or {}).get("attention_bwd")}
    gt = d.get("grd_train")
    if gt and "value" in gt:
        legs["grd_train_8"] = {"value": gt["value"], "ms": gt["ms_per_step"], "unit": "videos/s"}
    rc = d.get("recall")
    if rc and "recall50_hip" in rc:
        legs["recall"] = {"r50_hip": round(rc["recall50_hip"], 4), "r50_oracle": round(rc["recall50_oracle"], 4), "gate": rc["parity_gate_passed"],
                          "fast_gate": rc.get("fast_gate_passed"), "bf16_gate": rc.get("bf16_gate_passed")}
    line["legs"] = {k: v for k, v in legs.items() if v is not None}
    line["skipped"] = d.get("skipped_legs") or []
    line["host_ms"] = d.get("api_host_ms_total")
    line["run_s"] = d.get("run_s")
    line["detail"] = "bench_detail.json"
    s = json.dumps(line, separators=(",", ":"))
    if len(s) >= COMPACT_LIMIT:         # never print an unparsable line: drop the legs first
        line["legs"] = {}
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) < COMPACT_LIMIT, len(s)
    return s


def emit(detail, args):
    detail["run_s"] = round(time.time() - T_START, 1)
    blob = json.dumps(detail)
    for path in (os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        if os.path.isdir(os.path.dirname(path)):
            try:
                open(path, "w").write(blob + "\n")
            except OSError:
                pass
    if args.print_detail:
        print(blob)
    C.print_last(compact_line(detail))


def main():
    args = ARGS
    if args.mode == "train":
        return train_main(args)
    rank, world, device, dist = init_dist(args)

    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    # each rank owns its own shard of synthetic videos (weak scaling): seeds 1234 + rank*videos + v
    base = 1234 + rank * args.videos
    t0 = time.time()
    props = generate_on_device(lambda v: synthetic_proposal(base + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"]),
                               args.videos, device)
    torch.cuda.synchronize()
    n_cpu = min(args.cpu_videos, args.videos) if (rank == 0 and world == 1) else 0
    gen_s = time.time() - t0
    model = build_model(cfg, sd, device)
    lib = _lib.load()

    # ---- the step that is timed: the drop-in call itself, model(proposal_list, topk) (models/model_0v10.py:359-388) on device-resident
    # TrajProposals -- everything the product does per batch is inside: the ragged batch's tables and boxes (the feature rows are read in
    # place through a row-address table, csrc/linear_split.hip / frontend.hip), the forward, construct_triplet, the compaction to the
    # reference's return types, the one D2H copy (row offsets + range flag) and the per-video result tuples
    def api_step():
        return model(props, topk=args.topk)

    # ... and the same forward on a batch that stays packed in HBM (a resident PackedBatch, the engine-level entry point): what a
    # loop over resident data pays per pass; reported as `resident_packed_batch`, never the headline
    packed = PackedBatch(props, device, "lazy")
    if os.environ.get("BIGC_FEAT_HR8", "1") != "0":
        packed.materialize("hr8", cfg["dim_feat"])      # the resident form: RoI columns as f16 pairs + I3D columns f32, gathered by one launch
    pairs_per_step, rows_per_step = packed.n_pairs(), packed.n_rows

    def packed_step():
        out = model.forward_packed(packed, args.topk)
        return model.counts(out)       # the one D2H copy forward() needs (row offsets + the GEMM range flag) -> host sync

    step = packed_step if args.headline == "packed" else api_step
    for _ in range(args.warmup):
        step()
    # ---- the timed region: K steps, no instrumentation
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    thr0, cpu0 = throttled_periods(), time.process_time()
    t0 = time.perf_counter()
    stamps = [t0]
    for _ in range(args.steps):
        last = step()
        stamps.append(time.perf_counter())            # (every step ends with its own D2H copy: the stamps are step boundaries)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # what the host was doing meanwhile (diagnostics, outside `value`): CPU seconds of ALL threads of this process per second of wall
    # clock, and how often the container's CPU quota froze the process during the timed steps (cgroup cpu.stat)
    host_diag = {"cpu_threads_busy": round((time.process_time() - cpu0) / max(elapsed, 1e-9), 2), "cpu_quota": usable_cores(),
                 "torch_threads": torch.get_num_threads(), "quota_throttled_periods_during_timed_steps": (throttled_periods() - thr0) if thr0 is not None else None,
                 "load_average_1min": round(os.getloadavg()[0], 1)}
    elapsed_this_rank = elapsed
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # who took part (a collective on every rank): rank, host, device ordinal + UUID, each rank's own step time -- the N > 1 line proves
    # through the job's own RCCL group that it ran one process per distinct GPU
    from bigc_hip.shard import rank_report
    ranks = rank_report(dist, device, elapsed_this_rank, args.steps)
    n_triplets_last = int(sum(r[0].shape[0] for r in last if r is not None)) if args.headline != "packed" else int(last.sum())

    # ---- the same K steps once more with a HIP event pair around every kernel launch (on the launch stream, inside the library):
    # per-family launch durations for the roofline.  Kept out of `value` because the ~300 event records per step cost a few % of a
    # step; the instrumented wall time is reported next to it.  (Every rank runs it; rank 0 reports.)
    families, prof_elapsed = {}, None
    if not args.no_prof:
        lib.bigc_prof_reset()
        lib.bigc_prof_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        prof_elapsed = time.perf_counter() - t0
        lib.bigc_prof_enable(0)
        families = collect_families(lib, args.steps)

    # every rank leaves the process group HERE, at the same point (the legs below are rank 0's alone and use no collective): a teardown on
    # rank 0 long after its peers have exited is the one asymmetric step of the N > 1 run, and the RCCL banner a rank prints when its
    # communicator goes away then lands well before the final line
    if dist is not None:
        dist.destroy_process_group()
        dist = None
    if rank != 0:
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = pairs_per_step * world * args.steps / elapsed
    detail = {
        "metric": "tracklet_pairs_per_sec", "value": round(value, 1), "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "VidVRD-test classification-only (BASELINE configs[1]): BIG-C model_0v10 eval forward incl. "
                               "construct_triplet, exp2 dims (RoI 2048 + I3D 832), %d synthetic videos x N=%d tracklets "
                               "per GPU per step, topk=%d, f32 in / f32 accumulate / f32 out (%s)"
                               % (args.videos, args.n_trajs, args.topk, "GEMMs on the exact f32 MFMA" if EXACT_F32 else
                                  "GEMMs and attention on f16 MFMAs with exactly split f32 operands, f32-level error"),
                   "videos_per_gpu_per_step": args.videos, "tracklets_per_video": args.n_trajs, "topk": args.topk,
                   "rows_per_step": rows_per_step, "pairs_per_gpu_per_step": pairs_per_step, "parallelism": "dp%d" % world},
        "videos_per_sec": round(args.videos * world * args.steps / elapsed, 1),
        "triplets_last_step": n_triplets_last, "setup_s": round(gen_s, 1),
        "timed_steps_ms": [round(1e3 * (b - a), 2) for a, b in zip(stamps[:-1], stamps[1:])],
        "host": host_diag, "ranks": ranks,
        "kernel_families_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in families.items()},
        "timed_call": ("model(proposal_list, topk) on device-resident TrajProposals: tables + boxes of the ragged batch (the feature rows are read in place "
                       "through a row-address table, no copy) + forward + construct_triplet + compaction + one D2H copy + per-video result tuples" if args.headline == "api" else
                       "model.forward_packed(resident PackedBatch, topk) + the one D2H copy"),
        "modes": {"BIGC_FEAT_HR8": os.environ.get("BIGC_FEAT_HR8", "1"), "BIGC_PRESPLIT": os.environ.get("BIGC_PRESPLIT", "auto"),
                  "BIGC_CHAINED": os.environ.get("BIGC_CHAINED", "1"), "BIGC_FEAT_IN_PLACE": os.environ.get("BIGC_FEAT_IN_PLACE", "1"),
                  "resident_batch_holds_split_features": bool(packed.feats_hr8 is not None)},
        "skipped_legs": [],
    }
    me = dict(value=round(value, 1), unit="pairs/s", ms_per_step=round(ms_per_step, 3), steps=args.steps, note="= the headline")

    roofline = None
    if "linear_f32" in families:
        roofline = gemm_roofline(families, args.steps, 1e3 * prof_elapsed / args.steps)
        # what the matrix pipe of this chip sustains on operands that toggle (a register-only MFMA loop, benchmarks/probes/mfma_peak.hip,
        # committed as profiles/mfma_peak.json): context for `frac`, which stays priced against the guide's 2.5 PFLOP/s
        mp_path = os.path.join(ROOT, "profiles", "mfma_peak.json")
        if os.path.exists(mp_path) and not EXACT_F32:
            mp = json.load(open(mp_path))
            roofline["sustained_mfma_peak"] = dict(tflops_f16_random_operands=mp["random_f16_tflops"], tflops_f16_zero_operands=mp["zeros_tflops"],
                                                   three_mfma_gemm_ceiling=round(mp["random_f16_tflops"] / 3, 1),
                                                   frac_of_that_ceiling=round(roofline["achieved"] / (mp["random_f16_tflops"] / 3), 4),
                                                   source="not measured in this run: profiles/mfma_peak.json (benchmarks/probes/mfma_peak.hip)")
        roofline.update(kernel=("linear_f32_kernel (f32 MFMA 32x32x2)" if EXACT_F32 else
                                # every f32 product costs three f16 MFMAs (head x head, head x residual, residual x head): the roof for
                                # ALGORITHMIC flops is the dense f16 MFMA peak / 3
                                "split-operand GEMM family (f32 operands as 2 x f16, 3 x v_mfma_f32_32x32x16_f16 per product)"),
                        vs_f32_mfma_peak=round(roofline["achieved"] / F32_MFMA_PEAK_TFLOPS, 3),
                        measured="HIP events around each launch, over a repeat of the timed %d steps (instrumented step: %.3f ms)"
                                 % (args.steps, 1e3 * prof_elapsed / args.steps))
        roofline.update(pmc_traffic(args, families["linear_f32"], rows_per_step, width, sd))
    detail["roofline"] = roofline

    # ---- the other of the two entry points, same K steps (outside `value`), with its own GEMM roofline (the two paths differ in the
    # K = 2048 GEMM: rows read in place vs the DMA-only kernel on the stored split form)
    other = api_step if args.headline == "packed" else packed_step
    el_o, el_op, fam_o = C.timed_with_events(lib, other, max(1, args.warmup), args.steps, repeats=1) if not args.no_prof else (None, None, None)
    if el_o is None:
        for _ in range(max(1, args.warmup)):
            other()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            other()
        torch.cuda.synchronize()
        el_o = time.perf_counter() - t1
    other_leg = dict(value=round(pairs_per_step * args.steps / el_o, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * el_o / args.steps, 3), steps=args.steps)
    if fam_o:
        other_leg["roofline"] = gemm_roofline(fam_o, args.steps, 1e3 * el_op / args.steps)
    detail["forward_api"], detail["resident_packed_batch"] = (me, other_leg) if args.headline == "api" else (other_leg, me)

    # the API in two halves (model.forward_async: the host side of call i + 1 -- tables, launches -- runs while the GPU works on call i),
    # as tools/eval_vidvrd.py loops over its chunks; same kernels, same results; a named extra, never the headline
    pend = None
    for _ in range(2):
        nxt = model.forward_async(props, topk=args.topk)
        if pend is not None:
            pend.result()
        pend = nxt
    pend.result()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pend = None
    for _ in range(args.steps):
        nxt = model.forward_async(props, topk=args.topk)
        if pend is not None:
            pend.result()
        pend = nxt
    last_async = pend.result()
    torch.cuda.synchronize()
    el_p = time.perf_counter() - t1
    detail["forward_api_pipelined"] = dict(value=round(pairs_per_step * args.steps / el_p, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * el_p / args.steps, 3), steps=args.steps,
                                           results_identical_to_the_synchronous_call=bool(all((a is None) == (b is None) and (a is None or all(torch.equal(x, y) for x, y in zip(a, b)))
                                                                                             for a, b in zip(last_async, api_step()))),
                                           note="model.forward_async(proposal_list) / .result(): two calls in flight")
    del last_async, pend, nxt
    # host-side cost of the API call, each piece measured on its own with the GPU idle: the tables of the pack built from scratch, the kept
    # batch of the same containers revalidated and refreshed (what the timed loop pays from its second call on), and the results as the
    # lazy list the call returns / as 4 x B eagerly cut views
    tp = time.perf_counter()
    for _ in range(3):
        PackedBatch(props, device, "lazy")
    torch.cuda.synchronize()
    pack_host_ms = 1e3 * (time.perf_counter() - tp) / 3
    model(props, topk=args.topk)
    tp = time.perf_counter()
    for _ in range(3):
        model._pack(props, device)
    torch.cuda.synchronize()
    repack_host_ms = 1e3 * (time.perf_counter() - tp) / 3
    o = model.forward_packed(packed, args.topk)
    model.counts(o)
    tp = time.perf_counter()
    for _ in range(3):
        model.unpack_results(packed.n_vid, list(range(packed.n_vid)), o)
    unpack_host_ms = 1e3 * (time.perf_counter() - tp) / 3
    tp = time.perf_counter()
    for _ in range(3):
        model.unpack_results(packed.n_vid, None, o, lazy=True)
    lazy_host_ms = 1e3 * (time.perf_counter() - tp) / 3
    del o
    detail["api_host_ms"] = {"pack_tables_built": round(pack_host_ms, 2), "pack_tables_kept_and_refreshed": round(repack_host_ms, 2),
                             "result_tuples_eager": round(unpack_host_ms, 2), "result_list_lazy": round(lazy_host_ms, 2),
                             "note": "host side of one model(proposal_list) call at this batch size, each piece measured alone with the GPU idle; the timed "
                                     "call pays `kept_and_refreshed` + `lazy` (the same containers come back: models/_bigc.py _pack), a first call `built`"}
    detail["api_host_ms_total"] = round(repack_host_ms + lazy_host_ms, 2)
    # ... and the same call when the containers are NOT recognised (the kept batch dropped before every call: a loop over ever new videos)
    for _ in range(2):
        object.__setattr__(model, "_last_pack", None)
        api_step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        object.__setattr__(model, "_last_pack", None)
        r_ = api_step()
        sum(1 for x in r_ if x is not None)                # (every entry read: the lazy list is materialised inside the timing)
    torch.cuda.synchronize()
    el_f = time.perf_counter() - t1
    detail["forward_api_new_containers"] = dict(value=round(pairs_per_step * args.steps / el_f, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * el_f / args.steps, 3),
                                                steps=args.steps, note="model(proposal_list, topk) with the tables rebuilt on every call and every result entry read")
    # SURVEY.md 8(d) quotes the MEDIAN step time: a few steps once more, each one waited for (outside `value`)
    per_step = []
    for _ in range(min(args.steps, 10)):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        per_step.append(time.perf_counter() - t1)
    detail["ms_per_step_median_each_step_synchronised"] = round(1e3 * float(np.median(per_step)), 3) if per_step else None

    # the outputs that are compared with the oracle: those of the TIMED path (a batch gathered the way model(proposal_list) gathers it)
    def timed_batch():
        return PackedBatch(props, device, "lazy") if args.headline == "api" else packed
    ref_out = model.forward_packed(timed_batch(), args.topk, keep_intermediates=False)

    from legs import modes as M
    plain_logits = None
    if args.exact_steps > 0 and not EXACT_F32 and lib.bigc_get_accum_policy() != 0:
        detail["plain_accumulation"], plain_logits = M.plain_accumulation_leg(args, lib, model, timed_batch, step, ref_out, pairs_per_step,
                                                                                (args.videos, cfg["num_querys"], cfg["num_pred_cats"]))
    # ---- cpu_baseline + oracle_parity: required objects of the line, never skipped by the budget
    if n_cpu:
        from legs.parity import cpu_baseline_and_parity
        t1 = time.time()
        detail["cpu_baseline"], detail["oracle_parity"], plain_dist = cpu_baseline_and_parity(args, lib, cfg, sd, base, width, ref_out, plain_logits, n_cpu)
        detail["cpu_baseline_seconds"] = float(detail["cpu_baseline"]["sample"].rsplit(",", 1)[1].split()[0])
        detail["parity_leg_seconds"] = round(time.time() - t1, 1)
        if plain_dist and "plain_accumulation" in detail:
            detail["plain_accumulation"]["distance_to_float64"] = plain_dist
    else:
        detail["cpu_baseline"], detail["oracle_parity"] = None, None
    del plain_logits

    want = [x for x in args.legs.split(",") if x and x != "none"] if world == 1 else []

    def run_leg(name, fn):
        if over_budget(args):
            detail["skipped_legs"].append(name)
            return None
        t1 = time.time()
        out = fn()
        if isinstance(out, dict):
            out["leg_seconds"] = round(time.time() - t1, 1)
        return out

    if args.exact_steps > 0 and not EXACT_F32:
        detail["exact_f32_gemm_mode"] = run_leg("exact_f32", lambda: M.exact_mode_leg(args, lib, model, packed, packed_step, ref_out, pairs_per_step))
    if args.fast_steps > 0 and not EXACT_F32 and world == 1 and not lib.bigc_linear_get_fast():
        detail["fast_f16_gemm_mode"] = run_leg("fast_f16", lambda: M.fast_mode_leg(args, lib, model, packed, packed_step, ref_out, cfg, base, width, pairs_per_step, device))
    del ref_out
    model._engine = None
    del model, packed, last
    torch.cuda.empty_cache()
    if "batch512" in want and not EXACT_F32:
        from legs.batch512 import batch512_leg
        detail["batch_x2"] = run_leg("batch_x2", lambda: batch512_leg(args, lib, device, props))
    del props
    torch.cuda.empty_cache()
    if args.recall_steps > 0 and world == 1:
        from legs.recall import recall_leg
        detail["recall"] = run_leg("recall", lambda: recall_leg(args, lib, device))
    if "grounding" in want or "vidor" in want:
        from legs.vidor_grounding import vidor_grounding_legs
        got = run_leg("vidor+grounding", lambda: vidor_grounding_legs(args, lib, device, "vidor" in want, "grounding" in want))
        detail.update({k: v for k, v in (got or {}).items() if k != "leg_seconds"})
        torch.cuda.empty_cache()
    if "train" in want:
        from legs.train import train_leg
        detail["train"] = run_leg("train", lambda: train_leg(args, lib, device, budget=lambda: over_budget(args)))
        if "grounding" in want:
            from legs.train import grd_train_leg
            detail["grd_train"] = run_leg("grd_train", lambda: grd_train_leg(args, lib, device))
    if world > 1:
        time.sleep(1.0)                    # the other ranks (whose output the launcher merges into the same stream) have exited by then
    emit(detail, args)


if __name__ == "__main__":
    main()

"""Helpers shared by bench.py and the legs under benchmarks/legs/ (the model factory, the HIP-event family timers, the roofline arithmetic)."""
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from bigc_hip import _lib, configs  # noqa: E402
from bigc_hip.packed import PackedBatch  # noqa: E402
from bigc_hip.synthetic import TrajProposal, bigc_state_shapes, seeded_state_dict, synthetic_proposal  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0   # same table, "Peak BF16/FP16 MFMA" dense
HBM_PEAK_GBS = 8000.0           # same table, HBM3E peak
EXACT_F32 = os.environ.get("BIGC_LINEAR_EXACT", "0") not in ("", "0")     # run the whole benchmark on the exact f32-MFMA GEMM kernel


def throttled_periods():
    """scheduling periods in which the cgroup's CPU quota froze this container so far (None without cgroup v2 accounting)"""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("nr_throttled"):
                return int(line.split()[1])
    except OSError:
        pass
    return None


def usable_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota
    (the GPU boxes report 256 CPUs but run under a 16-core quota; 256 threads there is ~900x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def build_model(cfg, sd, device, variant="vidvrd", is_train=False):
    from models import BIG_C_vidor, BIG_C_vidvrd
    d = tempfile.mkdtemp()
    cfg = dict(cfg)
    np.save(os.path.join(d, "bias.npy"), sd["bias_matrix"].numpy())
    cfg["bias_matrix_path"] = os.path.join(d, "bias.npy")
    if "EntiNameEmb" in sd:
        np.save(os.path.join(d, "emb.npy"), sd["EntiNameEmb"].numpy())
        cfg["EntiNameEmb_path"] = os.path.join(d, "emb.npy")
    model = (BIG_C_vidvrd if variant == "vidvrd" else BIG_C_vidor)(cfg, is_train=is_train)
    model.load_state_dict(sd)
    return model.to(device)


def init_dist(args):
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, "WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("BIGC_BENCH_FORCE_DIST") == "1":      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    return rank, world, device, dist


def collect_families(lib, steps):
    out = {}
    _lib.check(lib.bigc_prof_collect())
    for f in range(lib.bigc_prof_family_count()):
        ms, n, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
        lib.bigc_prof_get(f, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
        if n.value:
            out[lib.bigc_prof_family_name(f).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out


def timed_with_events(lib, fn, warmup, steps, repeats=3):
    """(wall seconds un-instrumented, wall seconds instrumented, per-family HIP-event times) of `steps` calls of fn.  The extra legs time
    a few steps of a few milliseconds each on a box shared with other jobs: the un-instrumented loop runs `repeats` times and the MEDIAN
    repeat is reported (all of them are kept in timed_with_events.last_repeats_ms for the leg to print)."""
    for _ in range(warmup):
        fn()
    els = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        els.append(time.perf_counter() - t0)
    el = sorted(els)[len(els) // 2]
    timed_with_events.last_repeats_ms = [round(1e3 * e / steps, 3) for e in els]
    lib.bigc_prof_reset()
    lib.bigc_prof_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    el_prof = time.perf_counter() - t0
    lib.bigc_prof_enable(0)
    return el, el_prof, collect_families(lib, steps)


def gemm_roofline(fam, steps, step_ms=None):
    """roofline object of the GEMM family from its HIP-event totals (None when the family did not run).  The family's launches
    carry their algorithmic flops AND bytes (A + W + C (+ residual) once each, csrc/linear.hip); the bound is whichever of
    flops / MFMA peak and bytes / HBM peak takes longer -- MFMA for the K >= 512 GEMMs of BIG-C, HBM for the d = 128 GEMMs of
    the grounding scorer (32 flop per byte)."""
    lin = fam.get("linear_f32")
    if not lin:
        return None
    sec = lin["ms"] * 1e-3
    peak_f = F32_MFMA_PEAK_TFLOPS if EXACT_F32 else F16_MFMA_PEAK_TFLOPS / 3
    t_mfma, t_hbm = lin["flops"] / (peak_f * 1e12), lin["bytes"] / (HBM_PEAK_GBS * 1e9)
    if t_hbm > t_mfma:
        achieved = lin["bytes"] / sec / 1e9
        r = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                 gbytes_per_step=round(lin["bytes"] / steps / 1e9, 3), tflops=round(lin["flops"] / sec / 1e12, 2))
    else:
        achieved = lin["flops"] / sec / 1e12
        r = dict(bound="mfma", achieved=round(achieved, 2), peak=round(peak_f, 1), unit="TFLOP/s", frac=round(achieved / peak_f, 4))
    r.update(launches_per_step=lin["launches"] // steps, avg_launch_us=round(1e3 * lin["ms"] / lin["launches"], 2),
             gflop_per_step=round(lin["flops"] / steps / 1e9, 1))
    if step_ms:
        r["share_of_step"] = round(lin["ms"] / steps / step_ms, 3)
    return r


def family_rooflines(fam, steps, step_ms, mfma_peak_tflops=None):
    """Every kernel family of a leg against the roof that bounds it: the longer of flops / MFMA peak and bytes / HBM peak decides the bound
    (families whose launches carry neither -- index tables, sorts -- are listed with their time only).  -> (dict per family, share of
    the instrumented step that has a roofline, total GFLOP per step, total algorithmic GB per step)"""
    peak_f = (F32_MFMA_PEAK_TFLOPS if EXACT_F32 else F16_MFMA_PEAK_TFLOPS / 3) if mfma_peak_tflops is None else mfma_peak_tflops
    out, covered, flops_all, bytes_all = {}, 0.0, 0.0, 0.0
    for name, v in fam.items():
        sec = v["ms"] * 1e-3
        r = dict(ms_per_step=round(v["ms"] / steps, 4), launches_per_step=v["launches"] // steps)
        if sec > 0 and (v["flops"] > 0 or v["bytes"] > 0):
            t_mfma, t_hbm = v["flops"] / (peak_f * 1e12), v["bytes"] / (HBM_PEAK_GBS * 1e9)
            if t_hbm >= t_mfma:
                a = v["bytes"] / sec / 1e9
                r.update(bound="hbm", achieved=round(a, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(a / HBM_PEAK_GBS, 4))
            else:
                a = v["flops"] / sec / 1e12
                r.update(bound="mfma", achieved=round(a, 2), peak=round(peak_f, 1), unit="TFLOP/s", frac=round(a / peak_f, 4))
            r.update(gflop_per_step=round(v["flops"] / steps / 1e9, 2), gbytes_per_step=round(v["bytes"] / steps / 1e9, 4))
            covered += v["ms"]
            flops_all += v["flops"]
            bytes_all += v["bytes"]
        out[name] = r
    share = covered / steps / step_ms if step_ms else None
    return out, (round(share, 3) if share is not None else None), flops_all / steps / 1e9, bytes_all / steps / 1e9



def device_proposal(seed, n_trajs, feat_width, num_enti_cats, len_range, start_range, device):
    """synthetic_proposal with the (large) feature block drawn on the GPU: same structure recipe (numpy-seeded lengths,
    starts, categories, scores, boxes), features ~ N(0,1) from a seeded device generator.  For the legs that have no
    oracle comparison: 8 VidOR-sized videos are 4 GB of features, minutes of host RNG."""
    rs = np.random.RandomState(seed)
    lens = rs.randint(len_range[0], len_range[1], size=n_trajs)
    starts = rs.randint(start_range[0], start_range[1], size=n_trajs)
    duras = np.stack([starts, starts + lens - 1], axis=1).astype(np.int64)
    cat_ids = rs.randint(1, num_enti_cats, size=n_trajs).astype(np.int64)
    scores = np.sort(rs.uniform(0, 1, size=n_trajs).astype(np.float32))[::-1].copy()
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    total = int(lens.sum())
    xy = rs.uniform(0, 600, size=(total, 2))
    wh = rs.uniform(10, 110, size=(total, 2))
    boxes = torch.from_numpy(np.concatenate([xy, xy + wh], axis=1).astype(np.float32)).to(device)
    feats = torch.randn((total, feat_width), generator=gen, device=device, dtype=torch.float32)
    off = np.concatenate([[0], np.cumsum(lens)])
    return TrajProposal.from_blocks("synth_%d" % seed, int((starts + lens).max()), (1280, 720), cat_ids, scores, duras, lens, boxes, feats)



def generate_on_device(make, count, device, workers=None, window=16):
    """[make(v).to(device) for v in range(count)] with the host-side generation on a few threads (numpy's RandomState draws release the GIL)
    and at most `window` videos of features on the host at any time (the 256-video set is 6.7 GB; 8 ranks of it would not fit)."""
    from concurrent.futures import ThreadPoolExecutor
    workers = workers or max(1, min(8, usable_cores() // max(1, int(os.environ.get("WORLD_SIZE", 1)))))
    out = []
    with ThreadPoolExecutor(workers) as ex:
        for lo in range(0, count, window):
            for p in ex.map(make, range(lo, min(count, lo + window))):
                out.append(p.to(device))
    return out


def print_last(text):
    """print `text` as the LAST line of stdout: RCCL writes its version banner through C stdio, which -- on a pipe -- is flushed when the
    process exits, i.e. AFTER anything Python printed; so the process group is torn down by the caller first, C stdio is flushed, and only
    then the line goes out."""
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(text)
    sys.stdout.flush()

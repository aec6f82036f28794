#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of benchmarks/profile_bench.sh into the files kept under profiles/.

    python benchmarks/pmc_summary.py gpurun_out/prof_r01d profiles r01

* <tag>_bench_kernel_stats.csv   copy of the --kernel-trace --stats per-kernel summary
* <tag>_pmc_linear_hbm.csv       per launch of the dominant GEMM kernel in the LAST forward pass: FETCH_SIZE / WRITE_SIZE
                                 (separate --pmc passes; counter unit KiB; FETCH_SIZE doubled on gfx950 as
                                 /opt/skills/guides/MI355X_MICROARCH.md prescribes)
* <tag>_pmc_linear_sq.csv        SQ counters summed over the same launches
* pmc_traffic.json               HBM bytes per launch (read by bench.py for roofline.traffic)
"""
import csv
import json
import os
import shutil
import sys
from collections import OrderedDict, defaultdict

KERNEL = ("linear_split_kernel", "gemm_ps_kernel")      # the GEMM family: f32-operand kernel + pre-split-operand kernel


def csrc_sha16():
    """fingerprint of the kernel sources the measurement belongs to (tests/test_host_cpu.py fails when profiles/pmc_traffic.json
    was measured on other sources: a stale committed number is not evidence)"""
    import glob
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vidsgg-big_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")) + glob.glob(os.path.join(root, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def per_dispatch(path, kernel):
    """-> OrderedDict dispatch_id -> {counter: value}, and grid sizes, for kernels whose name contains `kernel`"""
    out, grid = OrderedDict(), {}
    for r in csv.DictReader(open(path)):
        if not any(k in r["Kernel_Name"] for k in ((kernel,) if isinstance(kernel, str) else kernel)):
            continue
        d = int(r["Dispatch_Id"])
        out.setdefault(d, defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
        grid[d] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    return out, grid


MARKER = "bbox_feat_kernel"        # first kernel of every BIG-C forward pass


def one_step(path):
    """(first, last) dispatch ids of one whole forward pass of the timed loop: from the second-to-last marker launch up to (not
    including) the last one"""
    marks = sorted(int(r["Dispatch_Id"]) for r in csv.DictReader(open(path)) if MARKER in r["Kernel_Name"])
    marks = sorted(set(marks))
    assert len(marks) >= 3, "need at least three forward passes in the trace"
    return marks[-3], marks[-2]


def in_step(d, span):
    return [i for i in sorted(d) if span[0] <= i < span[1]]


TRAIN_GEMM = ("linear_split_kernel", "linear_dw_split_kernel", "linear_dw_group_kernel", "linear_dw_kernel", "gemm_ps_kernel")


def train_traffic(src):
    """HBM bytes of ONE training step (bench.py --mode train --train-videos V): the GEMM kernels (forward + dX: linear_split; dW: linear_dw_*)
    per launch, and every kernel of the step -> {"videos_8": {...}, "videos_64": {...}} (absent passes are skipped)"""
    out = {}
    for v in (8, 64):
        fp = os.path.join(src, "pmc_train%d_FETCH_SIZE" % v, "pmc_counter_collection.csv")
        wp = os.path.join(src, "pmc_train%d_WRITE_SIZE" % v, "pmc_counter_collection.csv")
        if not (os.path.exists(fp) and os.path.exists(wp)):
            continue
        try:
            gf, _ = per_dispatch(fp, TRAIN_GEMM)
            gw, _ = per_dispatch(wp, TRAIN_GEMM)
            af, _ = per_dispatch(fp, "")
            aw, _ = per_dispatch(wp, "")
            sf, sw = one_step(fp), one_step(wp)
            fi, wi = in_step(gf, sf), in_step(gw, sw)
            fetch = sum(gf[i]["FETCH_SIZE"] for i in fi) * 1024 * 2
            write = sum(gw[i]["WRITE_SIZE"] for i in wi) * 1024
            step_f = sum(af[i]["FETCH_SIZE"] for i in in_step(af, sf)) * 1024 * 2
            step_w = sum(aw[i]["WRITE_SIZE"] for i in in_step(aw, sw)) * 1024
            out["videos_%d" % v] = {"gemm_launches_per_step": len(fi), "gemm_fetch_bytes_per_step": fetch, "gemm_write_bytes_per_step": write,
                                    "gemm_hbm_bytes_per_launch": (fetch + write) / max(1, len(fi)),
                                    "whole_step_fetch_bytes_all_kernels": step_f, "whole_step_write_bytes_all_kernels": step_w,
                                    "kernels": " + ".join(TRAIN_GEMM)}
        except Exception as e:                      # a missing or short trace must not lose the eval summary
            out["videos_%d" % v] = {"error": repr(e)}
    return out


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
    fetch, grid = per_dispatch(os.path.join(src, "pmc_FETCH_SIZE", "pmc_counter_collection.csv"), KERNEL)
    write, _ = per_dispatch(os.path.join(src, "pmc_WRITE_SIZE", "pmc_counter_collection.csv"), KERNEL)
    fpath, wpath = os.path.join(src, "pmc_FETCH_SIZE", "pmc_counter_collection.csv"), os.path.join(src, "pmc_WRITE_SIZE", "pmc_counter_collection.csv")
    fid, wid = in_step(fetch, one_step(fpath)), in_step(write, one_step(wpath))
    assert len(fid) == len(wid)
    tot_f = tot_w = 0.0
    with open(os.path.join(dst, tag + "_pmc_linear_hbm.csv"), "w") as f:
        f.write("launch_in_step,grid_workgroups,FETCH_SIZE_KB_raw,fetch_bytes_corrected_x2,WRITE_SIZE_KB,write_bytes\n")
        for i, (a, b) in enumerate(zip(fid, wid)):
            fk, wk = fetch[a]["FETCH_SIZE"], write[b]["WRITE_SIZE"]
            fb, wb = fk * 1024 * 2, wk * 1024
            tot_f += fb
            tot_w += wb
            f.write("%d,%d,%.1f,%d,%.1f,%d\n" % (i, grid[a], fk, fb, wk, wb))
    # every kernel of the last step (not only the GEMM family): the step's total HBM traffic
    all_f, _ = per_dispatch(os.path.join(src, "pmc_FETCH_SIZE", "pmc_counter_collection.csv"), "")
    all_w, _ = per_dispatch(os.path.join(src, "pmc_WRITE_SIZE", "pmc_counter_collection.csv"), "")
    step_f = sum(all_f[i]["FETCH_SIZE"] for i in in_step(all_f, one_step(fpath))) * 1024 * 2
    step_w = sum(all_w[i]["WRITE_SIZE"] for i in in_step(all_w, one_step(wpath))) * 1024
    # every launch of that step in order: kernel, HBM bytes fetched (corrected) and written -- where the step's traffic goes
    names = {}
    for r in csv.DictReader(open(fpath)):
        names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    fi, wi = in_step(all_f, one_step(fpath)), in_step(all_w, one_step(wpath))
    if len(fi) == len(wi):
        import re
        with open(os.path.join(dst, tag + "_pmc_step_hbm_by_launch.csv"), "w") as f:
            f.write("launch_in_step,kernel,fetch_MB_corrected_x2,write_MB\n")
            for i, (a, b) in enumerate(zip(fi, wi)):
                n = re.sub(r"\(anonymous namespace\)::", "", names.get(a, "?"))
                n = re.sub(r"\(.*", "", n).replace("void ", "")
                f.write("%d,%s,%.1f,%.1f\n" % (i, n.replace(",", ";"), all_f[a]["FETCH_SIZE"] * 2048 / 1e6, all_w[b]["WRITE_SIZE"] * 1024 / 1e6))
    json.dump({"train": train_traffic(src),
               "kernel": " + ".join(KERNEL), "launches_per_step": len(fid), "fetch_bytes_per_step": tot_f, "write_bytes_per_step": tot_w,
               "hbm_bytes_per_launch": (tot_f + tot_w) / len(fid),
               "whole_step_fetch_bytes_all_kernels": step_f, "whole_step_write_bytes_all_kernels": step_w,
               "csrc_sha16": csrc_sha16(), "tag": tag,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 2 --warmup 1 "
                         "--cpu-videos 0 --no-prof` (benchmarks/profile_bench.sh); the last step's GEMM launches; FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); counter unit KiB"},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    sums = OrderedDict()
    for sub in ("pmc_sq1", "pmc_sq2"):
        p = os.path.join(src, sub, "pmc_counter_collection.csv")
        if not os.path.exists(p):
            continue
        d, _ = per_dispatch(p, KERNEL)
        for i in in_step(d, one_step(p)):
            for k, v in d[i].items():
                sums[k] = sums.get(k, 0.0) + v
    with open(os.path.join(dst, tag + "_pmc_linear_sq.csv"), "w") as f:
        f.write("counter,sum_over_%d_gemm_launches_of_one_step\n" % len(fid))
        for k, v in sums.items():
            f.write("%s,%d\n" % (k, v))
    print(json.dumps({"whole_step_fetch_GB_all_kernels": step_f / 1e9, "whole_step_write_GB_all_kernels": step_w / 1e9,
                      "fetch_GB": tot_f / 1e9, "write_GB": tot_w / 1e9, "per_launch_MB": (tot_f + tot_w) / len(fid) / 1e6, "sq": sums}, indent=1))


if __name__ == "__main__":
    main()

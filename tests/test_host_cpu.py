"""CPU-side checks (no GPU): the C-ABI library builds/loads and exports every symbol the header
declares, the host packing logic, and the drop-in module surface."""
import ctypes
import os
import re
import sys
import tempfile

import numpy as np
import pytest
import torch

import cases
from bigc_hip import _lib
from bigc_hip.packed import PackedBatch
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal
from oracle import bigc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "bigc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:bigc|grd)_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "library does not export " + n
    # every declared function has a ctypes signature on the Python side too
    for n in names:
        assert n in _lib._SIGS, "no ctypes signature for " + n
    assert b"gfx950" in _lib.load().bigc_build_info()
    # and the other way round: the dynamic symbol table holds the header's functions and NOTHING else -- no mangled launcher, no
    # template instantiation, no data symbol (-fvisibility=hidden + csrc/exports.map; BIGC_API marks the entry points in the header)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.split())
    assert exported == names, (sorted(set(exported) - set(names)), sorted(set(names) - set(exported)))
    text = open(os.path.join(ROOT, "include", "bigc_hip.h")).read()
    assert len(re.findall(r"^BIGC_API ", text, flags=re.M)) == len(names)


def test_packed_batch_offsets_cpu():
    props = [synthetic_proposal(1, n_trajs=5, feat_width=12, len_range=(5, 9)),
             synthetic_proposal(2, n_trajs=3, feat_width=12, len_range=(20, 31))]
    pk = PackedBatch(props, "cpu")
    assert pk.n_vid == 2 and pk.n_trk == 8 and pk.max_trk == 5
    lens = [b.shape[0] for p in props for b in p.bboxes_list]
    assert pk.n_rows == sum(lens)
    assert pk.trk_row_off.tolist() == np.concatenate([[0], np.cumsum(lens)]).tolist()
    tmax = [max(b.shape[0] for b in p.bboxes_list) for p in props]
    assert pk.vid_tmax.tolist() == tmax
    t_o = [(tmax[0] - 1) // 2 + 1] * 5 + [(tmax[1] - 1) // 2 + 1] * 3
    assert pk.trk_conv_off.tolist() == np.concatenate([[0], np.cumsum(t_o)]).tolist()
    assert pk.n_conv == sum(t_o)
    assert pk.vid_trk_off.tolist() == [0, 5, 8]
    assert pk.n_pairs() == 5 * 4 + 3 * 2
    assert torch.equal(pk.feats[:lens[0]], props[0].features_list[0])
    assert pk.trk_dura.dtype == torch.int32 and pk.trk_dura.shape == (8, 2)


def test_dropin_module_surface():
    from models import BIG_C_vidvrd, BIG_C_vidor
    d = tempfile.mkdtemp()
    for variant, cls, cname in (("vidvrd", BIG_C_vidvrd, "CONFIG_V"), ("vidor", BIG_C_vidor, "CONFIG_O")):
        cfg = cases.get_cfg(cname)
        np.save(d + "/b.npy", np.zeros((cfg["num_enti_cats"], cfg["num_enti_cats"], cfg["num_pred_cats"]), np.float32))
        cfg["bias_matrix_path"] = d + "/b.npy"
        if variant == "vidvrd":
            np.save(d + "/e.npy", np.zeros((cfg["num_enti_cats"], cfg["dim_clsme"]), np.float32))
            cfg["EntiNameEmb_path"] = d + "/e.npy"
        m = cls(cfg, is_train=False)
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == dict(bigc_state_shapes(cfg, variant))
        if variant == "vidvrd":
            assert sum(v.numel() for v in m.state_dict().values()) == 27776445
        # checkpoints saved from the reference's DataParallel wrapper carry a "module." prefix
        sd = {"module." + k: v for k, v in m.state_dict().items()}
        m.load_state_dict({k[7:]: v for k, v in sd.items()})
        # empty videos never reach the device
        empty = synthetic_proposal(3, n_trajs=0)
        assert m([empty, empty]) == [None, None]


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libbigc_hip.so")
    with pytest.raises(_lib.BigcError):
        _lib.load()


def test_oracle_batch_independence():
    """the oracle treats videos independently: a list call equals per-video calls"""
    cfg, variant, sd, props, topk = cases.make_case("bigc_tiny_edges")
    both = O.forward_test(sd, cfg, props, topk, variant)
    for p, r in zip(props, both):
        one = O.forward_test(sd, cfg, [p], topk, variant)[0]
        assert (one is None) == (r is None)
        if r is not None:
            assert all(torch.equal(a, b) for a, b in zip(one, r))


def test_adj_from_viou_matches_oracle_alignment():
    """host half of the GT alignment (thresholding the vIoU table) against the oracle's restatement of
    enti_viou_align, fed with the oracle's own vIoU table"""
    import copy
    from bigc_hip.train_step import adj_from_viou
    from oracle import bigc_train_oracle as T
    for name in ("bigc_train_tiny", "bigo_train_tiny"):
        cfg, variant, sd, props, gts = cases.make_train_case(name)
        for p, g in zip(props, gts):
            want, vi = T.align_gt_to_proposals(p, copy.deepcopy(g), cfg["positive_vIoU_th"])
            got = adj_from_viou(vi.clone(), g.adj_matrix.clone(), cfg["positive_vIoU_th"])
            assert torch.equal(got, want)
            assert want.sum() > 0


def test_pair_label_assignment_matches_the_reference_loop():
    """labels_from_viou vs the literal double loop of prop_pair_to_gt_pred (tools/train_vidor.py:131-158) on random vIoU tables"""
    from collections import defaultdict
    from bigc_hip.pair_train_step import labels_from_viou
    rs = np.random.RandomState(3)
    for trial in range(20):
        n_prop, n_gt, n_pred, P, th = rs.randint(2, 9), rs.randint(1, 5), rs.randint(1, 7), 11, 0.5
        vi = torch.from_numpy(rs.uniform(0, 1, size=(n_prop, n_gt)).astype(np.float32))
        if trial == 0:
            vi[:] = 0.1                                        # nothing hit
        cats = rs.randint(1, P, size=n_pred)
        so = np.stack([rs.randint(0, n_gt, n_pred), rs.randint(0, n_gt, n_pred)], 1)
        want = defaultdict(list)
        for g in range(n_pred):
            for s_ in range(n_prop):
                for o_ in range(n_prop):
                    if s_ != o_ and vi[s_, so[g, 0]] > th and vi[o_, so[g, 1]] > th:
                        want[(s_, o_)].append(int(cats[g]))
        ids, multihot = labels_from_viou(vi, cats.tolist(), so, th, P)
        if not want:
            assert ids is None and multihot is None
            continue
        assert [tuple(r) for r in ids.tolist()] == list(want.keys())          # the reference's dict insertion order
        ref = torch.zeros(len(want), P)
        for i, c in enumerate(want.values()):
            ref[i, c] = 1
        assert torch.equal(multihot, ref)


def test_host_side_under_address_and_ub_sanitizers():
    """The library's host code (argument checks, parameter table, finalize's resolver, the workspace arena layout, mode switches,
    profiling table) compiled host-only with -fsanitize=address,undefined and driven by tests/native/host_sanitize.cpp -- no GPU,
    no device code: any heap misuse, leak or undefined integer arithmetic in those paths fails the run."""
    import subprocess
    from bigc_hip.engine import BigcEngine
    csrc = os.path.join(ROOT, "vidsgg-big_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asan", "-j8"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PATH="/opt/rocm/bin:" + os.environ.get("PATH", "")))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = []
    for variant, cname in (("vidvrd", "CONFIG_V"), ("vidor", "CONFIG_O"), ("vidvrd", "CONFIG_TINY")):
        cfg = cases.get_cfg(cname)
        eng = BigcEngine(cfg, variant)
        lines.append("model " + " ".join(str(int(getattr(eng.c, f))) for f, _ in _lib.bigc_config._fields_))
        for k, shape in bigc_state_shapes(cfg, variant):
            lines.append("param %s %d" % (k, int(np.prod(shape))))
        # (videos, tracklets, rows, conv rows, max tracklets per video): one video, the benchmark step, a VidOR-sized batch
        for b in ((1, 1, 5, 3, 1), (8, 300, 9000, 5000, 64), (256, 16384, 579295, 519168, 64), (16, 2880, 1456366, 1300000, 180)):
            lines.append("batch %d %d %d %d %d" % b)
    spec = os.path.join(tempfile.mkdtemp(), "spec.txt")
    open(spec, "w").write("\n".join(lines) + "\n")
    exe = os.path.join(ROOT, "vidsgg-big_amd", "lib", "host_sanitize")
    r = subprocess.run([exe, spec], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"))
    assert r.returncode == 0 and "host_sanitize: OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_committed_pmc_traffic_belongs_to_the_current_kernel_sources():
    """profiles/pmc_traffic.json (read by bench.py for roofline.traffic: HBM counters cannot be sampled from inside the process) carries
    the fingerprint of the kernel sources it was measured on: a number measured on other kernels is not evidence, bench.py then reports
    null -- and this test fails until benchmarks/profile_bench.sh has been re-run on the GPU box"""
    import json
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    from pmc_summary import csrc_sha16
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert pmc.get("csrc_sha16") == csrc_sha16(), "profiles/pmc_traffic.json is stale: re-run `TAG=r0X bash benchmarks/profile_bench.sh` on the GPU box"
    assert pmc["hbm_bytes_per_launch"] > 0 and pmc["whole_step_fetch_bytes_all_kernels"] > 0


def test_video_index_loads_only_the_ranks_share():
    """tools/_common.VideoIndex: the assignment of videos to ranks is computed from costs known WITHOUT loading (synthetic: the row
    count replayed from the seed), and a rank loads exactly its share (the reference iterates a DataLoader, tools/eval_vidvrd.py:91-121)"""
    sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd", "tools"))
    import _common as C
    cfg = cases.get_cfg("CONFIG_TINY")
    world, n = 3, 10
    shares = []
    for rank in range(world):
        idx = C.synthetic_index(cfg, n, 6, seed=77)
        assert len(idx) == n and idx.n_loaded == 0
        mine, items = C.shard_for_inference(idx, idx.costs, rank, world)
        assert idx.n_loaded == len(mine) == len(items)
        for i, (p, g) in zip(mine, items):
            assert p.video_name == idx.names[i] and sum(int(b.shape[0]) for b in p.bboxes_list) == idx.costs[i]     # the cost IS the row count
        shares.append(mine)
    assert sorted(i for s in shares for i in s) == list(range(n))
    loads = [sum(C.synthetic_index(cfg, n, 6, seed=77).costs[i] for i in s) for s in shares]
    assert max(loads) - min(loads) <= max(C.synthetic_index(cfg, n, 6, seed=77).costs)          # balanced to within one video
    # a bounded cache (training): old items are dropped, reloading works
    idx = C.synthetic_index(cfg, n, 6, seed=77, keep=2)
    for i in (0, 1, 2, 0):
        idx[i]
    assert idx.n_loaded == 4 and len(idx._cache) == 2


def test_block_backed_proposal_keeps_the_list_contract():
    """bigc_hip.synthetic.TrajProposal keeps a video's tracklets as views of two blocks (made on demand); the reference's list contract
    (dataloaders/dataloader_vidvrd.py:14-77) must still hold: reading, element assignment (what the reference's own .to() does), list
    assignment, in-place writes through a view, deepcopy, and .to() back into blocks"""
    import copy
    p = synthetic_proposal(3, n_trajs=6, feat_width=12, len_range=(4, 9)).to("cpu")
    assert p._lists is None and len(p.features_list) == 6 and p.features_list[0].shape[1] == 12
    want = [f.clone() for f in p.features_list]
    p.features_list[1][0, 0] = 7.0                                   # a write through a view lands in the block
    want[1][0, 0] = 7.0
    assert torch.equal(p.features_list[1], want[1]) and p._lists is None
    q = copy.deepcopy(p)
    pk = PackedBatch([q], "cpu")                                     # (the copied address table is stale: rebuilt from the copy's own tensors)
    assert torch.equal(pk.feats, torch.cat(want, 0))
    x = torch.full_like(want[4], 3.0)
    p.features_list[4] = x                                           # element assignment: the container now owns explicit lists
    want[4] = x
    assert p._lists is not None and not hasattr(p, "_bigc_pack")
    assert all(torch.equal(a, b) for a, b in zip(p.features_list, want))
    assert torch.equal(PackedBatch([p], "cpu").feats, torch.cat(want, 0))
    p.to("cpu")                                                      # back into blocks
    assert p._lists is None and all(torch.equal(a, b) for a, b in zip(p.features_list, want))
    p.bboxes_list = [b + 1.0 for b in p.bboxes_list]                 # list assignment
    assert p._lists is not None and torch.equal(PackedBatch([p], "cpu").boxes, torch.cat(p.bboxes_list, 0))


def test_the_restated_assignment_algorithm_is_scipys():
    """oracle/lsap.py restates csrc/loss.hip: hungarian_kernel (Crouse's shortest augmenting paths with scipy's choice among equal
    candidates written as an order-free rule); scipy.optimize.linear_sum_assignment is what the reference calls
    (models/model_0v10.py:637).  Equal assignments also where the optimum is not unique; the same inputs rejected."""
    from scipy.optimize import linear_sum_assignment
    from oracle import lsap
    rs = np.random.RandomState(0)
    checked = rejected = 0
    for trial in range(250):
        nq = int(rs.choice([5, 8, 17, 32, 48]))
        g = int(rs.randint(1, nq))
        kind = trial % 5
        if kind == 0:
            c = rs.rand(nq, g)
        elif kind == 1:
            c = rs.randint(0, 3, (nq, g)).astype(float)
        elif kind == 2:
            c = rs.rand(nq, g)[:, rs.randint(0, g, g)]                 # duplicated GT predicates
        elif kind == 3:
            c = rs.randint(0, 4, (nq, g)).astype(float)
            c[rs.rand(nq, g) < (0.3 if trial % 2 else 0.85)] = np.inf          # (0.85: some of these have no assignment at all)
        else:
            c = np.zeros((nq, g))
        try:
            iq, ig = linear_sum_assignment(c)
        except ValueError:
            with pytest.raises(ValueError):
                lsap.assign(c)
            rejected += 1
            continue
        want = -np.ones(nq, dtype=np.int64)
        want[iq] = ig
        assert np.array_equal(lsap.assign(c), want), (trial, kind)
        checked += 1
    assert checked > 180 and rejected > 0, (checked, rejected)
    for bad in (np.nan, -np.inf):
        c = rs.rand(6, 3)
        c[1, 1] = bad
        with pytest.raises(ValueError):
            lsap.assign(c)


def test_device_matching_is_only_chosen_where_it_reproduces_scipy(monkeypatch):
    """bigc_hungarian restates scipy for FEWER GT predicates than queries (scipy then works on the transposed matrix and every predicate
    is matched, so the loss normalisers are known without the assignment); anything else -- or BIGC_MATCH_ON_HOST=1 -- stays on scipy"""
    from types import SimpleNamespace as NS
    from bigc_hip import train_step as TS

    def g(n_pred, n_traj):
        return NS(adj_matrix=torch.zeros(2, n_pred, n_traj), traj_bboxes=[torch.zeros(3, 4)] * n_traj)
    monkeypatch.delenv("BIGC_MATCH_ON_HOST", raising=False)
    assert TS.match_on_device([g(4, 5), g(11, 2)], 12)
    assert not TS.match_on_device([g(4, 5), g(12, 2)], 12)          # as many predicates as queries: scipy does not transpose
    assert not TS.match_on_device([g(4, 5), g(40, 2)], 12)
    assert not TS.match_on_device([g(4, 0)], 12)                    # no GT trajectory: the alignment has nothing to arg-max over
    assert not TS.match_on_device([g(0, 3)], 12)
    # what the device kernels refuse takes the host route too (alignment: 8192 proposals per video in LDS; assignment: 1024 queries)
    assert TS.match_on_device([g(4, 5)], 1024, max_trk=8192) and not TS.match_on_device([g(4, 5)], 1025) and not TS.match_on_device([g(4, 5)], 12, max_trk=8193)
    monkeypatch.setenv("BIGC_MATCH_ON_HOST", "1")
    assert not TS.match_on_device([g(4, 5)], 12)

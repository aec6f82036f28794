"""Randomised differential test of the BIG-C eval forward: HIP path (through the C ABI) vs the CPU oracle on seeded random
CONFIGURATIONS -- widths of 32 .. 128 (so that the production kernels run: chained split-operand GEMMs, fused role attention
where a video has <= 64 tracklets, the unfused branch above), 1-2 encoder / decoder layers, 2-4 heads, 8-48 queries, with / without
I3D, the three VidOR head variants -- and random ragged batches (1 .. 90 tracklets of 5 .. 70 frames, empty videos, disjoint
tracklets).  Contract as everywhere: triplet rows, durations and query ids identical, logits within 1e-4 relative."""
import numpy as np
import pytest
import torch

import cases
from bigc_hip.packed import PackedBatch
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict
from oracle import bigc_oracle as O
from test_hip_bigc import LOGIT_RTOL, build_model, check_tuple_exact, rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def random_case(seed):
    rs = np.random.RandomState(9000 + seed)
    variant = "vidvrd" if seed % 2 == 0 else "vidor"
    width = int(rs.choice([32, 64, 128]))
    heads = int(rs.choice([h for h in (2, 4, 8) if width // h in (8, 16, 32, 64)]))
    cfg = dict(num_enti_cats=int(rs.randint(3, 40)), num_pred_cats=int(rs.randint(3, 60)), dim_ffn=int(rs.choice([32, 64, 128])),
               dim_enti=width, dim_pred=width, dim_att=int(rs.choice([32, 64, 128])), dim_feat=int(rs.choice([32, 64, 96, 160])),
               dim_clsme=int(rs.choice([12, 20, 32])), enco_pool_len=4, n_enco_layers=int(rs.randint(1, 3)), n_deco_layers=int(rs.randint(1, 3)),
               n_att_head=heads, num_querys=int(rs.choice([8, 12, 20, 33, 48])), neg_weight=0.1, positive_vIoU_th=0.5,
               cost_coeff_dict=dict(classification=1.0, adj_matrix=30.0), loss_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
               EntiNameEmb_path=None, bias_matrix_path=None)
    if variant == "vidvrd":
        if rs.rand() < 0.7:
            cfg["dim_i3d"] = int(rs.choice([16, 32, 48]))
    else:
        mode = int(rs.randint(0, 3))
        cfg["use_clsme"] = mode != 2
        if mode == 1:
            cfg["_with_EntiNameEmb"] = True
    n_vid = int(rs.randint(1, 6))
    specs = []
    for v in range(n_vid):
        kind = rs.rand()
        if kind < 0.1:
            specs.append(dict(seed=int(rs.randint(1 << 20)), n_trajs=0))
            continue
        n = int(rs.choice([1, 2, 5, 17, 40, 64, 65, 90]))
        lo = int(rs.randint(5, 30))
        specs.append(dict(seed=int(rs.randint(1 << 20)), n_trajs=n, len_range=(lo, lo + int(rs.randint(1, 40))), start_range=(0, int(rs.randint(1, 40))),
                          disjoint=bool(kind > 0.93)))
    return variant, cfg, specs, int(rs.randint(1, 6)), float(rs.choice([1.0, 4.0, 8.0]))


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_configuration_vs_oracle(seed):
    variant, cfg, specs, topk, gain = random_case(seed)
    topk = min(topk, cfg["num_pred_cats"] - 1)
    sd = seeded_state_dict(bigc_state_shapes(cfg, variant), 100 + seed, att_gain=gain)
    props = [cases.make_proposal(cfg, variant, s) for s in specs]
    model = build_model(cfg, variant, sd)
    got = model(props, topk=topk)
    want = O.forward_test(sd, cfg, props, topk=topk, variant=variant)
    assert len(got) == len(want) == len(props)
    for gv, wv in zip(got, want):
        if wv is None:
            assert gv is None
        else:
            check_tuple_exact(gv, *[t.numpy() for t in wv])
    live = [p for p in props if p.num_proposals > 0]
    if not live:
        return
    out = model.forward_packed(PackedBatch(live, DEV), topk)
    logits = out["pred_logits"].cpu().numpy().reshape(len(live), cfg["num_querys"], -1)
    soid = out["soid"].cpu().numpy()
    for j, p in enumerate(live):
        o = O.encode2decode(sd, cfg, p, variant)
        assert rel(logits[j], o["pred_logits"].numpy()) < LOGIT_RTOL, (seed, j)
        assert np.array_equal(soid[j], o["att"].argmax(-1).numpy()), (seed, j)

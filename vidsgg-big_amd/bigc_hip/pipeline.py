"""Classification-then-grounding glue: how the two stages meet (reference: ``evaluate_combined``,
tools/eval_vidor.py:218-256).  Host-side tensor bookkeeping only; the two stages themselves are the HIP models."""
import torch


def combine_with_grounding(cls_result, grounding_out, video_len):
    """Expand every classified triplet by its kept temporal bins (tools/eval_vidor.py:245-253).

    ``cls_result`` = the 4-tuple of ``BIG_C.forward`` for the video, ``grounding_out`` = ``(pooled_se (n,k+1,2),
    bins_probs (n,k+1), bins_mask (n,k+1) bool)`` of ``DEBUG.forward``.  Returns the 3-tuple the evaluator consumes:
    quintuples (m,5) i64, scores (m,) f32 = mean(cls scores) * bin probability, durations (m,2) i64 = grounded spans
    scaled to frames and rounded (closed)."""
    quint, scores, _dura, _qid = cls_result
    pooled, probs, mask = grounding_out
    n, nb = probs.shape
    base = scores.mean(dim=-1) if scores.dim() == 2 else scores
    mask_c = mask.cpu()
    q = quint.cpu()[:, None, :].repeat(1, nb, 1)[mask_c, :]
    s = (base.to(probs.device)[:, None] * probs).cpu()[mask_c]
    d = torch.round((pooled * video_len).cpu()[mask_c, :]).type(torch.long)
    return q, s, d


def classify_then_ground(bigc_model, grd_model, proposal, clip_features, topk=3, score_th=0.9, tiou_th=0.5, bins_th=0.2, nms_th=0.8,
                         max_triplets=None):
    """Both stages for one video. ``max_triplets`` keeps the highest-scoring classified triplets only (the reference's
    ``rt_triplets_topk`` cut for the grounding stage, experiments/exp6/config_rt200.py:14)."""
    res = bigc_model([proposal], topk=topk)[0]
    if res is None:
        return None
    if max_triplets is not None and res[0].shape[0] > max_triplets:
        keep = torch.argsort(res[1].mean(dim=-1), descending=True)[:max_triplets]
        keep = torch.sort(keep)[0]
        res = tuple(t[keep] for t in res)
    g = grd_model([clip_features], [(res[0], res[2], proposal.video_len)], score_th, tiou_th, bins_th, nms_th, with_gt_data=False)
    if g[0] is None:
        return None
    return combine_with_grounding(res, g, proposal.video_len)

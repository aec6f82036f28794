# VidOR pairwise baseline Base_C: the values of the reference's experiments/exp6/config_.py (config_rt200.py only differs by
# rt_triplets_topk = 200, the cut applied before the grounding stage); dataset locations are site specific.
model_config = dict(
    dataset_type="VidOR", num_enti_cats=81, num_pred_cats=51, dim_ffn=512, dim_enti=512, dim_pred=512, dim_att=512, dim_feat=1024,
    dim_clsme=300, enco_pool_len=4, positive_vIoU_th=0.5, rt_triplets_topk=-1, EntiNameEmb_path=None, use_clsme=True,
    bias_matrix_path="prepared_data/pred_bias_matrix_vidor.npy",
)
test_dataset_config = dict(split="val", video_dir="datasets/vidor-dataset/val_videos", ann_dir="datasets/vidor-dataset/annotation",
                           proposal_dir="proposals/miss60_minscore0p3/VidORval_freq1",
                           classeme_dir="proposals/miss60_minscore0p3/VidORval_freq1_classeme",
                           max_proposal=180, max_preds=200, score_th=0.4, dim_boxfeature=1024, min_frames_th=15, cache_tag="MEGAv9_m60s0.3_freq1")
train_dataset_config = dict(split="train", ann_dir="datasets/vidor-dataset/annotation", video_dir="datasets/vidor-dataset/train_videos",
                            classeme_dir="proposals/miss60_minscore0p3/VidORtrain_freq1_classeme",
                            proposal_dir={i: "proposals/miss60_minscore0p3/VidORtrain_freq1_part%02d" % (i + 1) for i in range(14)},
                            cache_dir="datasets/cache", cache_tag="MEGAv7", dim_boxfeature=1024, min_frames_th=15, max_proposal=180,
                            max_preds=200, score_th=0.4)
train_config = dict(batch_size=4, total_epoch=80, initial_lr=5e-5, lr_decay=0.2, epoch_lr_milestones=[50])
inference_config = dict(topk=3)

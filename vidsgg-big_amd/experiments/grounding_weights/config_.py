# Temporal grounding stage (VidOR): the values of the reference's experiments/grounding_weights/config_.py:2-80
# (dataset locations are site specific and left to the user).
model_config = dict(
    dim_feat=1024, dim_clsme=300, dim_hidden=128, num_bins=10,
    EntiNameEmb_path="prepared_data/vidor_EntiNameEmb.npy", PredNameEmb_path="prepared_data/vidor_PredNameEmb.npy",
    loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0),
)
test_dataset_config = dict(split="val", video_dir="datasets/vidor-dataset/val_videos", ann_dir="datasets/vidor-dataset/annotation",
                           proposal_dir="proposals/miss60_minscore0p3/VidORval_freq1",
                           classeme_dir="proposals/miss60_minscore0p3/VidORval_freq1_classeme",
                           video_feature_dir="I3D_clip_features/clip16_overlap0.5_val",
                           max_proposal=180, max_preds=200, score_th=0.4, dim_boxfeature=1024, min_frames_th=15, cache_tag="MEGAv9_m60s0.3_freq1")
train_dataset_config = dict(split="train", ann_dir="datasets/vidor-dataset/annotation", video_dir="datasets/vidor-dataset/train_videos",
                            classeme_dir="proposals/miss60_minscore0p3/VidORtrain_freq1_classeme",
                            video_feature_dir="I3D_clip_features/clip16_overlap0.5",
                            proposal_dir={i: "proposals/miss60_minscore0p3/VidORtrain_freq1_part%02d" % (i + 1) for i in range(14)},
                            cache_dir="datasets/cache", cache_tag="MEGAv7", dim_boxfeature=1024, min_frames_th=15, max_proposal=180,
                            max_preds=200, score_th=0.4)
train_config = dict(batch_size=8, total_epoch=80, initial_lr=5e-5, lr_decay=0.2, epoch_lr_milestones=[40, 60])
inference_config = dict(score_th=0.9, tiou_th=0.5, bins_th=0.2, nms_th=0.8, eval_tiouths=0.5)

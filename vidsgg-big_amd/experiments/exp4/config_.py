# VidOR BIG-C WITHOUT classemes (use_clsme = False, RoI 1024-d only): the values of the reference's experiments/exp4/config_.py:3-95
# (dataset locations are site specific and left to the user).
model_config = dict(
    dataset_type="VidOR", num_enti_cats=81, num_pred_cats=51, dim_ffn=512, dim_enti=512, dim_pred=512, dim_att=512, dim_feat=1024,
    dim_clsme=300, enco_pool_len=4, n_enco_layers=6, n_deco_layers=4, n_att_head=8, num_querys=192, neg_weight=0.1, positive_vIoU_th=0.5,
    EntiNameEmb_path=None, use_clsme=False, bias_matrix_path="prepared_data/pred_bias_matrix_vidor.npy",
    cost_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
    loss_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
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
inference_config = dict(use_post=True, topk=3, use_regr=True)
extra_config = dict(dataloader_name="dataloader_vidor")

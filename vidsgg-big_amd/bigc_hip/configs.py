"""Model configurations of the BIG hot path.

The dictionaries carry the same keys the reference reads in its constructors
(reference: models/model_0v10.py:247-276, models/model_0v7.py:247-283,
models/grd_model_v5.py:148-156); the values are those of the reference's
experiment configs (experiments/exp2/config_.py:2-31 = "config V",
experiments/exp5/config_.py:3-31 = "config O",
experiments/grounding_weights/config_.py:2-14 = "config G").
The two ``*_path`` entries are filled in by whoever builds a model
(they point at ``.npy`` files exactly as in the reference).
"""

CONFIG_V = dict(
    num_enti_cats=36,
    num_pred_cats=133,
    dim_ffn=512,
    dim_enti=512,
    dim_pred=512,
    dim_att=512,
    dim_feat=2048,
    dim_clsme=300,
    dim_i3d=832,
    enco_pool_len=4,
    n_enco_layers=2,
    n_deco_layers=6,
    n_att_head=8,
    num_querys=192,
    neg_weight=0.1,
    positive_vIoU_th=0.5,
    EntiNameEmb_path=None,
    bias_matrix_path=None,
    cost_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
    loss_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
)

# VidVRD exp1: same as V without I3D (experiments/exp1/config_.py)
CONFIG_V_NOI3D = dict(CONFIG_V)
CONFIG_V_NOI3D.pop("dim_i3d")

# A tiny configuration with the same structure, used by parity tests where every
# intermediate of the path is compared (fixture G1).
CONFIG_TINY = dict(
    num_enti_cats=7,
    num_pred_cats=9,
    dim_ffn=16,
    dim_enti=16,
    dim_pred=16,
    dim_att=16,
    dim_feat=32,
    dim_clsme=12,
    dim_i3d=16,
    enco_pool_len=4,
    n_enco_layers=2,
    n_deco_layers=2,
    n_att_head=2,
    num_querys=12,
    neg_weight=0.1,
    positive_vIoU_th=0.5,
    EntiNameEmb_path=None,
    bias_matrix_path=None,
    cost_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
    loss_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
)

CONFIG_O = dict(
    num_enti_cats=81,
    num_pred_cats=51,
    dim_ffn=512,
    dim_enti=512,
    dim_pred=512,
    dim_att=512,
    dim_feat=1024,
    dim_clsme=300,
    enco_pool_len=4,
    n_enco_layers=6,
    n_deco_layers=4,
    n_att_head=8,
    num_querys=192,
    neg_weight=0.1,
    positive_vIoU_th=0.5,
    use_clsme=True,
    EntiNameEmb_path=None,
    bias_matrix_path=None,
    cost_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
    loss_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
)

# tiny VidOR-structured configuration (sine pos-embedding, classeme in the features, 2-layer head)
CONFIG_TINY_O = dict(
    num_enti_cats=9,
    num_pred_cats=7,
    dim_ffn=32,
    dim_enti=16,
    dim_pred=16,
    dim_att=16,
    dim_feat=24,
    dim_clsme=12,
    enco_pool_len=4,
    n_enco_layers=3,
    n_deco_layers=2,
    n_att_head=2,
    num_querys=10,
    neg_weight=0.1,
    positive_vIoU_th=0.5,
    use_clsme=True,
    EntiNameEmb_path=None,
    bias_matrix_path=None,
    cost_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
    loss_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
)

CONFIG_G = dict(
    dim_feat=1024,
    dim_clsme=300,
    dim_hidden=128,
    num_bins=10,
    EntiNameEmb_path=None,
    PredNameEmb_path=None,
    loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0),
)

"""Command-line flags: same names, types and defaults as the reference's
`utility.parser.parse_args()` (/root/reference/MMSSL/utility/parser.py:3-102) so existing
launch lines keep working. Table-driven; `parse_args(argv=None)` also accepts an explicit
argv (the reference only reads sys.argv, at import time, in four modules).

Live flags on the hot path: embed_size weight_size layers drop_rate model_cat_rate id_cat_rate
head_num tau cl_rate regs feat_reg_decay batch_size lr sparse seed T m_topk_rate Ks.
"""
import argparse

_HOME = "/home/ww/Code/work"
_LOAD = (_HOME + "3/BSTRec/Model/retailrocket/for_meta_hidden_dim_dim__8_retailrocket_2021_07_10__18_35_32"
         "_lr_0.0003_reg_0.01_batch_size_1024_gnn_layer_[16,16,16].pth")

# (flag, type, default); type None = string-valued flag declared with nargs='?'
_FLAGS = [
    # not read by the training path
    ("verbose", int, 5), ("core", int, 5), ("lambda_coeff", float, 0.9),
    ("early_stopping_patience", int, 7), ("layers", int, 1), ("mess_dropout", None, "[0.1, 0.1]"),
    ("sparse", int, 1), ("test_flag", None, "part"), ("metapath_threshold", int, 2), ("sc", float, 1.0),
    ("ssl_c_rate", float, 1.3), ("ssl_s_rate", float, 0.8), ("g_rate", float, 0.000029),
    ("sample_num", int, 1), ("sample_num_neg", int, 1), ("sample_num_ii", int, 8), ("sample_num_co", int, 2),
    ("mask_rate", float, 0.75), ("gss_rate", float, 0.85), ("anchor_rate", float, 0.75),
    ("feat_reg_decay", float, 1e-5), ("ad1_rate", float, 0.2), ("ad2_rate", float, 0.2),
    ("ad_sampNum", int, 1), ("ad_topk_multi_num", int, 100), ("fake_gene_rate", float, 0.0001),
    ("ID_layers", int, 1), ("reward_rate", float, 1), ("G_embed_size", int, 64), ("model_num", float, 2),
    ("negrate", float, 0.01), ("cis", int, 25), ("confidence", float, 0.5), ("ii_it", int, 15),
    ("isload", bool, False), ("isJustTest", bool, False), ("loadModelPath", str, _LOAD),
    ("title", str, "try_to_draw_line"),
    # train
    ("data_path", None, _HOME + "5/MMSSL/data/"), ("seed", int, 2022), ("dataset", None, ""),
    ("epoch", int, 1000), ("batch_size", int, 1024), ("embed_size", int, 64), ("D_lr", float, 3e-4),
    ("topk", int, 10), ("cf_model", None, "slmrec"), ("cl_rate", float, 0.03), ("norm_type", None, "sym"),
    ("gpu_id", int, 0), ("Ks", None, "[10, 20, 50]"), ("regs", None, "[1e-5,1e-5,1e-2]"),
    ("lr", float, 0.00055), ("emm", float, 1e-3), ("L2_alpha", float, 1e-3), ("weight_decay", float, 1e-4),
    # GNN
    ("drop_rate", float, 0.2), ("model_cat_rate", float, 0.55), ("gnn_cat_rate", float, 0.55),
    ("id_cat_rate", float, 0.36), ("id_cat_rate1", float, 0.36), ("head_num", int, 4), ("dgl_nei_num", int, 8),
    # GAN
    ("weight_size", None, "[64, 64]"), ("G_rate", float, 0.0001), ("G_drop1", float, 0.31),
    ("G_drop2", float, 0.5), ("gp_rate", float, 1), ("real_data_tau", float, 0.005), ("ui_pre_scale", int, 100),
    # contrastive
    ("T", int, 1), ("tau", float, 0.5), ("geneGraph_rate", float, 0.1), ("geneGraph_rate_pos", float, 2),
    ("geneGraph_rate_neg", float, -1), ("m_topk_rate", float, 0.0001), ("log_log_scale", int, 0.00001),
    ("point", str, ""),
]


def build_parser():
    p = argparse.ArgumentParser(description="")
    for name, typ, default in _FLAGS:
        if typ is None:
            p.add_argument("--" + name, nargs="?", default=default)
        else:
            p.add_argument("--" + name, type=typ, default=default)
    p.add_argument("--debug", action="store_true")
    return p


def parse_args(argv=None):
    """argparse.Namespace with the reference's flag set. argv=None reads sys.argv like the
    reference; pass a list (e.g. []) to get defaults without touching the process argv."""
    return build_parser().parse_args(argv)

"""Configuration dictionaries in the reference's own schema.

The reference reads a YAML file into nested dictionaries (utils/config_utils.py:15-54) and the
model code indexes them directly (model/dino_pqgo.py:21-67, wrapper/PQGOWrapper.py:23-35,
model/dino/dino_featurizer.py:14-38).  These helpers build the same nested dictionaries for the
two benchmark shapes so tests, the bench and the golden-vector generator agree on one definition.
"""
from copy import deepcopy

__all__ = ["pqgo_config", "S_CFG", "B_CFG", "VIT_ARCH"]

# model_type -> (embed_dim, depth, heads)   (model/dino/vision_transformer.py:261-279)
VIT_ARCH = {
    "vit_tiny": (192, 12, 3),
    "vit_small": (384, 12, 6),
    "vit_base": (768, 12, 12),
}

_STEGO = dict(  # config/pqgo_baseline.yaml:68-122 (cocostuff27, vit-small)
    neg_inter_weight=0.63, pos_inter_weight=0.25, pos_intra_weight=0.67,
    neg_inter_shift=0.66, pos_inter_shift=0.02, pos_intra_shift=0.08,
    zero_clamp=True, pointwise=True, stabilize=False, feature_samples=11, neg_samples=5,
)


def pqgo_config(model_type="vit_small", patch_size=8, embed_dim=1024, num_pq=8, num_codebook=256,
                num_classes=27, extra_classes=0, normalize="l2", need_initialized="uni",
                pretrained_weights=None, output_type="vq0", drop_prob=0.1, beta=0.25, book=1.0,
                jsd_temperature=0.1):
    """A config/pqgo_baseline.yaml-shaped dictionary (only the keys the model path reads)."""
    return deepcopy({
        "num_classes": num_classes,
        "dataset_name": "cocostuff27",
        "wandb": {"name": "PQGO_synthetic"},
        "model": {
            "pretrained": {
                "model_type": model_type,
                "dino_patch_size": patch_size,
                "freeze_backbone": True,
                "dropout": True,
                "drop_prob": drop_prob,
                "pretrained_weights": pretrained_weights,
            },
            "vq": {
                "vq_type": "param",
                "num_codebooks": [num_codebook],
                "embed_dims": [embed_dim],
                "beta": beta,
                "book": book,
                "normalize": normalize,
                "use_restart": False,
                "use_split": False,
                "use_weighted_sum": False,
                "use_gumbel": False,
                "need_initialized": need_initialized,
                "pq_dropout": 0.0,
                "decay": 0.99,
                "eps": 1.0e-6,
                "num_pq": [num_pq],
                "agg_type": "concat",
            },
        },
        "loss": {
            "stego_weight": 1.0,
            "vq_weight": 1.0,
            "mse_weight": 0.0,
            "jsd": {"num_query": 7, "num_pos": 10, "temperature": jsd_temperature},
            "stego": dict(_STEGO),
        },
        "eval": {"output_type": output_type, "extra_classes": extra_classes},
    })


# ViT-S/8, D=1024, 8 sub-spaces x 256 codewords: the 320x320 benchmark shape.
S_CFG = pqgo_config("vit_small", 8, 1024, 8, 256)
# ViT-B/8, D=2048, 16 sub-spaces x 512 codewords: the 768x768 crop shape.
B_CFG = pqgo_config("vit_base", 8, 2048, 16, 512)

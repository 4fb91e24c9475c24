"""PQGOWrapper (wrapper/PQGOWrapper.py:12-103): loss assembly + probes around DIONPQGO, with the
reference's constructor, attributes (``model``, ``evaluator``), forward signature and 4-tuple."""
from typing import Dict, Tuple

import torch
import torch.nn as nn

from .evaluator import UnSegEvaluator
from .pqgo import DIONPQGO

__all__ = ["PQGOWrapper", "build_model"]


class PQGOWrapper(nn.Module):
    def __init__(self, cfg, model: DIONPQGO) -> None:
        super().__init__()
        self.model = model
        self.num_classes = cfg["num_classes"]
        self.extra_classes = cfg["eval"]["extra_classes"]
        self.num_pq = cfg["model"]["vq"]["num_pq"][0]
        self.dataset = cfg["dataset_name"]
        self.stego_weight = cfg["loss"]["stego_weight"]
        self.recon_weight = cfg["loss"].get("recon_weight", 0.0)
        self.cls_weight = cfg["loss"].get("cls_weight", 0.0)
        self.mse_weight = cfg["loss"].get("mse_weight", 0.0)
        self.vq_weight = cfg["loss"]["vq_weight"]
        self.output_type = cfg["eval"]["output_type"]
        if self.output_type == "feat":
            output_dim = cfg["model"]["vq"]["embed_dims"][0]
        elif "vq" == self.output_type[:2]:
            output_dim = cfg["model"]["vq"]["embed_dims"][int(self.output_type[2:])]
        else:
            raise ValueError(f"Unsupported output type {self.output_type}.")
        self.output_dim = output_dim
        self.evaluator = UnSegEvaluator(output_dim, self.num_classes, self.extra_classes)

    def forward(self, img: torch.Tensor, aug_img: torch.Tensor = None, label: torch.Tensor = None,
                img_pos: torch.Tensor = None, img_path=None, it: int = -1, is_crf: bool = False
                ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], Tuple[torch.Tensor, torch.Tensor], torch.Tensor]:
        """train.py convention (train.py:81-83,:276-278): returns
        (total_loss (1,), output dict, (linear_preds, cluster_preds), feat_vqs (b, D, h, w)).

        eval.py calls ``model(img, label[, is_crf=...])`` and unpacks three values (eval.py:74,:212):
        when the second positional argument is an integer tensor and no ``label`` is given it is
        taken to be the label and (total_loss, output, (linear_preds, cluster_preds)) is returned."""
        legacy = label is None and aug_img is not None and not torch.is_floating_point(aug_img)
        if legacy:
            label, aug_img = aug_img, img
        model_loss = torch.zeros(1, device=img.device)
        code, feat_vqs, (_z_split, _z_quantized, _z_index), output = self.model(
            img=img, aug_img=aug_img, img_pos=img_pos, it=it)
        if self.training and self.stego_weight > 0.0:
            model_loss = model_loss + (output["stego-loss"] * self.stego_weight)
        if self.recon_weight > 0.0:
            model_loss = model_loss + output["recon-loss"] * self.recon_weight
        if self.vq_weight > 0.0:
            model_loss = model_loss + (output["vq-loss"] * self.vq_weight)
        if self.cls_weight > 0.0:
            model_loss = model_loss + (output["cls-loss"] * self.cls_weight)
        if self.mse_weight > 0.0:
            model_loss = model_loss + (output["mse-loss"] * self.mse_weight)
        output["loss"] = model_loss
        if self.output_type == "feat":
            out = code.detach()
        elif "vq" == self.output_type[:2]:
            out = feat_vqs.detach()
        else:
            raise ValueError(f"Unsupported output type {self.output_type}.")
        linear_loss, linear_preds, cluster_loss, cluster_preds = self.evaluator(out, img, label=label, is_crf=is_crf)
        output["linear-loss"] = linear_loss
        output["cluster-loss"] = cluster_loss
        total_loss = model_loss + linear_loss + cluster_loss
        if legacy:
            return total_loss, output, (linear_preds, cluster_preds)
        return total_loss, output, (linear_preds, cluster_preds), feat_vqs


def build_model(cfg: dict, name: str = None, world_size: int = 1) -> nn.Module:
    """build.py:35-69 for the one self-consistent EQUSS entry: a name containing "pqgo" builds
    PQGOWrapper(DIONPQGO).  Other research variants of the reference are outside this build."""
    name = (name or cfg.get("wandb", {}).get("name", "pqgo")).lower()
    if "pqgocls" in name or "pqgo" not in name:
        raise ValueError(f"Unsupported model type {name}.")
    return PQGOWrapper(cfg, DIONPQGO(cfg["model"], cfg["loss"]))

"""MI355X-native EQUSS hot path: DINO-ViT patch features -> expansion head -> product-quantisation
assign, behind the reference's own nn.Module surface.  The compute runs in hand-written HIP kernels
(csrc/) reached through a C ABI (include/equss_hip.h); PyTorch-ROCm supplies device memory, streams
and torch.distributed only.

Reference class                                   -> here
  model/dino/vision_transformer.py VisionTransformer -> featurizer.VisionTransformer
  model/dino/dino_featurizer.py DinoFeaturizer       -> featurizer.DinoFeaturizer
  model/dino_pqgo.py DIONPQGO / Codebook / ProductQuantizerWrapper -> pqgo.*
  model/quantizer.py VectorQuantizer / EMAVectorQuantizer / ...    -> quantizer.*
  model/evaluator.py UnSegEvaluator / ClusterLookup  -> evaluator.*
  model/metric.py UnSegMetrics                       -> metric.*
  wrapper/PQGOWrapper.py PQGOWrapper, build.py build_model -> wrapper.*
  utils/dist_utils.py                                -> dist_utils.*
"""
from . import configs, synth  # noqa: F401
from . import dist_utils, native  # noqa: F401
from .featurizer import DinoFeaturizer, VisionTransformer  # noqa: F401
from .pqgo import DIONPQGO, Codebook, ProductQuantizerWrapper, STEGOLoss  # noqa: F401
from .evaluator import UnSegEvaluator, ClusterLookup  # noqa: F401
from .metric import UnSegMetrics  # noqa: F401
from .wrapper import PQGOWrapper, build_model  # noqa: F401
from . import quantizer  # noqa: F401
from .index_dump import CodeIndexDump  # noqa: F401

__all__ = ["CodeIndexDump", "configs", "synth", "native", "dist_utils", "quantizer", "DinoFeaturizer", "VisionTransformer",
           "DIONPQGO", "Codebook", "ProductQuantizerWrapper", "STEGOLoss", "UnSegEvaluator", "ClusterLookup",
           "UnSegMetrics", "PQGOWrapper", "build_model"]

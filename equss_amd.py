"""Importable alias: ``import equss_amd`` is the package in the directory whose name (with hyphens)
cannot appear in an import statement."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")

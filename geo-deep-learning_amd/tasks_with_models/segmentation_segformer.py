"""Alias of geo_deep_learning.tasks_with_models.segmentation_segformer (configs/segformer_config_RGB.yaml:41)."""

from geo_deep_learning.tasks_with_models.segmentation_segformer import SegmentationSegformer  # noqa: F401

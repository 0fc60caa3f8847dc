"""Alias of geo_deep_learning.tasks_with_models.segmentation_unetplus (configs/unetplus_config_RGB.yaml)."""

from geo_deep_learning.tasks_with_models.segmentation_unetplus import SegmentationUnetPlus  # noqa: F401

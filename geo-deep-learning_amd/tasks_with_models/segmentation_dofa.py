"""Alias of geo_deep_learning.tasks_with_models.segmentation_dofa (both import roots resolve, like
the reference's ``train.py`` script dir vs package imports)."""

from geo_deep_learning.tasks_with_models.segmentation_dofa import SegmentationDOFA  # noqa: F401

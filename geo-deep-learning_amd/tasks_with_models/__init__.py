"""Script-dir import root used by the reference's YAML class_paths
(``tasks_with_models.segmentation_dofa.SegmentationDOFA``, configs/dofa_config_RGB.yaml:45)."""

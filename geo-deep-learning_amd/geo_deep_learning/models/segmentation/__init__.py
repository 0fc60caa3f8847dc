"""MI355X-native drop-in for geo_deep_learning.models.segmentation (DOFA path)."""

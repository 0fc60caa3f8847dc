"""MI355X-native drop-in for the matching geo_deep_learning package (hot path only)."""

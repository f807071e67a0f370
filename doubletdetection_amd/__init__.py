"""MI355X-native drop-in for DoubletDetection's BoostClassifier hot path."""

"""Model classes with the reference's public surface, backed by libcae_hip (ConvAEModel, UNET, DSDataset, ModelMetric)."""

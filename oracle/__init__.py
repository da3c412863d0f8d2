"""CPU oracles (test infrastructure): cae_oracle (ConvAE path), unet_oracle (UNET path)."""

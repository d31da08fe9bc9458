"""MI355X-native StyleGAN2 generator hot path."""

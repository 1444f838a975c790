"""MI355X-native k-diffusion sampling hot path."""

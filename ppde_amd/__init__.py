"""ppde_amd — MI355X-native Plug & Play Directed Evolution sampler hot path."""
__version__ = "0.1.0"

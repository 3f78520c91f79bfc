"""gkgnet_amd — MI355X-native (gfx950) implementation of GKGNet's Group-KNN graph-convolution hot path."""
__version__ = "0.1.0"

"""dclip_amd — MI355X-native DCLIP distillation step."""
__version__ = "0.1.0"

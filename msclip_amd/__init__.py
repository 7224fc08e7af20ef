"""msclip_amd -- MI355X-native MS-CLIP-S contrastive hot path."""
__version__ = "0.1.0"

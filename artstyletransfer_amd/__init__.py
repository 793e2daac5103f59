"""MI355X-native pyramid neural style transfer (drop-in for irenemizus/ArtStyleTransfer's hot path)."""
__all__ = ["_lib", "engine"]

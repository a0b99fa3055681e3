"""smoltts_amd: MI355X-native DualAR + Mimi decode hot path."""
__version__ = "0.1.0"

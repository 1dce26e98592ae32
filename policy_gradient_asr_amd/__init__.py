"""MI355X-native acoustic-model policy-gradient training path (drop-in for the hot path of
ana-kuznetsova/Policy-Gradient-ASR).  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"

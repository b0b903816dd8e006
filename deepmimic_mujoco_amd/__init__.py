"""MI355X-native batched DeepMimic humanoid engine."""
__version__ = "0.1.0"

"""dagl_amd: MI355X-native dynamic patch-graph attention block (DAGL ``CE``)."""
__version__ = "0.1.0"

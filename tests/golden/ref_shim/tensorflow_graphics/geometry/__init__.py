"""Stub: mint/core/metrics.py imports it at module level; nothing on the FACT forward path uses it."""

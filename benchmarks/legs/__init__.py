"""The legs of bench.py beside the headline: one module per BASELINE config / diagnostic (all outside the timed region)."""

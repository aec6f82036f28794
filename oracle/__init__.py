"""CPU oracles: torch / numpy restatements of the reference algorithms.  TEST INFRASTRUCTURE ONLY -- imported by tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py, never by the product path (vidsgg-big_amd/)."""

"""ORACLE — CPU restatement of the reference hot path.  Test infrastructure only.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
(`2048_amd/`, `game2048/`) never imports this package.
"""

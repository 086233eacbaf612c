"""CPU oracle for the BPE hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (minbpe_amd/) never does.
"""
from .oracle import (  # noqa: F401
    build, get_stats, merge, merge_chunks, train, train_fast, encode, dedup, OracleEmptyStats,
)

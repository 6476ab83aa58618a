"""CPU oracle (test infrastructure).  See quake_oracle.c for scope and citations.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from .oracle import *  # noqa: F401,F403

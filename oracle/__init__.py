"""CPU oracle for the VFS Ranker linear-search path -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
(see oracle/vfs_oracle.c).  PARITY UNPINNED (the reference holds no golden vectors).
"""

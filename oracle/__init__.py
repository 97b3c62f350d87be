"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see knox_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""

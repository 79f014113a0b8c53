"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see xm_oracle.py / xm_oracle.c headers).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""

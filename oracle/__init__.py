"""CPU oracle for the himut SBS pileup scan -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  See oracle/himut_oracle.c."""

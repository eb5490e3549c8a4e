"""CPU oracle package -- TEST INFRASTRUCTURE (see oracle/oracle.c header). Import only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""

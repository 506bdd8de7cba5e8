"""CPU oracle for the LiDAR projection + instance point-filter path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this.
"""

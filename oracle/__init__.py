"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (bdd_amd/, bdd_amd/csrc/libbdd_mma_hip.so) never does.
"""

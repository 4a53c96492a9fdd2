"""TEST INFRASTRUCTURE ONLY — CPU restatements used as the parity checker.

Nothing under oracle/ is part of the product path: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  The product (v3d_amd) never routes compute through this package.
"""

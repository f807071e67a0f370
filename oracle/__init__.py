"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the reference algorithm (see dd_oracle.py).

Nothing in the product package imports this; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do.
"""

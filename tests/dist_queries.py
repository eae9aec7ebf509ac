"""Query variants for the multi-rank tests: q18 with a HAVING threshold that leaves rows at the tiny
scale factor the CPU tests run at (the TPCH constant 300 selects nothing below SF ~0.05)."""
from sdqlpy_amd import tpch_queries as Q


def register():
    Q.register("q18_low", Q.large_orders(230), Q.TPCH_ORDER["q18"])

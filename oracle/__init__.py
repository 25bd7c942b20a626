"""CPU oracle (test infrastructure only -- see oracle/oracle.cpp). Importable from tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() only; the product package never imports this."""

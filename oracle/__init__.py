"""CPU oracle (test infrastructure only; see oracle_np.py header)."""

"""NMF and projection-back (mirror of the reference's src/algorithm for the HIP hot path)."""

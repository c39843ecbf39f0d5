"""Blind source separation classes (mirror of the reference's src/bss for the HIP hot path)."""

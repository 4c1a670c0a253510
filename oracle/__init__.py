"""CPU oracle for the BSVD hot path -- test infrastructure only (see bsvd_oracle.py header)."""

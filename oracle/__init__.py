"""CPU oracle (test infrastructure only; see pqd_oracle.h)."""

"""CPU oracle package (test infrastructure only; see oracle/q2048_oracle.h)."""

"""CPU oracle package (test infrastructure only; see aq_oracle.c)."""

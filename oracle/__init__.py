"""CPU oracle -- test infrastructure only (see gsr_oracle.c)."""

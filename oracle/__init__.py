"""CPU oracle (test infrastructure only). See oracle/gomoku_oracle.h."""

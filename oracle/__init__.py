"""CPU oracle for the AGAThA guided-alignment path (test infrastructure only; see agatha_oracle.c)."""

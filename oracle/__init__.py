"""CPU oracle — test infrastructure only (see oracle/trt_oracle.c)."""

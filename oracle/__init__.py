"""CPU oracle for the X-engine hot path.  TEST INFRASTRUCTURE ONLY (see xeng_oracle.c)."""

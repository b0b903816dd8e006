"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/dm_oracle.h)."""

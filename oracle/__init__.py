"""CPU oracle of the SAM-Decoding hot path -- TEST INFRASTRUCTURE ONLY (see sam_oracle.c)."""

"""CPU restatement of the reference algorithm -- TEST INFRASTRUCTURE ONLY (see han_oracle.py)."""

"""CPU oracle package -- TEST INFRASTRUCTURE, NOT PRODUCT (see oracle/oracle.py)."""

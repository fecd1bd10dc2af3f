"""Test infrastructure only: CPU restatement of the reference algorithm (see blr_oracle.py)."""

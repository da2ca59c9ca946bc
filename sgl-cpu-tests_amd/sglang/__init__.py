"""Import shim only: the reference harness imports one enum from sglang (/root/reference/test_moe.py:4)."""

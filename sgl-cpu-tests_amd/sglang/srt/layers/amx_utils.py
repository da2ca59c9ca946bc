"""`from sglang.srt.layers.amx_utils import CPUQuantMethod` (/root/reference/test_moe.py:4) — enum shim."""
from enum import IntEnum


class CPUQuantMethod(IntEnum):
    UNQUANT = 0
    INT8_W8A8 = 1
    FP8_W8A16 = 2

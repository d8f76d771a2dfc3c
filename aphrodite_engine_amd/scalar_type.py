"""Working pure-Python ScalarType for the integer weight types the hot path uses.

Mirrors the C++ class kernels/core/scalar_type.hpp:12-260 (the reference's
Python fallback aphrodite/_core_ext.py:28-171 is a typing mock whose min()/max()
raise).  value = stored - bias."""
from dataclasses import dataclass


@dataclass(frozen=True)
class ScalarType:
    size_bits: int
    bias: int = 0
    signed: bool = False

    @classmethod
    def uint(cls, size_bits: int, bias=None) -> "ScalarType":
        return cls(size_bits, bias or 0, False)

    @classmethod
    def int_(cls, size_bits: int, bias=None) -> "ScalarType":
        return cls(size_bits, bias or 0, True)

    def is_integer(self) -> bool:
        return True

    def is_signed(self) -> bool:
        return self.signed

    def has_bias(self) -> bool:
        return self.bias != 0

    def min(self) -> int:
        return (-(1 << (self.size_bits - 1)) if self.signed else 0) - self.bias

    def max(self) -> int:
        return ((1 << (self.size_bits - (1 if self.signed else 0))) - 1) - self.bias

    def __str__(self) -> str:
        s = f"{'int' if self.signed else 'uint'}{self.size_bits}"
        return s + (f"b{self.bias}" if self.bias else "")


class scalar_types:
    int4 = ScalarType.int_(4)
    uint4 = ScalarType.uint(4)
    int8 = ScalarType.int_(8)
    uint8 = ScalarType.uint(8)
    uint4b8 = ScalarType.uint(4, 8)
    uint8b128 = ScalarType.uint(8, 128)

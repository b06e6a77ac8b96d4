"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes wrappers for pdt_oracle_decoding.c."""

__all__ = []


def declare(L):
    pass

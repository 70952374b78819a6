"""GradeSet (reference: src/grade_set.rs) on 64-bit masks; thin wrapper over the C++ host code."""
from __future__ import annotations

from . import _lib


class GradeSet:
    __slots__ = ("mask",)

    def __init__(self, mask=0):
        self.mask = int(mask)

    @staticmethod
    def empty():                                   # grade_set.rs:52-55
        return GradeSet(0)

    @staticmethod
    def single(k):                                 # grade_set.rs:65-71
        return GradeSet(_lib.lib().gaast_gs_single(int(k)))

    @staticmethod
    def range(x, y):                               # grade_set.rs:74-80
        return GradeSet(_lib.lib().gaast_gs_range(int(x), int(y)))

    @staticmethod
    def of(grades):
        m = 0
        for k in grades:
            m |= 1 << int(k)
        return GradeSet(m)

    def intersection(self, rhs):                   # grade_set.rs:85-91
        return GradeSet(self.mask & rhs.mask)

    def __add__(self, rhs):                        # grade_set.rs:287-293
        return GradeSet(self.mask | rhs.mask)

    def __mul__(self, rhs):                        # grade_set.rs:305-327
        return GradeSet(_lib.lib().gaast_gs_mul(self.mask, rhs.mask))

    def __eq__(self, rhs):                         # grade_set.rs:35-42
        return isinstance(rhs, GradeSet) and self.mask == rhs.mask

    def __hash__(self):
        return hash(self.mask)

    def iter(self):                                # grade_set.rs:94-96
        return [k for k in range(64) if (self.mask >> k) & 1]

    __iter__ = lambda self: iter(self.iter())

    def is_empty(self):                            # grade_set.rs:124-126
        return self.mask == 0

    def is_single(self):                           # grade_set.rs:129-138
        return bin(self.mask).count("1") == 1

    def contains(self, k):                         # grade_set.rs:141-146
        return bool((self.mask >> int(k)) & 1) if k >= 0 else False

    def includes(self, other):                     # grade_set.rs:149-151
        return (self.mask | other.mask) == self.mask

    def is_just(self, k):                          # grade_set.rs:154-156
        return self.mask == (1 << int(k))

    def add_grade(self, k):                        # grade_set.rs:159-165
        return GradeSet(self.mask | (1 << int(k)))

    def rm_grade(self, k):                         # grade_set.rs:168-173
        return GradeSet(self.mask & ~(1 << int(k)))

    def parts_contributing_to_product(self, kind, left, right):   # grade_set.rs:239-252
        import ctypes as C
        ol, orr = C.c_uint64(), C.c_uint64()
        _lib.lib().gaast_gs_parts_contributing_to_product(self.mask, kind, left.mask, right.mask,
                                                          C.byref(ol), C.byref(orr))
        return GradeSet(ol.value), GradeSet(orr.value)

    def __repr__(self):
        return f"GradeSet{self.iter()}"

"""(replace, mask_complement, mask_structure, transpose_first, transpose_second) -> builtin
GrB_DESC_* handle; mirrors graphblas/core/descriptor.py:51-156."""
from . import _lib


class Descriptor:
    __slots__ = ("name", "output_replace", "mask_complement", "mask_structure", "transpose_first", "transpose_second")

    def __init__(self, name, r, c, s, t0, t1):
        self.name = name
        self.output_replace, self.mask_complement, self.mask_structure = r, c, s
        self.transpose_first, self.transpose_second = t0, t1

    @property
    def _carg(self):
        return _lib.handle(self.name)

    def __repr__(self):
        return self.name


def _name(r, c, s, t0, t1):
    return "GrB_DESC_" + ("R" if r else "") + ("S" if s else "") + ("C" if c else "") + ("T0" if t0 else "") + ("T1" if t1 else "")


_desc_map = {}
for _r in (False, True):
    for _c in (False, True):
        for _s in (False, True):
            for _t0 in (False, True):
                for _t1 in (False, True):
                    _key = (_r, _c, _s, _t0, _t1)
                    _desc_map[_key] = Descriptor(_name(*_key), *_key) if any(_key) else None


def lookup(*, output_replace=False, mask_complement=False, mask_structure=False, transpose_first=False,
           transpose_second=False, **opts):
    if opts:
        raise ValueError(
            "Extra descriptor options not possible with 'mi355x' backend; got " + ", ".join(str(x) for x in opts)
        )
    return _desc_map[(bool(output_replace), bool(mask_complement), bool(mask_structure), bool(transpose_first),
                      bool(transpose_second))]

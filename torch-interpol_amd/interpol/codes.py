"""Boundary / interpolation aliases -> integer codes.

Same vocabulary and error behaviour as the reference's
`bound_to_nitorch` / `inter_to_nitorch` (reference interpol/autograd.py:56-154)
and its enums (interpol/bounds.py:8-21, interpol/splines.py:7-15); written as
lookup tables instead of if-chains.
"""
from enum import Enum


class BoundType(Enum):
    zero = zeros = 0
    replicate = nearest = 1
    dct1 = mirror = 2
    dct2 = reflect = 3
    dst1 = antimirror = 4
    dst2 = antireflect = 5
    dft = wrap = 6


class InterpolationType(Enum):
    nearest = zeroth = 0
    linear = first = 1
    quadratic = second = 2
    cubic = third = 3
    fourth = 4
    fifth = 5
    sixth = 6
    seventh = 7


class ExtrapolateType(Enum):
    no = 0      # keep samples inside (-0.05, n-1+0.05)
    yes = 1
    hist = 2    # keep samples inside (-0.55, n-1+0.55)


_BOUND_ALIASES = {
    "replicate": 1, "repeat": 1, "border": 1, "nearest": 1,
    "zero": 0, "zeros": 0, "constant": 0,
    "dct2": 3, "reflect": 3, "reflection": 3, "neumann": 3,
    "dct1": 2, "mirror": 2,
    "dft": 6, "wrap": 6, "circular": 6,
    "dst2": 5, "antireflect": 5, "dirichlet": 5,
    "dst1": 4, "antimirror": 4,
}
_BOUND_NAMES = {0: "zero", 1: "replicate", 2: "dct1", 3: "dct2", 4: "dst1", 5: "dst2", 6: "dft"}

_ORDER_ALIASES = {
    "nearest": 0, "linear": 1, "quadratic": 2, "cubic": 3,
    "fourth": 4, "fifth": 5, "sixth": 6, "seventh": 7,
}
_ORDER_NAMES = {v: k for k, v in _ORDER_ALIASES.items()}


def bound_to_code(b):
    """One boundary condition (str | int | BoundType) -> int code 0..6."""
    if isinstance(b, BoundType):
        return b.value
    if isinstance(b, str):
        try:
            return _BOUND_ALIASES[b.lower()]
        except KeyError:
            raise ValueError(f'Unknown boundary condition {b}') from None
    if isinstance(b, int) and not isinstance(b, bool):
        try:
            return BoundType(b).value
        except ValueError:
            raise ValueError(f'{b} is not a valid BoundType') from None
    raise ValueError(f'Unknown boundary condition {b}')


def order_to_code(o):
    """One interpolation order (str | int | InterpolationType) -> int 0..7."""
    if isinstance(o, InterpolationType):
        return o.value
    if isinstance(o, str):
        try:
            return _ORDER_ALIASES[o.lower()]
        except KeyError:
            raise ValueError(f'Unknown interpolation order {o}') from None
    if isinstance(o, int) and not isinstance(o, bool) and 0 <= o <= 7:
        return int(o)
    raise ValueError(f'Unknown interpolation order {o}')


def _listify(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def bound_to_nitorch(bound, as_type='str'):
    """Drop-in for reference interpol/autograd.py:56-106."""
    intype = type(bound)
    codes = [bound_to_code(b) for b in _listify(bound)]
    if as_type in ('int', int):
        out = codes
    elif as_type in ('str', str):
        out = [_BOUND_NAMES[c] for c in codes]
    else:
        out = [BoundType(c) for c in codes]
    if issubclass(intype, (list, tuple)):
        return intype(out)
    return out[0]


def inter_to_nitorch(inter, as_type='str'):
    """Drop-in for reference interpol/autograd.py:109-154."""
    intype = type(inter)
    codes = [order_to_code(o) for o in _listify(inter)]
    if as_type in ('str', str):
        out = [_ORDER_NAMES[c] for c in codes]
    elif as_type == 'enum':
        out = [InterpolationType(c) for c in codes]
    else:
        out = codes
    if issubclass(intype, (list, tuple)):
        return intype(out)
    return out[0]


def pad_codes(x, dim):
    """Pad with the last element, truncate when longer (the reference's
    `pad_list_int`, interpol/jit_utils.py:9-15: a 3-element list on a 2-D problem
    silently keeps its first two entries)."""
    x = list(x)
    if len(x) < dim:
        x = x + x[-1:] * (dim - len(x))
    return x[:dim]

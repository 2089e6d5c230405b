"""Test helper: turn the golden manifest's API-level bound/interpolation
arguments into integer code lists using the product's own alias tables."""


def to_int_lists(bound, interpolation):
    from interpol.codes import bound_to_code, order_to_code
    b = bound if isinstance(bound, (list, tuple)) else [bound]
    o = interpolation if isinstance(interpolation, (list, tuple)) else [interpolation]
    return [bound_to_code(x) for x in b], [order_to_code(x) for x in o]

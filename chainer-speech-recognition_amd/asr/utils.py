"""Object <-> dict helpers used by the Configuration classes (reference asr/utils.py:3-40; the ANSI printing
helpers of that file are out of scope)."""


def to_dict(obj):
    return {k: v for k, v in vars(obj).items() if not k.startswith("_")}


def _set(obj, params):
    for k, v in params.items():
        if isinstance(v, list):
            v = tuple(v)
        setattr(obj, k, v)


def to_object(params):
    class Object(object):
        pass
    o = Object()
    _set(o, params)
    return o


def dump_dict(params, indent=0):
    for k in sorted(params):
        print("  " * indent + "{}: {}".format(k, params[k]))

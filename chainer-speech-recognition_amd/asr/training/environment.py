"""Hyper-parameters a running training process re-reads on SIGUSR1 (asr/training/environment.py:13-37).

The train scripts hang the values they want to steer from outside on an Environment -- learning rate, momentum, the augmentation
switches (run/ctc/cnn/train.py:115-131) --, `save()` writes them to a JSON file, the operator edits that file and sends
`kill -USR1 <pid>`: the handler re-reads the file and calls the script's callback, which pushes the values into the optimiser.
Same rules as the reference: only attributes that already exist are overwritten (asr/training/environment.py:4-11), nested option
objects (the augmentation switches) are walked, a missing or unparsable file is an error, names with a leading underscore are private.
Nothing here touches the GPU: the callback runs in the main thread between two Python byte codes, i.e. between two steps' launches."""
import json
import os
import signal
import sys


def _is_leaf(value):
    return isinstance(value, (bool, int, float, str, list, tuple, type(None)))


def _public_names(obj):
    names = getattr(type(obj), "__slots__", None)
    if names is None:
        names = vars(obj).keys()
    return sorted(n for n in names if not n.startswith("_"))


def _to_dict(obj):
    out = {}
    for name in _public_names(obj):
        value = getattr(obj, name)
        if callable(value):
            continue
        out[name] = value if _is_leaf(value) else _to_dict(value)
    return out


def _overwrite(obj, values):
    for name, value in values.items():
        if name.startswith("_") or not hasattr(obj, name):
            continue                            # the file cannot add attributes, only change the ones the script declared
        if isinstance(value, dict):
            _overwrite(getattr(obj, name), value)
        else:
            setattr(obj, name, value)


class Environment(object):
    def __init__(self, filename, handler, signum=signal.SIGUSR1):
        self._filename = filename
        self._handler = handler
        self._signum = signum
        signal.signal(signum, self.handler)

    def handler(self, _signum=None, _frame=None):
        self.load()
        self._handler()

    def save(self):
        with open(self._filename, "w") as f:
            json.dump(_to_dict(self), f, indent=4, sort_keys=True, separators=(",", ": "))

    def load(self):
        values = None
        if os.path.isfile(self._filename):
            with open(self._filename, "r") as f:
                try:
                    values = json.load(f)
                except ValueError:
                    values = None
        assert values is not None, "could not load {}".format(self._filename)
        _overwrite(self, values)

    def dump(self, out=None):
        out = out or sys.stdout
        out.write("[Environment]\n")

        def walk(d, depth):
            for name in sorted(d):
                if isinstance(d[name], dict):
                    out.write("\t" * depth + "%s:\n" % name)
                    walk(d[name], depth + 1)
                else:
                    out.write("\t" * depth + "%s:\t%s\n" % (name, d[name]))
        walk(_to_dict(self), 1)

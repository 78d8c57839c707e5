"""Feature / bucket configuration with the reference's field names and JSON file format (asr/model/base.py:4-37).

Configurations are declared as tables (``FIELDS`` of every class in the MRO, merged base first) instead of a list of
attribute assignments; instances still expose plain attributes, ``save`` writes the same sorted, 4-space JSON object and
``load`` returns True when the file was read and None when it does not exist, as the reference's does.
"""
import json
import os

from ..utils import _set, dump_dict, to_dict


class FieldTable(object):
    """attributes initialised from the FIELDS dictionaries along the class hierarchy"""

    FIELDS = {}

    def __init__(self, **overrides):
        for klass in reversed(type(self).__mro__):
            for name, default in vars(klass).get("FIELDS", {}).items():
                setattr(self, name, default)
        for name, value in overrides.items():
            if not hasattr(self, name):
                raise AttributeError("unknown configuration field %r" % name)
            setattr(self, name, value)

    # -- reference surface ------------------------------------------------------------------------
    def dump(self):
        print("[Configuration]")
        dump_dict(to_dict(self), 1)

    def check(self):
        """hook for subclasses: raise if the configuration must not be written in this state"""

    def save(self, filename):
        self.check()
        text = json.dumps(to_dict(self), indent=4, sort_keys=True, separators=(",", ": "))
        with open(filename, "w") as fp:
            fp.write(text)

    def load(self, filename):
        if not os.path.isfile(filename):
            return None
        print("Loading {} ...".format(filename))
        try:
            with open(filename, "r") as fp:
                stored = json.load(fp)
        except ValueError:
            raise Exception("could not load {}".format(filename))
        _set(self, stored)
        return True


class Configuration(FieldTable):
    # feature extraction (run/ctc/cnn/args.py:18-25) and the length buckets of the readers
    FIELDS = dict(sampling_rate=16000, frame_width=0.032, frame_shift=0.01, num_mel_filters=40, window_func="hanning",
                  using_delta=True, using_delta_delta=True, bucket_split_sec=0.5)


class NeedsVocabulary(Configuration):
    """model configurations: writable only once the vocabulary size is known (asr/model/cnn.py:22-24)"""

    FIELDS = dict(vocab_size=-1)

    def check(self):
        assert self.vocab_size > 0


def configure():
    return Configuration()

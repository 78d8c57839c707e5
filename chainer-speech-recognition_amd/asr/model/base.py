"""Feature / bucket configuration -- asr/model/base.py:4-37 (JSON save / load with the same keys)."""
import json
import os

from ..utils import _set, dump_dict, to_dict


class Configuration():
    def __init__(self):
        self.sampling_rate = 16000
        self.frame_width = 0.032
        self.frame_shift = 0.01
        self.num_mel_filters = 40
        self.window_func = "hanning"
        self.using_delta = True
        self.using_delta_delta = True
        self.bucket_split_sec = 0.5

    def dump(self):
        print("[Configuration]")
        dump_dict(to_dict(self), 1)

    def save(self, filename):
        with open(filename, "w") as f:
            json.dump(to_dict(self), f, indent=4, sort_keys=True, separators=(',', ': '))

    def load(self, filename):
        if os.path.isfile(filename):
            print("Loading {} ...".format(filename))
            with open(filename, "r") as f:
                try:
                    params = json.load(f)
                except Exception:
                    raise Exception("could not load {}".format(filename))
            _set(self, params)
            return True
        return None


def configure():
    return Configuration()

"""Which augmentations ``Processor.extract_batch_features`` applies (the option object of asr/data/augment.py:3-16).

Three independent switches, all off by default:
  change_speech_rate   time axis of the power spectrum resampled by a random factor in [0.8, 1.2]   (asr/fft.py:25-34)
  change_vocal_tract   frequency axis resampled likewise (vocal-tract-length perturbation)           (asr/fft.py:36-48)
  add_noise            white noise of random gain added to the waveform                              (asr/data/processing.py:74-78)
"""

_SWITCHES = ("change_vocal_tract", "change_speech_rate", "add_noise")


class AugmentationOption(object):
    __slots__ = _SWITCHES

    def __init__(self, **switches):
        for name in _SWITCHES:
            setattr(self, name, bool(switches.pop(name, False)))
        if switches:
            raise TypeError("unknown augmentation switch: %s" % ", ".join(sorted(switches)))

    def using_augmentation(self):
        """True when at least one switch is on (the reference calls fft.augment_specgram only then)."""
        return any(getattr(self, name) for name in _SWITCHES)

    def __repr__(self):
        return "AugmentationOption(%s)" % ", ".join("%s=%s" % (n, getattr(self, n)) for n in _SWITCHES)

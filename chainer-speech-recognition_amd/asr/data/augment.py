"""asr/data/augment.py: which augmentations Processor.extract_batch_features applies."""


class AugmentationOption(object):
    def __init__(self):
        self.change_vocal_tract = False
        self.change_speech_rate = False
        self.add_noise = False

    def using_augmentation(self):
        return bool(self.change_vocal_tract or self.change_speech_rate or self.add_noise)

from .ctc import gram_ctc, connectionist_temporal_classification  # noqa: F401

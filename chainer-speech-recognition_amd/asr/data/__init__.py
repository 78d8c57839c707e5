"""Feature extraction and minibatch assembly of the reference's asr/data package (the parts on the train-step path)."""

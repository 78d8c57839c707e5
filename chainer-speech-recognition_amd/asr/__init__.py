"""MI355X-native drop-in for the `asr` operator API of musyoku/chainer-speech-recognition.

Sub-packages mirror the reference: ``asr.nn`` (operator API), ``asr.model`` (AcousticModel),
``asr.loss`` (CTC / Gram-CTC), ``asr.fft`` (log-mel filterbank features).  Everything numeric runs in
hand-written HIP kernels behind the C ABI of ``libasr_hip.so`` (include/asr_hip.h); PyTorch supplies
device memory, streams, autograd bookkeeping and torch.distributed (RCCL).
"""
